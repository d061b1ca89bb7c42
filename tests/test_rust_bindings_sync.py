"""bindings/rust cannot be compiled in this image (no Rust toolchain), so it is kept MECHANICALLY in sync with
include/sqlrs_hip.h: ffi.rs is the generator's output for the current header, every function of the header is
declared in ffi.rs with the same number of arguments and the same pointer depth per argument, every declared
function is exported by libsqlrs_hip.so, and every `sqlrs_*` call in the hand-written shim names a declared
function and passes the declared number of arguments."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_rust_ffi as gen  # noqa: E402

RUST = os.path.join(ROOT, "bindings", "rust", "src")


def header_parsed():
    with open(gen.HEADER) as f:
        return gen.parse_header(f.read())


def rust_fns():
    text = open(os.path.join(RUST, "ffi.rs")).read()
    out = {}
    for m in re.finditer(r"pub fn (sqlrs_\w+)\((.*?)\)( -> [^;]+)?;", text):
        args = [a.strip() for a in m.group(2).split(",") if a.strip()]
        out[m.group(1)] = [a.split(":", 1)[1].strip() for a in args]
    return out


def test_ffi_rs_is_the_generators_output_for_the_current_header():
    structs, enums, opaque, funcs = header_parsed()
    assert open(os.path.join(RUST, "ffi.rs")).read() == gen.emit(structs, enums, opaque, funcs), \
        "include/sqlrs_hip.h changed: run python tools/gen_rust_ffi.py"


def test_every_header_function_is_declared_with_matching_arity_and_pointer_depth():
    _, _, opaque, funcs = header_parsed()
    rf = rust_fns()
    header_text = gen.strip_comments(open(gen.HEADER).read())
    assert len(funcs) >= 73 and set(rf) == {n for n, _, _ in funcs}
    for name, params, _ in funcs:
        # independent count: commas of the C declaration
        m = re.search(r"\b" + name + r"\s*\(([^;{}]*?)\)\s*;", header_text, flags=re.S)
        c_args = [a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"]
        assert len(rf[name]) == len(c_args) == len(params), name
        for c_arg, r_ty in zip(c_args, rf[name]):
            assert c_arg.count("*") == r_ty.count("*"), (name, c_arg, r_ty)
    text = open(os.path.join(RUST, "ffi.rs")).read()
    for t in opaque:
        assert f"pub struct {t} " in text, t
    for s in ("sqlrs_column_t", "sqlrs_batch_t", "sqlrs_expr_node_t", "sqlrs_expr_t", "sqlrs_agg_func_t", "sqlrs_order_by_t"):
        assert f"pub struct {s} {{" in text


def test_declared_functions_are_exported_by_the_library():
    from sqlrs_amd import build
    if not os.path.exists(build.OUT):
        build.build(verbose=False)
    syms = subprocess.run(["nm", "-D", "--defined-only", build.OUT], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in syms.splitlines() if " T " in l}
    assert set(rust_fns()) <= exported


def split_top_level(args: str):
    out, depth, cur = [], 0, ""
    for ch in args:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def test_shim_calls_name_declared_functions_with_the_declared_number_of_arguments():
    rf = rust_fns()
    calls = 0
    used = set()
    for fn in ("executors.rs", "convert.rs"):
        text = open(os.path.join(RUST, fn)).read()
        for m in re.finditer(r"\b(sqlrs_[a-z_0-9]+)\(", text):
            name = m.group(1)
            assert name in rf, f"{fn}: {name} is not declared in ffi.rs"
            # balanced argument list
            i, depth = m.end(), 1
            while depth:
                depth += {"(": 1, ")": -1}.get(text[i], 0)
                i += 1
            args = split_top_level(text[m.end():i - 1])
            assert len(args) == len(rf[name]), f"{fn}: {name} called with {len(args)} arguments, declared with {len(rf[name])}"
            calls += 1
            used.add(name)
        for m in re.finditer(r"Guard\(\w+, (sqlrs_\w+)\)", text):  # destroy functions passed as values
            assert m.group(1) in rf
            used.add(m.group(1))
    assert calls >= 30
    # every operator family of the header has its create / push / finish / destroy calls in the shim
    for fam in ("filter", "hash_join", "hash_agg", "order", "project", "limit", "simple_agg", "join_agg"):
        fam_fns = {n for n in rf if n.startswith(f"sqlrs_{fam}_")}
        core = {n for n in fam_fns if n.rsplit("_", 1)[-1] in ("create", "push", "finish", "destroy") or n.endswith(("build_push", "build_finish", "probe_push"))}
        assert core <= used, (fam, sorted(core - used))
    for s in ("hip_visit_physical_filter", "hip_visit_physical_hash_join", "hip_visit_physical_hash_agg", "hip_visit_physical_order",
              "as_physical_hash_join", "as_physical_filter", "sqlrs_join_agg_set_probe_filter", "sqlrs_hash_agg_set_filter"):
        assert s in open(os.path.join(RUST, "executors.rs")).read()
