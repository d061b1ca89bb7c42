"""CPU-side checks of the C ABI: the library builds, loads and exports every symbol that
include/sqlrs_hip.h declares; without a GPU the product path fails loudly (no fallback)."""
import ctypes
import os
import re

import pytest

import sqlrs_amd
from sqlrs_amd import abi, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    return ctypes.CDLL(build.build(verbose=False))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sqlrs_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sqlrs_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 35
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in sqlrs_hip.h but not exported: {missing}"


def test_version(lib):
    lib.sqlrs_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.sqlrs_version()


def test_struct_layouts_match_header():
    # sizes implied by the C declarations (LP64)
    assert ctypes.sizeof(abi.Column) == 48
    assert ctypes.sizeof(abi.Batch) == 32
    assert ctypes.sizeof(abi.ExprNode) == 40
    assert ctypes.sizeof(abi.Expr) == 16
    assert ctypes.sizeof(abi.AggFunc) == 32
    assert ctypes.sizeof(abi.OrderBy) == 24


def test_no_cpu_fallback_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    ctx = ctypes.c_void_p()
    lib.sqlrs_ctx_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    st = lib.sqlrs_ctx_create(0, ctypes.byref(ctx))
    assert st == abi.ERR_DEVICE and not ctx.value
    with pytest.raises(sqlrs_amd.ExecutorError):
        sqlrs_amd.new_ctx(0)


def test_product_does_not_reference_oracle():
    """Nothing under sqlrs_amd/ may import, link or call the oracle."""
    pkg = os.path.join(ROOT, "sqlrs_amd")
    for dirpath, _, files in os.walk(pkg):
        if "build" in dirpath.split(os.sep)[-1:]:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                for line in text.splitlines():
                    code = line.split("//")[0].split("#")[0] if f.endswith(".py") is False else line
                    if re.search(r"(import|include|CDLL|dlopen).*oracle", code):
                        raise AssertionError(f"{f}: {line.strip()}")


def test_bench_self_launch_command(monkeypatch):
    """`python bench.py --gpus N` without a launcher re-runs its own command line as N ranks under
    torch.distributed.run on 127.0.0.1 (what the driver's N > 1 command relies on)"""
    import importlib
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    assert bench.self_launch(4) == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
