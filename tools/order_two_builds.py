"""ORDER BY on 1e8 rows alternating between two BUILDS of the library (LIB_A, LIB_B) in one process on the same columns:
compile-time variants of order_fast.hip (e.g. -DOW_RANK_SELECT) compared without the process-to-process spread.
  FILE=order_fast FLAGS=-DOW_RANK_SELECT NAME=sel bash tools/build_obj_variant.sh
  LIB_A=sqlrs_amd/csrc/libsqlrs_hip.so LIB_B=tools/_bin/lib_sel.so python tools/order_two_builds.py"""
import os as _os; _os.environ.setdefault("SQLRS_HOOKS", "1")  # the SQLRS_* hooks are consulted only in a process that opts in
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sqlrs_amd import abi, datagen
from sqlrs_amd.expr import InputRef
dev = torch.device("cuda", 0)
n = int(float(os.environ.get("N", 1e8)))
g = torch.Generator(device=dev).manual_seed(3)
keys = {"i64_31bit": (datagen.fill_chunks(torch.empty(n, dtype=torch.int64, device=dev), lambda i: datagen._lsr(datagen.splitmix64_t(0xC2, i), 33)), abi.INT64),
        "f64_unit": (torch.rand(n, dtype=torch.float64, device=dev, generator=g), abi.FLOAT64)}
val = datagen.fill_chunks(torch.empty(n, dtype=torch.float64, device=dev), lambda i: datagen.val_t(0xF2, i))
torch.cuda.synchronize()
KEYS = [k for k in "ABCD" if os.environ.get(f"LIB_{k}")]
bes = {k: abi.Backend(os.environ[f"LIB_{k}"], "sqlrs_", 0) for k in KEYS}
val = val
val2 = datagen.fill_chunks(torch.empty(n, dtype=torch.int64, device=dev), lambda i: datagen._lsr(datagen.splitmix64_t(0xD7, i), 40))
k31 = keys["i64_31bit"][0]
klo = k31 & 0xFFFF          # two keys: (k >> 16, k & 0xffff) orders like k
khi = k31 >> 16
torch.cuda.synchronize()
def ob(cols):
    packed = [InputRef(c).pack() for c in cols]
    arr = (abi.OrderBy * len(cols))(*[abi.OrderBy(p_.abi, 1, 0) for p_ in packed])
    return arr, packed
# shape -> (columns, dtypes, ORDER BY columns, env hooks)
shapes = {
    "i64_31bit": ([k31, val], [abi.INT64, abi.FLOAT64], [0], {}),
    "i64_31bit_counting_form": ([k31, val], [abi.INT64, abi.FLOAT64], [0], {"SQLRS_ORDER_LB": "0"}),
    "i64_31bit_three_columns": ([k31, val, val2], [abi.INT64, abi.FLOAT64, abi.INT64], [0], {}),
    "two_int64_keys": ([khi, klo, val], [abi.INT64, abi.INT64, abi.FLOAT64], [0, 1], {}),
    "f64_unit": ([keys["f64_unit"][0], val], [abi.FLOAT64, abi.FLOAT64], [0], {}),
}
only = os.environ.get("SHAPES")
def run(be, b, obs, nob):
    h = C.c_void_p()
    be.check(be.fn("order_create")(be.ctx, nob, obs, C.byref(h)))
    be.check(be.fn("order_push_retained")(h, b.ptr))
    o = C.POINTER(abi.Batch)()
    be.check(be.fn("order_finish")(h, abi.MEM_DEVICE, C.byref(o)))
    be.fn("batch_release")(o)
    be.fn("order_destroy")(h)
for name, (cols, dts, obc, env) in shapes.items():
    if only and name not in only.split(","):
        continue
    b = bench.device_batch(abi, cols, dts)
    obs, _keep = ob(obc)
    for k_, v_ in env.items():
        os.environ[k_] = v_
    for rep in range(int(os.environ.get("REPS", 2))):
        for lk in KEYS:
            be = bes[lk]
            run(be, b, obs, len(obc)); be.synchronize()
            be.profile(True)
            t = time.perf_counter()
            for _ in range(5):
                run(be, b, obs, len(obc))
            be.synchronize()
            ms = (time.perf_counter() - t) / 5 * 1e3
            pr = be.profile_read(); be.profile(False)
            print(f"{name} build {lk}: {ms:.3f} ms | " + " ".join(f"{kk} {v[0]/5:.3f}" for kk, v in sorted(pr.items(), key=lambda kv: -kv[1][0])[:4]), flush=True)
    for k_ in env:
        del os.environ[k_]
