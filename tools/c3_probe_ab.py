"""C3 probe (1e8 x 1e6, all hit) in ONE process under several settings of a per-call hook: VAR=<env name> VALUES=a,b,c.
python tools/c3_probe_ab.py"""
import os as _os; _os.environ.setdefault("SQLRS_HOOKS", "1")  # the SQLRS_* tuning hooks are consulted only in a process that opts in (common.hpp: hook)
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, sqlrs_amd
from sqlrs_amd import abi, datagen
from sqlrs_amd.expr import InputRef
dev = torch.device("cuda", 0)
be = sqlrs_amd.new_ctx(0)
nP, nB = 100_000_000, 1_000_000
mod = int(os.environ.get("MOD", nB))
dk = datagen.fill_chunks(torch.empty(nB, dtype=torch.int64, device=dev), lambda i: datagen.dim_key_t(i, nB))
fk = datagen.fill_chunks(torch.empty(nP, dtype=torch.int64, device=dev), lambda i: datagen.key_t(0xF1, i, mod))
torch.cuda.synchronize()
db, fb = bench.device_batch(abi, [dk], [abi.INT64]), bench.device_batch(abi, [fk], [abi.INT64])
lk, _k1 = abi.pack_exprs([InputRef(0)]); rk, _k2 = abi.pack_exprs([InputRef(0)])
rd = (C.c_int32 * 1)(abi.INT64)
j = C.c_void_p()
be.check(be.fn("hash_join_create")(be.ctx, abi.JOIN_INNER, 1, lk, rk, None, 1, rd, C.byref(j)))
be.check(be.fn("hash_join_build_push")(j, db.ptr)); be.check(be.fn("hash_join_build_finish")(j))
def probe():
    o = C.POINTER(abi.Batch)()
    be.check(be.fn("hash_join_probe_indices")(j, fb.ptr, abi.MEM_DEVICE, C.byref(o)))
    be.fn("batch_release")(o)
VAR = os.environ.get("VAR", "SQLRS_PROBE_ALLHIT")
for rep in range(2):
    for val in os.environ.get("VALUES", "1,0").split(","):
        os.environ[VAR] = val
        probe(); probe(); be.synchronize()
        t = C.c_void_p(); be.check(be.fn("timer_create")(be.ctx, C.byref(t)))
        best = 1e9
        for _ in range(7):
            be.check(be.fn("timer_start")(t)); probe(); be.check(be.fn("timer_stop")(t))
            ms = C.c_double(); be.check(be.fn("timer_elapsed_ms")(t, C.byref(ms))); best = min(best, ms.value)
        be.fn("timer_destroy")(t)
        print(f"{VAR}={val}: probe {best:.3f} ms", flush=True)
