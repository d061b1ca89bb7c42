"""C3 with sparse keys (1e8 probe x 1e6 build, all hit): probe time + kernel classes, with an in-process A/B of a per-call
hook (VAR / VALUES), e.g.  VAR=SQLRS_LDS_JOIN VALUES=0,-1  or  VAR=SQLRS_LJ_RPI VALUES=1,8."""
import os as _os; _os.environ.setdefault("SQLRS_HOOKS", "1")  # the SQLRS_* tuning hooks are consulted only in a process that opts in (common.hpp: hook)
import os, sys, time, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import sqlrs_amd
from sqlrs_amd import abi, datagen
from sqlrs_amd.expr import InputRef
from bench import device_batch
dev = torch.device("cuda", 0)
be = abi.Backend(os.environ["LIB"], "sqlrs_", 0) if os.environ.get("LIB") else sqlrs_amd.new_ctx(0)
nP, nB = int(float(os.environ.get("NP", 1e8))), int(float(os.environ.get("NB", 1e6)))
A_s = 0x9E3779B97F4A7C15 - (1 << 64)
dk = datagen.fill_chunks(torch.empty(nB, dtype=torch.int64, device=dev), lambda i: datagen.dim_key_t(i, nB)) * A_s + 12345
fk = datagen.fill_chunks(torch.empty(nP, dtype=torch.int64, device=dev), lambda i: datagen.key_t(0xF1, i, nB)).mul_(A_s).add_(12345)
torch.cuda.synchronize()
db, fb = device_batch(abi, [dk], [abi.INT64]), device_batch(abi, [fk], [abi.INT64])
lk, _k1 = abi.pack_exprs([InputRef(0)]); rk, _k2 = abi.pack_exprs([InputRef(0)])
rd = (C.c_int32 * 1)(abi.INT64)
VAR = os.environ.get("VAR", "SQLRS_LDS_JOIN")
for rep in range(int(os.environ.get("REPS", 2))):
    for val in os.environ.get("VALUES", "-1").split(","):
        os.environ[VAR] = val
        j = C.c_void_p()
        be.check(be.fn("hash_join_create")(be.ctx, abi.JOIN_INNER, 1, lk, rk, None, 1, rd, C.byref(j)))
        be.check(be.fn("hash_join_build_push")(j, db.ptr)); be.check(be.fn("hash_join_build_finish")(j))
        def probe():
            o = C.POINTER(abi.Batch)()
            be.check(be.fn("hash_join_probe_indices")(j, fb.ptr, abi.MEM_DEVICE, C.byref(o)))
            m = o.contents.num_rows; be.fn("batch_release")(o); return m
        for _ in range(2): m = probe()
        be.synchronize(); t = time.perf_counter()
        for _ in range(5): probe()
        be.synchronize(); ms = (time.perf_counter() - t) * 200
        be.profile(True); probe(); pr = be.profile_read(); be.profile(False)
        be.fn("hash_join_destroy")(j)
        print(f"{VAR}={val:>4}: pairs {m}  probe {ms:6.3f} ms | " + " ".join(f"{k} {v[0]:.3f}" for k, v in sorted(pr.items(), key=lambda kv: -kv[1][0]) if v[0] > 0.01), flush=True)
# BOTH=1: build + probe timed and profiled together (what the bench leg's ms_build_probe covers)
if os.environ.get("BOTH") == "1":
    def both():
        j = C.c_void_p()
        be.check(be.fn("hash_join_create")(be.ctx, abi.JOIN_INNER, 1, lk, rk, None, 1, rd, C.byref(j)))
        be.check(be.fn("hash_join_build_push")(j, db.ptr)); be.check(be.fn("hash_join_build_finish")(j))
        o = C.POINTER(abi.Batch)()
        be.check(be.fn("hash_join_probe_indices")(j, fb.ptr, abi.MEM_DEVICE, C.byref(o)))
        be.fn("batch_release")(o); be.fn("hash_join_destroy")(j)
    for _ in range(3): both()
    be.synchronize(); t = time.perf_counter()
    for _ in range(5): both()
    be.synchronize(); ms = (time.perf_counter() - t) * 200
    be.profile(True); both(); pr = be.profile_read(); be.profile(False)
    print(f"build + probe {ms:6.3f} ms | " + " ".join(f"{k} {v[0]:.3f}" for k, v in sorted(pr.items(), key=lambda kv: -kv[1][0]) if v[0] > 0.002), flush=True)
