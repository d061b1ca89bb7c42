"""C3-style probe (1e8 int64 probe keys, Inner, index pairs) against dense build sides of several sizes:
how much of the probe time is the table's cache footprint (4 B per build key)."""
import sys, os, time, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import sqlrs_amd
from sqlrs_amd import abi
from sqlrs_amd.expr import InputRef
from bench import device_batch
dev = torch.device("cuda", 0); be = sqlrs_amd.new_ctx(0)
nP = 100_000_000
D = abi.MEM_DEVICE
only = int(os.environ.get("PROBE_SWEEP_ONLY", "0"))
for nB in ((only,) if only else (100_000, 250_000, 500_000, 1_000_000, 2_000_000, 4_000_000, 16_000_000)):
    for hit in (1.0, 0.5, 0.0):
        dim_key = torch.randperm(nB, device=dev, dtype=torch.int64)
        hi = nB if hit == 1.0 else (2 * nB if hit == 0.5 else nB)
        fk = torch.randint(0, hi, (nP,), device=dev, dtype=torch.int64)
        if hit == 0.0:
            fk += nB
        torch.cuda.synchronize()
        db, fb = device_batch(abi, [dim_key], [abi.INT64]), device_batch(abi, [fk], [abi.INT64])
        lk, _k1 = abi.pack_exprs([InputRef(0)]); rk, _k2 = abi.pack_exprs([InputRef(0)])
        rd = (C.c_int32 * 1)(abi.INT64); j = C.c_void_p()
        be.check(be.fn("hash_join_create")(be.ctx, abi.JOIN_INNER, 1, lk, rk, None, 1, rd, C.byref(j)))
        be.check(be.fn("hash_join_build_push")(j, db.ptr)); be.check(be.fn("hash_join_build_finish")(j))
        def probe():
            o = C.POINTER(abi.Batch)()
            be.check(be.fn("hash_join_probe_indices")(j, fb.ptr, D, C.byref(o)))
            m = o.contents.num_rows; be.fn("batch_release")(o); return m
        for _ in range(3): m = probe()
        be.synchronize(); t = time.perf_counter()
        for _ in range(10): probe()
        be.synchronize(); ms = (time.perf_counter() - t) * 100
        be.fn("hash_join_destroy")(j)
        print(f"build {nB:9d} ({4*nB/2**20:6.1f} MB table) hit {hit:3.1f}: pairs {m:9d}  probe {ms:6.3f} ms  {nP/ms/1e6:6.1f} Grows/s", flush=True)
        del fk, dim_key
