"""bench.py's C3_join_two_column_keys leg alone (1e8 probe x 1e6 build rows, ON a.x = b.x AND a.y = b.y matched by the combined
hash like the reference): build + probe time and kernel classes."""
import os as _os; _os.environ.setdefault("SQLRS_HOOKS", "1")
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, sqlrs_amd
from sqlrs_amd import abi, datagen
from sqlrs_amd.expr import InputRef
dev = torch.device("cuda", 0); be = sqlrs_amd.new_ctx(0)
nP, nB = 100_000_000, 1_000_000
dim_key = datagen.fill_chunks(torch.empty(nB, dtype=torch.int64, device=dev), lambda i: datagen.dim_key_t(i, nB))
pk = datagen.fill_chunks(torch.empty(nP, dtype=torch.int64, device=dev), lambda i: datagen.key_t(0xF1, i, nB))
bx, by_ = dim_key % 1000, dim_key // 1000 + 10_000
px, py = pk % 1000, pk // 1000 + 10_000
torch.cuda.synchronize()
db, fb = bench.device_batch(abi, [bx, by_], [abi.INT64] * 2), bench.device_batch(abi, [px, py], [abi.INT64] * 2)
lk, _k1 = abi.pack_exprs([InputRef(0), InputRef(1)]); rk, _k2 = abi.pack_exprs([InputRef(0), InputRef(1)])
rd = (C.c_int32 * 2)(abi.INT64, abi.INT64)
m = [0]
def both():
    j = C.c_void_p()
    be.check(be.fn("hash_join_create")(be.ctx, abi.JOIN_INNER, 2, lk, rk, None, 2, rd, C.byref(j)))
    be.check(be.fn("hash_join_build_push")(j, db.ptr)); be.check(be.fn("hash_join_build_finish")(j))
    o = C.POINTER(abi.Batch)()
    be.check(be.fn("hash_join_probe_indices")(j, fb.ptr, abi.MEM_DEVICE, C.byref(o)))
    m[0] = o.contents.num_rows
    be.fn("batch_release")(o); be.fn("hash_join_destroy")(j)
for _ in range(3): both()
be.synchronize(); t = time.perf_counter()
for _ in range(5): both()
be.synchronize(); ms = (time.perf_counter() - t) * 200
be.profile(True); both(); pr = be.profile_read(); be.profile(False)
print(f"two-column keys: {m[0]} pairs, build + probe {ms:.3f} ms | " + " ".join(f"{k} {v[0]:.3f}" for k, v in sorted(pr.items(), key=lambda kv: -kv[1][0]) if v[0] > 0.005), flush=True)
