"""Kernel classes of ONE step of the sparse-key C5 variant (keys k -> k * A + B: hashed LDS buckets, general join
table deferred) — where bench.py's `c5_variants.sparse_keys` spends its time.  python tools/c5_sparse_profile.py"""
import os as _os; _os.environ.setdefault("SQLRS_HOOKS", "1")  # the SQLRS_* tuning hooks are consulted only in a process that opts in (common.hpp: hook)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, sqlrs_amd
from sqlrs_amd import abi, datagen
dev = torch.device("cuda", 0)
be = abi.Backend(os.environ["LIB"], "sqlrs_", 0) if os.environ.get("LIB") else sqlrs_amd.new_ctx(0)
n_fact, n_dim = int(float(os.environ.get("SQLRS_BENCH_ROWS", 1e9))), int(float(os.environ.get("SQLRS_BENCH_DIM", 1e7)))
fk = datagen.fill_chunks(torch.empty(n_fact, dtype=torch.int64, device=dev), lambda i: datagen.key_t(0xF1, i, n_dim))
fv = datagen.fill_chunks(torch.empty(n_fact, dtype=torch.float64, device=dev), lambda i: datagen.val_t(0xF2, i))
dk = datagen.fill_chunks(torch.empty(n_dim, dtype=torch.int64, device=dev), lambda i: datagen.dim_key_t(i, n_dim))
A_s, B_s = 0x9E3779B97F4A7C15 - (1 << 64), 0x632BE59BD9B4E019
for t in (fk, dk):
    for lo in range(0, t.numel(), 1 << 27):
        t[lo:lo + (1 << 27)].mul_(A_s).add_(B_s)
torch.cuda.synchronize()
pipe = bench.Pipeline(be, abi, 0.5)
def step():
    out = pipe.step(bench.device_batch(abi, [dk], [abi.INT64]), bench.device_batch(abi, [fk, fv], [abi.INT64, abi.FLOAT64]))
    be.synchronize()
    return out
for _ in range(2):
    step().release()
t0 = time.perf_counter()
for _ in range(3):
    step().release()
ms = (time.perf_counter() - t0) * 1e3 / 3
be.profile(True)
step().release()
pr = be.profile_read()
be.profile(False)
print(f"sparse-key C5: {ms:.2f} ms/step, fused batches {pipe.fused_batches}")
for k, v in sorted(pr.items(), key=lambda kv: -kv[1][0]):
    if v[0] > 0.01:
        print(f"  {k:28s} {v[0]:8.3f} ms  x{v[1]}")
