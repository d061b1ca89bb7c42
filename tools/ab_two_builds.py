"""C5 steps alternating between two builds of the library (LIB_A, LIB_B) in ONE process on the same input tensors:
kernel times differ by 10-15 % between processes (physical placement), so compile-time variants are compared here."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sqlrs_amd import abi, datagen
dev = torch.device("cuda", 0)
n, nd = int(float(os.environ.get("N", 1e9))), int(float(os.environ.get("ND", 1e7)))
fk = datagen.fill_chunks(torch.empty(n, dtype=torch.int64, device=dev), lambda i: datagen.key_t(0xF1, i, nd))
fv = datagen.fill_chunks(torch.empty(n, dtype=torch.float64, device=dev), lambda i: datagen.val_t(0xF2, i))
dk = datagen.fill_chunks(torch.empty(nd, dtype=torch.int64, device=dev), lambda i: datagen.dim_key_t(i, nd))
if os.environ.get("SHAPE") == "sorted":   # the fact rows sorted by key (bench.py, c5_variants.adversarial)
    o = torch.sort(fk).indices
    fk, fv = fk[o], fv[o]
    del o
torch.cuda.synchronize()
KEYS = [k for k in "ABCDEF" if os.environ.get(f"LIB_{k}")]  # LIB_A, LIB_B[, LIB_C ...]
bes = {k: abi.Backend(os.environ[f"LIB_{k}"], "sqlrs_", 0) for k in KEYS}
pipes = {k: bench.Pipeline(be, abi, 0.5) for k, be in bes.items()}
def step(k):
    pipes[k].step(bench.device_batch(abi, [dk], [abi.INT64]), bench.device_batch(abi, [fk, fv], [abi.INT64, abi.FLOAT64])).release()
for rep in range(int(os.environ.get("REPS", 3))):
    for k in KEYS:
        be = bes[k]
        step(k); be.synchronize()
        be.profile(True)
        t = time.perf_counter()
        for _ in range(5):
            step(k)
        be.synchronize()
        ms = (time.perf_counter() - t) / 5 * 1e3
        pr = be.profile_read(); be.profile(False)
        print(f"build {k}: step {ms:6.2f} ms | " + " ".join(f"{kk} {v[0]/max(v[1],1):.2f}" for kk, v in sorted(pr.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get("TOP", 4))]), flush=True)
        be.fn("ctx_pool_trim")(be.ctx)
