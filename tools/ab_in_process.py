"""In-process A/B of a tuning hook (VAR=name VALUES=a,b,c): C5 steps with the fused filter, per-kernel
device time from the ctx profile.  Kernel times differ by 10-15 % BETWEEN processes (physical placement
of the buffers) but repeat to 0.2 % inside one, so variants must be compared in one process."""
import os as _os; _os.environ.setdefault("SQLRS_HOOKS", "1")  # the SQLRS_* tuning hooks are consulted only in a process that opts in (common.hpp: hook)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, sqlrs_amd
from sqlrs_amd import abi, datagen
dev = torch.device("cuda", 0)
be = sqlrs_amd.new_ctx(0)
n, nd = int(float(os.environ.get("N", 1e9))), int(float(os.environ.get("ND", 1e7)))
fk = datagen.fill_chunks(torch.empty(n, dtype=torch.int64, device=dev), lambda i: datagen.key_t(0xF1, i, nd))
fv = datagen.fill_chunks(torch.empty(n, dtype=torch.float64, device=dev), lambda i: datagen.val_t(0xF2, i))
dk = datagen.fill_chunks(torch.empty(nd, dtype=torch.int64, device=dev), lambda i: datagen.dim_key_t(i, nd))
if os.environ.get("SPARSE"):  # sparse 64-bit keys (k -> k * odd + c on both sides): hash partition + LDS hash tables
    A_s = 0x9E3779B97F4A7C15 - (1 << 64)
    fk.mul_(A_s).add_(12345); dk.mul_(A_s).add_(12345)
if os.environ.get("SHAPE") == "sorted":   # the fact rows sorted by key (bench.py, c5_variants.adversarial)
    o = torch.sort(fk).indices
    fk, fv = fk[o], fv[o]
    del o
elif os.environ.get("SHAPE") == "hot":    # 30 % of the fact rows carry one key
    fk = torch.where((torch.arange(n, device=dev, dtype=torch.int64) * 0x9E3779B1 % 10) < 3, torch.full_like(fk, int(dk[nd // 3].item())), fk)
torch.cuda.synchronize()
print("ptrs", hex(fk.data_ptr()), hex(fv.data_ptr()))
pipe = bench.Pipeline(be, abi, 0.5, fused=os.environ.get("UNFUSED") != "1")  # UNFUSED=1: Filter, HashJoin, HashAgg as three operators
def step():
    pipe.step(bench.device_batch(abi, [dk], [abi.INT64]), bench.device_batch(abi, [fk, fv], [abi.INT64, abi.FLOAT64])).release()
VAR = os.environ.get("VAR", "SQLRS_RP_CHUNK_TILES")  # a hook the library reads per call
for rep in range(int(os.environ.get("REPS", 3))):
    if os.environ.get("TRIM"):  # fresh pool blocks every round: the variants are compared over several placements
        be.fn("ctx_pool_trim")(be.ctx)
    for st in os.environ.get("VALUES", "1").split(","):
        os.environ[VAR] = st
        step(); be.synchronize()
        be.profile(True)
        t = time.perf_counter()
        for _ in range(5):
            step()
        be.synchronize()
        ms = (time.perf_counter() - t) / 5 * 1e3
        pr = be.profile_read(); be.profile(False)
        print(f"{VAR}={st:>6}: step {ms:6.2f} ms | " + " ".join(f"{k} {v[0]/max(v[1],1):.2f}" for k, v in sorted(pr.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get('TOP', 4))]), flush=True)
