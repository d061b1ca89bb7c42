"""Kernel times of C5 steps against WHERE the big buffers sit: REPS rounds in one process, the ctx pool trimmed between
rounds (fresh hipMalloc blocks = fresh placement), SQLRS_RP_TRACE prints the device pointers.  The spread of the partition
kernels between processes (10-20 %) follows placement; this shows whether it follows the VIRTUAL addresses.
    python tools/placement_log.py        (REPS=10)"""
import os as _os; _os.environ.setdefault("SQLRS_HOOKS", "1")  # the SQLRS_* tuning hooks are consulted only in a process that opts in (common.hpp: hook)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SQLRS_RP_TRACE"] = "1"
import torch
import bench, sqlrs_amd
from sqlrs_amd import abi, datagen
dev = torch.device("cuda", 0)
be = sqlrs_amd.new_ctx(0)
n, nd = int(float(os.environ.get("N", 1e9))), int(float(os.environ.get("ND", 1e7)))
fk = datagen.fill_chunks(torch.empty(n, dtype=torch.int64, device=dev), lambda i: datagen.key_t(0xF1, i, nd))
fv = datagen.fill_chunks(torch.empty(n, dtype=torch.float64, device=dev), lambda i: datagen.val_t(0xF2, i))
dk = datagen.fill_chunks(torch.empty(nd, dtype=torch.int64, device=dev), lambda i: datagen.dim_key_t(i, nd))
torch.cuda.synchronize()
pipe = bench.Pipeline(be, abi, 0.5)
def step():
    pipe.step(bench.device_batch(abi, [dk], [abi.INT64]), bench.device_batch(abi, [fk, fv], [abi.INT64, abi.FLOAT64])).release()
for rep in range(int(os.environ.get("REPS", 10))):
    be.synchronize(); t_trim = time.perf_counter()
    be.fn("ctx_pool_trim")(be.ctx)
    t_trim = (time.perf_counter() - t_trim) * 1e3
    if os.environ.get("HOLD"):  # perturb the next placement: keep an odd-sized block alive across the round
        hold = torch.empty((rep * 37 + 11) << 20, dtype=torch.uint8, device=dev)
    t_first = time.perf_counter(); step(); be.synchronize(); t_first = (time.perf_counter() - t_first) * 1e3
    os.environ.pop("SQLRS_RP_TRACE", None)
    be.profile(True)
    t = time.perf_counter()
    for _ in range(4):
        step()
    be.synchronize()
    ms = (time.perf_counter() - t) / 4 * 1e3
    pr = be.profile_read(); be.profile(False)
    os.environ["SQLRS_RP_TRACE"] = "1"
    print(f"round {rep}: trim {t_trim:6.1f} ms, first step {t_first:6.1f} ms, step {ms:6.2f} ms | " + " ".join(f"{k} {v[0]/max(v[1],1):.2f}" for k, v in sorted(pr.items(), key=lambda kv: -kv[1][0])[:3]), flush=True)
