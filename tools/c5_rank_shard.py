"""What ONE rank of `bench.py --gpus W` runs after the exchange, emulated on one GPU: the C5 tables are hash-partitioned
W ways with the library's own partition function (sqlrs_hash_partition / sqlrs_hash_partition_filter), partition 0 —
the dim keys and the kept fact rows rank 0 would receive — is collected, and the local HashJoin + HashAgg over it is
timed.  The received keys are a pseudo-random 1/W subset of 0..n_dim: whether they still take the direct-addressed
(dense) routes decides how the per-rank time scales with W.
  python tools/c5_rank_shard.py [W ...]      (default 1 2 4 8; SQLRS_BENCH_ROWS / SQLRS_BENCH_DIM as in bench.py)"""
import os as _os; _os.environ.setdefault("SQLRS_HOOKS", "1")  # the SQLRS_* tuning hooks are consulted only in a process that opts in (common.hpp: hook)
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, sqlrs_amd
from sqlrs_amd import abi, datagen
from sqlrs_amd.expr import Constant, InputRef

dev = torch.device("cuda", 0)
be = sqlrs_amd.new_ctx(0)
n_fact, n_dim = int(float(os.environ.get("SQLRS_BENCH_ROWS", 1e9))), int(float(os.environ.get("SQLRS_BENCH_DIM", 1e7)))
worlds = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
fact_key = datagen.fill_chunks(torch.empty(n_fact, dtype=torch.int64, device=dev), lambda i: datagen.key_t(0xF1, i, n_dim))
fact_val = datagen.fill_chunks(torch.empty(n_fact, dtype=torch.float64, device=dev), lambda i: datagen.val_t(0xF2, i))
dim_key = datagen.fill_chunks(torch.empty(n_dim, dtype=torch.int64, device=dev), lambda i: datagen.dim_key_t(i, n_dim))
torch.cuda.synchronize()  # the library reads the columns on its own stream
pred = InputRef(1) > Constant(0.5, abi.FLOAT64)
T = {abi.INT64: torch.int64, abi.FLOAT64: torch.float64}
res = {}
for W in worlds:
    # partition 0 of the dim keys
    parts, offs = be.hash_partition(bench.device_batch(abi, [dim_key], [abi.INT64]), InputRef(0), W, abi.MEM_DEVICE)
    be.synchronize()  # (the library works on its own stream; torch reads the result on torch's)
    dk = bench._tensor_view(torch, parts.column(0).values, parts.column(0).length, torch.int64, dev)[offs[0]:offs[1]].clone()
    torch.cuda.synchronize()
    parts.release()
    # partition 0 of the kept fact rows, slice by slice as the W source ranks would send them
    fk = torch.empty(int(n_fact * 0.55 / W) + 4096, dtype=torch.int64, device=dev)
    fv = torch.empty_like(fk, dtype=torch.float64)
    got = 0
    for r in range(W * 4):
        lo, hi = n_fact * r // (W * 4), n_fact * (r + 1) // (W * 4)
        b = bench.device_batch(abi, [fact_key[lo:hi], fact_val[lo:hi]], [abi.INT64, abi.FLOAT64])
        parts, starts, rows = be.hash_partition_filter(b, InputRef(0), pred, W, abi.MEM_DEVICE)
        be.synchronize()
        for ci, (dst, dt) in enumerate(((fk, abi.INT64), (fv, abi.FLOAT64))):
            v = bench._tensor_view(torch, parts.column(ci).values, parts.column(ci).length, T[dt], dev)
            dst[got:got + rows[0]] = v[starts[0]:starts[0] + rows[0]]
        got += rows[0]
        torch.cuda.synchronize()
        parts.release()
    fk, fv = fk[:got], fv[:got]
    pipe = bench.Pipeline(be, abi, 0.5)
    def step():
        out = pipe.join_agg(bench.device_batch(abi, [dk], [abi.INT64]), bench.device_batch(abi, [fk, fv], [abi.INT64, abi.FLOAT64]))
        be.synchronize()
        return out
    print(f"W={W}: {dk.numel():,} dim keys, {got:,} kept fact rows received", file=sys.stderr, flush=True)
    out = step()
    groups = out.num_rows
    rng = (int(dk.min()), int(dk.max()), int(fk.min()), int(fk.max()))
    assert rng[0] >= 0 and rng[1] < n_dim and rng[2] >= 0 and rng[3] < n_dim, rng
    # check against torch on the same received rows
    exp_cnt = torch.bincount(fk, minlength=n_dim)
    has = torch.zeros(n_dim, dtype=torch.bool, device=dev)
    has[dk] = True
    ok = bench._tensor_view(torch, out.column(0).values, groups, torch.int64, dev)
    cnt = bench._tensor_view(torch, out.column(1).values, groups, torch.int64, dev)
    good = bool((exp_cnt[ok] == cnt).all().item()) and groups == int(((exp_cnt > 0) & has).sum().item())
    out.release()
    for _ in range(3):
        step().release()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 8
    for _ in range(K):
        step().release()
    ms = (time.perf_counter() - t0) * 1e3 / K
    be.profile(True)
    step().release()
    prof = {k: round(v[0], 3) for k, v in be.profile_read().items() if v[0] > 0.02}
    be.profile(False)
    res[W] = {"dim_rows": int(dk.numel()), "fact_rows_kept": got, "groups": groups, "counts_match": good, "join_agg_ms": round(ms, 3),
              "fused_batches": int(pipe.fused_batches), "ms_times_W": round(ms * W, 2), "kernel_classes": prof}
    print(W, json.dumps(res[W]), flush=True)
    del fk, fv, dk
