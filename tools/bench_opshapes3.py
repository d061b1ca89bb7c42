"""Third pathology sweep: fused join+aggregate variants, big build sides, multi-batch probes, exchange partition."""
import sys, os, time, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, pyarrow as pa
import sqlrs_amd
from sqlrs_amd import abi
from sqlrs_amd.expr import AggFunc, InputRef, Constant, BinaryOp, JoinCondition
from sqlrs_amd.executor import HashJoinExecutor, HashAggExecutor, HashJoinAggExecutor
be = sqlrs_amd.new_ctx(0)
D = abi.MEM_DEVICE
n = int(float(os.environ.get("N", 2e7)))
rng = np.random.default_rng(5)
def dev(arrays):
    return be.to_device(pa.RecordBatch.from_arrays([pa.array(a) if isinstance(a, np.ndarray) else a for a in arrays], names=[f"c{i}" for i in range(len(arrays))]))
def drain(it):
    rows = 0
    for b in it:
        rows += b.num_rows
        b.release()
    return rows
def timed(label, make, rows_in):
    drain(make()); drain(make()); drain(make()); be.synchronize()
    t = time.perf_counter(); out = drain(make()); be.synchronize(); ms = (time.perf_counter() - t) * 1e3
    be.profile(True); drain(make()); pr = be.profile_read(); be.profile(False)
    top = sorted(pr.items(), key=lambda kv: -kv[1][0])[:4]
    print(f"{label:60s} in {rows_in:.1e} out {out:9d} {ms:9.2f} ms {rows_in/ms/1e6:7.2f} Grows/s   " + ", ".join(f"{a} {v[0]:.2f}" for a, v in top), flush=True)
sch = pa.schema([pa.field("l.0", pa.int64()), pa.field("l.1", pa.int64()), pa.field("r.0", pa.int64()), pa.field("r.1", pa.float64())])
on = JoinCondition([(InputRef(0), InputRef(0))])
aggs = [AggFunc("count", InputRef(3), abi.INT64), AggFunc("sum", InputRef(3), abi.FLOAT64)]
nb = 1_000_000
dimk = rng.permutation(nb).astype(np.int64)
pk = rng.integers(0, nb, n, dtype=np.int64); pv = rng.random(n)
for label, lk, rk, gb in (("join_agg dense unique keys, group by join key (fused)", dimk, pk, [InputRef(0)]),
                          ("join_agg sparse unique keys (fused, hashed buckets)", dimk * 1_000_003, pk * 1_000_003, [InputRef(0)]),
                          ("join_agg group by build payload (composed)", dimk, pk, [InputRef(1)]),
                          ("join_agg duplicate build keys (composed)", dimk // 2, pk // 2, [InputRef(0)])):
    lb, rb = dev([lk, (lk % 1000)]), dev([rk, pv])
    timed(label, lambda: HashJoinAggExecutor(be, [lb], [rb], on, sch, 2, aggs, gb, out_mem=D).execute(), n)
    lb.release(); rb.release()
# probe in 16 batches
lb = dev([dimk, dimk % 1000])
parts = [dev([pk[i::16].copy(), pv[i::16].copy()]) for i in range(16)]
timed("join_agg fused, probe pushed as 16 batches", lambda: HashJoinAggExecutor(be, [lb], parts, on, sch, 2, aggs, [InputRef(0)], out_mem=D).execute(), n)
timed("join inner, probe pushed as 16 batches", lambda: HashJoinExecutor(be, [lb], parts, "inner", on, sch, 2, out_mem=D).execute(), n)
lb.release()
# big build, small probe
bigk = rng.permutation(n).astype(np.int64)
lb = dev([bigk, bigk % 7]); rb = dev([rng.integers(0, n, 1_000_000, dtype=np.int64), rng.random(1_000_000)])
timed("join inner, build 2e7 rows, probe 1e6 rows", lambda: HashJoinExecutor(be, [lb], [rb], "inner", on, sch, 2, out_mem=D).execute(), n)
lb.release(); rb.release()
# exchange partition, general path (NULLs) and fast path
from bench import device_batch
hp = be.fn("hash_partition")
for label, arrays in (("hash_partition 8 ways, 3 x int64 (fast path)", [rng.integers(0, 1 << 40, n, dtype=np.int64), rng.integers(0, 9, n, dtype=np.int64), rng.random(n)]),
                      ("hash_partition 8 ways, NULLs + int32 (general path)", [pa.array(rng.integers(0, 1 << 40, n, dtype=np.int64), mask=rng.random(n) < 0.02), pa.array(rng.integers(0, 9, n).astype(np.int32)), rng.random(n)])):
    b = dev(arrays)
    kx, _k = abi.pack_exprs([InputRef(0)])
    def run():
        o = C.POINTER(abi.Batch)(); cnt = (C.c_int64 * 9)()
        be.check(hp(be.ctx, b.ptr, kx, 8, D, C.byref(o), cnt)); r = o.contents.num_rows; be.fn("batch_release")(o); return r
    for _ in range(3): run()
    be.synchronize(); t = time.perf_counter(); run(); be.synchronize(); ms = (time.perf_counter() - t) * 1e3
    print(f"{label:60s} in {n:.1e} {ms:9.2f} ms {n/ms/1e6:7.2f} Grows/s", flush=True)
    b.release()
