"""Second pathology sweep: NULLs, Utf8, int32 keys, DISTINCT, many small batches (see bench_opshapes.py)."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, pyarrow as pa
import sqlrs_amd
from sqlrs_amd import abi
from sqlrs_amd.expr import AggFunc, InputRef, Constant, BinaryOp, OrderBy, JoinCondition
from sqlrs_amd.executor import FilterExecutor, HashJoinExecutor, HashAggExecutor, OrderExecutor
be = sqlrs_amd.new_ctx(0)
D = abi.MEM_DEVICE
n = int(float(os.environ.get("N", 1e7)))
rng = np.random.default_rng(9)
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
    top = sorted(pr.items(), key=lambda kv: -kv[1][0])[:3]
    print(f"{label:56s} in {rows_in:.1e} out {out:9d} {ms:9.2f} ms {rows_in/ms/1e6:7.2f} Grows/s   " + ", ".join(f"{a} {v[0]:.2f}" for a, v in top), flush=True)
def utf8(vals, idx, mask=None):
    d = pa.DictionaryArray.from_arrays(pa.array(idx.astype(np.int32), mask=mask), pa.array(vals)).cast(pa.string())
    return d.combine_chunks() if isinstance(d, pa.ChunkedArray) else d
states = ["CA", "CO", "NY", "TX", "WA", "Colorado State", "California State", "", "zz", "abcdefghij"]
m5 = rng.random(n) < 0.05
# ---- filters
fb = dev([pa.array(rng.random(n), mask=m5), pa.array(rng.integers(0, 100, n).astype(np.int32), mask=m5), utf8(states, rng.integers(0, 10, n), m5), pa.array(rng.integers(0, 9, n, dtype=np.int64))])
timed("filter f64 > 0.5, 5% NULLs, 4 columns", lambda: FilterExecutor(be, BinaryOp(">", InputRef(0), Constant(0.5, abi.FLOAT64)), [fb], out_mem=D).execute(), n)
timed("filter int32 < 50, 5% NULLs", lambda: FilterExecutor(be, BinaryOp("<", InputRef(1), Constant(50, abi.INT32)), [fb], out_mem=D).execute(), n)
timed("filter utf8 = 'CA'", lambda: FilterExecutor(be, BinaryOp("=", InputRef(2), Constant("CA", abi.UTF8)), [fb], out_mem=D).execute(), n)
timed("filter keeps nothing (c3 > 100)", lambda: FilterExecutor(be, BinaryOp(">", InputRef(3), Constant(100, abi.INT64)), [fb], out_mem=D).execute(), n)
timed("filter keeps everything (c3 >= 0)", lambda: FilterExecutor(be, BinaryOp(">=", InputRef(3), Constant(0, abi.INT64)), [fb], out_mem=D).execute(), n)
fb.release()
small = [dev([rng.integers(0, 1000, 10_000, dtype=np.int64), rng.random(10_000)]) for _ in range(200)]
timed("filter 200 batches of 1e4 rows", lambda: FilterExecutor(be, BinaryOp(">", InputRef(0), Constant(500, abi.INT64)), small, out_mem=D).execute(), 2_000_000)
timed("agg    200 batches of 1e4 rows, 1000 groups", lambda: HashAggExecutor(be, [AggFunc("count", InputRef(1), abi.INT64), AggFunc("sum", InputRef(1), abi.FLOAT64)], [InputRef(0)], small, out_mem=D).execute(), 2_000_000)
# ---- joins
nb = 500_000
def jschema(lt, rt):
    return pa.schema([pa.field(f"l.{i}", t) for i, t in enumerate(lt)] + [pa.field(f"r.{i}", t) for i, t in enumerate(rt)])
dk = rng.permutation(nb)
pk = rng.integers(0, nb, n)
for label, lcols, rcols, lt, rt in (
    ("join inner, int32 keys", [dk.astype(np.int32), dk.astype(np.int64)], [pk.astype(np.int32), rng.random(n)], [pa.int32(), pa.int64()], [pa.int32(), pa.float64()]),
    ("join inner, int64 keys, 5% NULL probe keys", [dk.astype(np.int64), dk.astype(np.int64)], [pa.array(pk.astype(np.int64), mask=m5), rng.random(n)], [pa.int64(), pa.int64()], [pa.int64(), pa.float64()]),
    ("join inner, utf8 keys (5e5 distinct strings)", [pa.array([f"k{x}" for x in dk]), dk.astype(np.int64)], [pa.array([f"k{x}" for x in pk[: n // 10]]), rng.random(n // 10)], [pa.string(), pa.int64()], [pa.string(), pa.float64()]),
):
    lb, rb = dev(lcols), dev(rcols)
    rows = rb.num_rows
    timed(label, lambda: HashJoinExecutor(be, [lb], [rb], "inner", JoinCondition([(InputRef(0), InputRef(0))]), jschema(lt, rt), 2, out_mem=D).execute(), rows)
    lb.release(); rb.release()
# ---- aggregates
ab = dev([rng.integers(0, 100_000, n, dtype=np.int64), rng.integers(0, 1000, n, dtype=np.int64), utf8(states, rng.integers(0, 10, n)), rng.integers(0, 100, n).astype(np.int32)])
timed("agg count(distinct c1), 1e5 groups", lambda: HashAggExecutor(be, [AggFunc("count", InputRef(1), abi.INT64, distinct=True)], [InputRef(0)], [ab], out_mem=D).execute(), n)
timed("agg min(utf8), max(utf8), 1e5 groups", lambda: HashAggExecutor(be, [AggFunc("min", InputRef(2), abi.UTF8), AggFunc("max", InputRef(2), abi.UTF8)], [InputRef(0)], [ab], out_mem=D).execute(), n)
timed("agg group by utf8, int64 (two-column key)", lambda: HashAggExecutor(be, [AggFunc("count", InputRef(1), abi.INT64)], [InputRef(2), InputRef(3)], [ab], out_mem=D).execute(), n)
ab.release()
for label, keys in (("agg mostly distinct dense int64 keys (8e6 of 1e7)", rng.integers(0, 20_000_000, n, dtype=np.int64)),
                    ("agg mostly distinct sparse int64 keys", rng.integers(0, 20_000_000, n, dtype=np.int64) * 1_000_003),
                    ("agg all-distinct keys (permutation)", rng.permutation(n).astype(np.int64))):
    hb = dev([keys, rng.random(n)])
    timed(label, lambda: HashAggExecutor(be, [AggFunc("count", InputRef(1), abi.INT64), AggFunc("sum", InputRef(1), abi.FLOAT64)], [InputRef(0)], [hb], out_mem=D).execute(), n)
    hb.release()
# ---- order
ob = dev([utf8(states, rng.integers(0, 10, n // 5)), pa.array(rng.integers(0, 1000, n // 5, dtype=np.int64), mask=m5[: n // 5]), rng.random(n // 5)])
timed("order by utf8 (10 distinct)", lambda: OrderExecutor(be, [OrderBy(InputRef(0), True)], [ob], out_mem=D).execute(), n // 5)
timed("order by int64 with 5% NULLs desc", lambda: OrderExecutor(be, [OrderBy(InputRef(1), False)], [ob], out_mem=D).execute(), n // 5)
