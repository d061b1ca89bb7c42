"""One timing per operator shape that is NOT on the headline configs (join types, duplicate / sparse /
two-column / Utf8 join keys, compound filter predicates, multi-key / DESC / Utf8 sorts, wide aggregate
lists): a pathology detector, not a benchmark.  Inputs are device batches, outputs stay on the device."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, pyarrow as pa
import sqlrs_amd
from sqlrs_amd import abi
from sqlrs_amd.expr import AggFunc, InputRef, Constant, BinaryOp, OrderBy, JoinCondition
from sqlrs_amd.executor import FilterExecutor, HashJoinExecutor, HashAggExecutor, OrderExecutor
be = sqlrs_amd.new_ctx(0)
D = abi.MEM_DEVICE
n = int(float(os.environ.get("N", 2e7)))
rng = np.random.default_rng(7)
def dev(arrays, names=None):
    names = names or [f"c{i}" for i in range(len(arrays))]
    return be.to_device(pa.RecordBatch.from_arrays([pa.array(a) if isinstance(a, np.ndarray) else a for a in arrays], names=names))
def drain(it):
    rows = 0
    for b in it:
        rows += b.num_rows
        b.release()
    return rows
def timed(label, make, rows_in):
    drain(make()); drain(make()); drain(make()); be.synchronize()  # first uses load code objects and grow the pool
    t = time.perf_counter(); out = drain(make()); be.synchronize(); ms = (time.perf_counter() - t) * 1e3
    be.profile(True); drain(make()); pr = be.profile_read(); be.profile(False)
    top = sorted(pr.items(), key=lambda kv: -kv[1][0])[:3]
    print(f"{label:58s} in {rows_in:.1e} out {out:10d}  {ms:8.2f} ms {rows_in/ms/1e6:7.2f} Grows/s   " + ", ".join(f"{a} {v[0]:.2f}" for a, v in top), flush=True)
def jschema(l, r):
    return pa.schema([pa.field(f"l.{i}", f.type) for i, f in enumerate(l.schema)] + [pa.field(f"r.{i}", f.type) for i, f in enumerate(r.schema)])

# ---- filter shapes
a = rng.integers(0, 1000, n, dtype=np.int64); b = rng.random(n); c = rng.integers(0, 100, n).astype(np.int32)
fb = dev([a, b, c])
fbh = pa.RecordBatch.from_arrays([pa.array(a), pa.array(b), pa.array(c)], names=["c0", "c1", "c2"])
for label, e in (("filter c0 > 500 (3 columns carried)", BinaryOp(">", InputRef(0), Constant(500, abi.INT64))),
                 ("filter c0 > 200 AND c1 < 0.5", BinaryOp("and", BinaryOp(">", InputRef(0), Constant(200, abi.INT64)), BinaryOp("<", InputRef(1), Constant(0.5, abi.FLOAT64)))),
                 ("filter c0 + 1 > c2 (arith + cast-free compare)", BinaryOp(">", BinaryOp("+", InputRef(0), Constant(1, abi.INT64)), Constant(50, abi.INT64))),
                 ("filter int32 c2 = 7", BinaryOp("=", InputRef(2), Constant(7, abi.INT32)))):
    timed(label, lambda e=e: FilterExecutor(be, e, [fb], out_mem=D).execute(), n)
fb.release()

# ---- join shapes (probe n rows)
nb = 1_000_000
pk = rng.integers(0, nb, n, dtype=np.int64); pv = rng.random(n)
probe = dev([pk, pv]); probe_h = pa.RecordBatch.from_arrays([pa.array(pk), pa.array(pv)], names=["c0", "c1"])
def join(label, lkeys_arrays, on, jt="inner", pr=probe, prh=probe_h):
    lb = dev(lkeys_arrays); lbh = pa.RecordBatch.from_arrays([pa.array(x) if isinstance(x, np.ndarray) else x for x in lkeys_arrays], names=[f"c{i}" for i in range(len(lkeys_arrays))])
    sch = jschema(lbh, prh)
    timed(label, lambda: HashJoinExecutor(be, [lb], [pr], jt, JoinCondition(on), sch, len(lkeys_arrays), out_mem=D).execute(), n)
    lb.release()
dimk = rng.permutation(nb).astype(np.int64); dimp = (dimk * 3 + 1)
on1 = [(InputRef(0), InputRef(0))]
join("join inner, unique dense keys, payload gathered", [dimk, dimp], on1)
join("join left,  unique dense keys", [dimk, dimp], on1, "left")
join("join right, unique dense keys", [dimk, dimp], on1, "right")
join("join full,  unique dense keys", [dimk, dimp], on1, "full")
join("join inner, unique SPARSE keys (hash table)", [dimk * 1_000_003, dimp], on1, pr=dev([pk * 1_000_003, pv]), prh=probe_h)
dupk = rng.integers(0, nb // 4, nb, dtype=np.int64)
join("join inner, duplicate build keys (x4)", [dupk, dimp], on1, pr=dev([pk // 4, pv]))
join("join inner, two-column key", [dimk % 1000, dimk // 1000, dimp], [(InputRef(0), InputRef(0)), (InputRef(1), InputRef(2))],
     pr=dev([pk % 1000, pv, pk // 1000]), prh=pa.RecordBatch.from_arrays([pa.array(pk), pa.array(pv), pa.array(pk)], names=["c0", "c1", "c2"]))

# ---- aggregate lists
gk = rng.integers(0, 100_000, n, dtype=np.int64); v1 = rng.random(n); v2 = rng.integers(-1000, 1000, n, dtype=np.int64)
ab = dev([gk, v1, v2])
timed("agg 5 aggregates over 2 columns, 1e5 groups", lambda: HashAggExecutor(be, [AggFunc("count", InputRef(1), abi.INT64), AggFunc("sum", InputRef(1), abi.FLOAT64),
      AggFunc("sum", InputRef(2), abi.INT64), AggFunc("min", InputRef(2), abi.INT64), AggFunc("max", InputRef(1), abi.FLOAT64)], [InputRef(0)], [ab], out_mem=D).execute(), n)
timed("agg min+max f64, 1e5 groups", lambda: HashAggExecutor(be, [AggFunc("min", InputRef(1), abi.FLOAT64), AggFunc("max", InputRef(1), abi.FLOAT64)], [InputRef(0)], [ab], out_mem=D).execute(), n)
ab.release()
mask = rng.random(n) < 0.1
abn = dev([pa.array(gk, mask=rng.random(n) < 0.02), pa.array(v1, mask=mask), pa.array(v2)])
timed("agg count+sum, NULLs in key and value, 1e5 groups", lambda: HashAggExecutor(be, [AggFunc("count", InputRef(1), abi.INT64), AggFunc("sum", InputRef(1), abi.FLOAT64)], [InputRef(0)], [abn], out_mem=D).execute(), n)
abn.release()

# ---- order shapes
ob = dev([rng.integers(0, 1 << 40, n, dtype=np.int64), rng.integers(0, 100, n, dtype=np.int64), rng.random(n)])
timed("order by c0 (40-bit int64), 2 payload columns", lambda: OrderExecutor(be, [OrderBy(InputRef(0), True)], [ob], out_mem=D).execute(), n)
timed("order by c1, c0 desc (two keys)", lambda: OrderExecutor(be, [OrderBy(InputRef(1), True), OrderBy(InputRef(0), False)], [ob], out_mem=D).execute(), n)
timed("order by c2 (f64)", lambda: OrderExecutor(be, [OrderBy(InputRef(2), True)], [ob], out_mem=D).execute(), n)
