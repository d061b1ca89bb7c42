"""HashAgg over other key shapes than one int64 column: two-column keys, Utf8 keys, few groups
(time per batch; correctness of these shapes is covered by the parity tests and the fuzz)."""
import sys, os, time, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, pyarrow as pa, torch
import sqlrs_amd
from sqlrs_amd import abi
from sqlrs_amd.expr import AggFunc, InputRef
be = sqlrs_amd.new_ctx(0)
n = int(float(os.environ.get("N", 5e7)))
rng = np.random.default_rng(1)
val = pa.array(rng.random(n))
states = np.array(["CA", "CO", "NY", "TX", "WA", "Colorado State", "California State", "", "zz", "abcdefghij"])
shapes = {
    "int64, 1e6 groups": ([pa.array(rng.integers(0, 1_000_000, n, dtype=np.int64))], [InputRef(0)]),
    "int64, 50 groups": ([pa.array(rng.integers(0, 50, n, dtype=np.int64))], [InputRef(0)]),
    "int64 sparse (x1000003), 1e6 groups": ([pa.array(rng.integers(0, 1_000_000, n, dtype=np.int64) * 1000003)], [InputRef(0)]),
    "(int64, int64), 1e6 groups": ([pa.array(rng.integers(0, 1000, n, dtype=np.int64)), pa.array(rng.integers(0, 1000, n, dtype=np.int64))], [InputRef(0), InputRef(1)]),
    "int32, 1e5 groups": ([pa.array(rng.integers(0, 100_000, n, dtype=np.int32))], [InputRef(0)]),
    "utf8 state, 10 groups": ([pa.DictionaryArray.from_arrays(pa.array(rng.integers(0, len(states), n, dtype=np.int32)), pa.array(states.tolist())).cast(pa.string())], [InputRef(0)]),
}
for name, (kcols, gb) in shapes.items():
    kcols = [k.combine_chunks() if isinstance(k, pa.ChunkedArray) else k for k in kcols]
    b = pa.RecordBatch.from_arrays(kcols + [val], names=[f"k{i}" for i in range(len(kcols))] + ["v"])
    dev = be.to_device(b)
    vi = len(kcols)
    keep = []
    aggs = (abi.AggFunc * 2)(AggFunc("count", InputRef(vi), abi.INT64).abi_struct(keep), AggFunc("sum", InputRef(vi), abi.FLOAT64).abi_struct(keep))
    gbx, _k = abi.pack_exprs(gb)
    def run():
        a = C.c_void_p(); be.check(be.fn("hash_agg_create")(be.ctx, len(gb), gbx, 2, aggs, C.byref(a)))
        be.check(be.fn("hash_agg_push")(a, dev.ptr)); o = C.POINTER(abi.Batch)()
        be.check(be.fn("hash_agg_finish")(a, abi.MEM_DEVICE, C.byref(o))); g = o.contents.num_rows
        be.fn("batch_release")(o); be.fn("hash_agg_destroy")(a); return g
    g = run(); be.synchronize()
    t = time.perf_counter(); run(); run(); be.synchronize(); ms = (time.perf_counter() - t) * 500
    be.profile(True); run(); pr = be.profile_read(); be.profile(False)
    top = sorted(pr.items(), key=lambda kv: -kv[1][0])[:4]
    print(f"{name:38s} rows {n:.0e} groups {g:8d}  {ms:8.2f} ms  {n/ms/1e6:7.2f} Grows/s   " + ", ".join(f"{a} {v[0]:.2f}" for a, v in top), flush=True)
    dev.release()
