"""Operator-shape sweeps: one timing per operator shape that is NOT on the headline configs — a pathology detector,
not a benchmark.  Inputs are device batches, outputs stay on the device.
  python tools/bench_opshapes.py [1] [2] [3] [4]      (no argument: all four; N=<rows> overrides a sweep's size)
  1  join types, duplicate / sparse / two-column / Utf8 join keys, compound predicates, multi-key / DESC / Utf8 sorts, wide aggregate lists
  2  NULLs, Utf8 filters / keys / sorts, int32 join keys, DISTINCT, mostly distinct group keys, many small batches
  3  fused join+aggregate variants, probes pushed as 16 batches, big build side, exchange partition fast / general path
  4  ORDER BY over long / all-distinct Utf8 keys, many payload columns, 64 input batches, 1000 rows"""
import ctypes as C
import os, sys, time
import numpy as np, pyarrow as pa
from opshapes_common import be, D, abi, dev, drain, timed
from sqlrs_amd.expr import AggFunc, InputRef, Constant, BinaryOp, OrderBy, JoinCondition
from sqlrs_amd.executor import FilterExecutor, HashJoinExecutor, HashAggExecutor, HashJoinAggExecutor, OrderExecutor


def sweep1():
    """One timing per operator shape that is NOT on the headline configs (join types, duplicate / sparse / two-column / Utf8 join keys, compound filter predicates, multi-key / DESC / Utf8 sorts, wide aggregate lists): a pathology detector, not a benchmark.  Inputs are device batches, outputs stay on the device."""
    n = int(float(os.environ.get("N", 2e7)))
    rng = np.random.default_rng(7)
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


def sweep2():
    """Second pathology sweep: NULLs, Utf8, int32 keys, DISTINCT, many small batches (see bench_opshapes.py)."""
    n = int(float(os.environ.get("N", 1e7)))
    rng = np.random.default_rng(9)
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


def sweep3():
    """Third pathology sweep: fused join+aggregate variants, big build sides, multi-batch probes, exchange partition."""
    n = int(float(os.environ.get("N", 2e7)))
    rng = np.random.default_rng(5)
    sch = pa.schema([pa.field("l.0", pa.int64()), pa.field("l.1", pa.int64()), pa.field("r.0", pa.int64()), pa.field("r.1", pa.float64())])
    on = JoinCondition([(InputRef(0), InputRef(0))])
    aggs = [AggFunc("count", InputRef(3), abi.INT64), AggFunc("sum", InputRef(3), abi.FLOAT64)]
    nb = 1_000_000
    dimk = rng.permutation(nb).astype(np.int64)
    pk = rng.integers(0, nb, n, dtype=np.int64); pv = rng.random(n)
    for label, lk, rk, gb in (("join_agg dense unique keys, group by join key (fused)", dimk, pk, [InputRef(0)]),
                              ("join_agg sparse unique keys (fused, hashed buckets)", dimk * 1_000_003, pk * 1_000_003, [InputRef(0)]),
                              ("join_agg group by build payload (eager aggregation)", dimk, pk, [InputRef(1)]),
                              ("join_agg duplicate build keys (fused, multiplicities)", dimk // 2, pk // 2, [InputRef(0)])):
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


def sweep4():
    """Fourth pathology sweep: ORDER BY shapes (long / many distinct Utf8 keys, many payload columns, many input batches)."""
    n = int(float(os.environ.get("N", 2e6)))
    rng = np.random.default_rng(3)
    words = np.array(["".join(chr(97 + int(c)) for c in rng.integers(0, 26, int(rng.integers(3, 24)))) for _ in range(200_000)])
    s1 = pa.array(words[rng.integers(0, len(words), n)].tolist())
    s2 = pa.array([f"customer#{x:09d}" for x in rng.integers(0, 10**9, n)])
    ob = dev([s1, s2, rng.integers(0, 1 << 30, n, dtype=np.int64), rng.random(n), rng.random(n), rng.integers(0, 9, n, dtype=np.int64), rng.random(n)])
    timed("order by utf8, 2e5 distinct words of 3-23 bytes", lambda: OrderExecutor(be, [OrderBy(InputRef(0), True)], [ob], out_mem=D).execute(), n)
    timed("order by utf8 'customer#%09d' (18 bytes, all distinct)", lambda: OrderExecutor(be, [OrderBy(InputRef(1), True)], [ob], out_mem=D).execute(), n)
    timed("order by int64, 6 payload columns (2 utf8)", lambda: OrderExecutor(be, [OrderBy(InputRef(2), True)], [ob], out_mem=D).execute(), n)
    ob.release()
    parts = [dev([rng.integers(0, 1 << 30, n // 64, dtype=np.int64), rng.random(n // 64)]) for _ in range(64)]
    timed("order by int64, input pushed as 64 batches", lambda: OrderExecutor(be, [OrderBy(InputRef(0), True)], parts, out_mem=D).execute(), n // 64 * 64)
    small = dev([rng.integers(0, 100, 1000, dtype=np.int64), rng.random(1000)])
    timed("order by int64, 1000 rows", lambda: OrderExecutor(be, [OrderBy(InputRef(0), False)], [small], out_mem=D).execute(), 1000)



if __name__ == "__main__":
    which = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]
    for k in which:
        print(f"--- sweep {k}", flush=True)
        {1: sweep1, 2: sweep2, 3: sweep3, 4: sweep4}[k]()
