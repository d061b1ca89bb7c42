"""Seeded differential fuzzing of the four operators: random schemas, NULL fractions, key
cardinalities, batch splits and plan parameters, HIP path vs the CPU oracle through the same
Python operator structs.  Every case is reproducible from its seed."""
import numpy as np
import pyarrow as pa
import pytest

from sqlrs_amd import abi
from sqlrs_amd.executor import FilterExecutor, HashAggExecutor, HashJoinExecutor, OrderExecutor
from sqlrs_amd.expr import AggFunc, BinaryOp, Constant, InputRef, JoinCondition, OrderBy

from test_gpu_parity import assert_same, col, join_schema, rows_of

pytestmark = pytest.mark.gpu

import os

# SQLRS_FUZZ_EXTRA=N adds N more seeds per operator for a longer soak (default suite: 40 / 30)
_EXTRA = int(os.environ.get("SQLRS_FUZZ_EXTRA", "0"))

_DT = {"i64": abi.INT64, "f64": abi.FLOAT64, "i32": abi.INT32}


def _split(rng, b):
    """random batch boundaries (including empty batches)"""
    n = b.num_rows
    if n == 0 or rng.random() < 0.3:
        return [b]
    cuts = sorted(set(int(x) for x in rng.integers(0, n + 1, rng.integers(1, 5))))
    edges = [0] + cuts + [n]
    return [b.slice(lo, hi - lo) for lo, hi in zip(edges[:-1], edges[1:])]


def _rand_batch(rng, n, kinds, key_card):
    cols = []
    for i, k in enumerate(kinds):
        nf = float(rng.choice([0.0, 0.0, 0.05, 0.4]))
        if i == 0:
            cols.append(col(rng, n, k, nf, 0, key_card))
        else:
            cols.append(col(rng, n, k, nf, -1000, 1000))
    return pa.RecordBatch.from_arrays(cols, names=[f"c{i}" for i in range(len(cols))])


@pytest.mark.parametrize("seed", range(40 + _EXTRA))
def test_fuzz_filter(hip, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([0, 1, 77, 4096, 20_000, 70_000]))
    kinds = [str(rng.choice(["i64", "f64", "i32"])) for _ in range(int(rng.integers(1, 4)))]
    b = _rand_batch(rng, n, kinds, 50)
    c = int(rng.integers(0, len(kinds)))
    op = str(rng.choice([">", "<", ">=", "<=", "=", "!="]))
    const = Constant(float(rng.normal(0, 300)) if kinds[c] == "f64" else int(rng.integers(-50, 60)), _DT[kinds[c]])
    e = BinaryOp(op, InputRef(c), const)
    if len(kinds) > 1 and rng.random() < 0.5:  # general predicate path
        d = (c + 1) % len(kinds)
        e2 = BinaryOp("<", InputRef(d), Constant(0.0 if kinds[d] == "f64" else 0, _DT[kinds[d]]))
        e = (e & e2) if rng.random() < 0.5 else (e | e2)
    bs = _split(rng, b)
    assert_same(rows_of(FilterExecutor(hip, e, bs).execute()), rows_of(FilterExecutor(oracle, e, bs).execute()))


@pytest.mark.parametrize("seed", range(40 + _EXTRA))
def test_fuzz_hash_agg(hip, oracle, seed):
    rng = np.random.default_rng(2000 + seed)
    n = int(rng.choice([1, 50, 3000, 40_000, 150_000]))
    card = int(rng.choice([1, 7, 300, 20_000]))
    key_kind = str(rng.choice(["i64", "i64", "i32", "f64"]))
    b = _rand_batch(rng, n, [key_kind, "i64", "f64"], card)
    if key_kind == "f64":
        k = np.floor(b.column(0).to_numpy(zero_copy_only=False))
        b = b.set_column(0, "c0", pa.array(k, mask=np.array(b.column(0).is_null())))
    pool = [("count", 1, abi.INT64), ("count", 2, abi.INT64), ("sum", 1, abi.INT64), ("sum", 2, abi.FLOAT64),
            ("min", 1, abi.INT64), ("max", 1, abi.INT64), ("min", 2, abi.FLOAT64), ("max", 2, abi.FLOAT64)]
    picks = [pool[i] for i in rng.choice(len(pool), int(rng.integers(1, 5)), replace=False)]
    aggs = [AggFunc(f, InputRef(c), t) for f, c, t in picks]
    fl = {1 + i for i, (f, c, t) in enumerate(picks) if f == "sum" and t == abi.FLOAT64}
    bs = _split(rng, b)
    assert_same(rows_of(HashAggExecutor(hip, aggs, [InputRef(0)], bs).execute()),
                rows_of(HashAggExecutor(oracle, aggs, [InputRef(0)], bs).execute()), float_cols=fl)


@pytest.mark.parametrize("seed", range(40 + _EXTRA))
def test_fuzz_hash_join(hip, oracle, seed):
    rng = np.random.default_rng(3000 + seed)
    nb = int(rng.choice([0, 1, 40, 900, 6000]))
    npb = int(rng.choice([0, 1, 300, 9000, 40_000]))
    card = int(rng.choice([3, 100, 5000, 20_000]))
    unique = rng.random() < 0.5
    lb = _rand_batch(rng, nb, ["i64", "f64"], card)
    if unique and nb:
        keys = rng.permutation(max(card, nb))[:nb].astype(np.int64)
        lb = lb.set_column(0, "c0", pa.array(keys, mask=(rng.random(nb) < 0.02) if rng.random() < 0.3 else None))
    rb = _rand_batch(rng, npb, ["i64", "i64"], card)
    jt = str(rng.choice(["inner", "left", "right", "full"]))
    filt = (InputRef(1) > Constant(0.0, abi.FLOAT64)) if rng.random() < 0.3 else None
    cond = JoinCondition([(InputRef(0), InputRef(0))], filt)
    sch = join_schema(lb, rb)
    lbs, rbs = _split(rng, lb), _split(rng, rb)
    got = list(HashJoinExecutor(hip, lbs, rbs, jt, cond, sch, 2).execute())
    exp = list(HashJoinExecutor(oracle, lbs, rbs, jt, cond, sch, 2).execute())
    assert [x.num_rows for x in got] == [x.num_rows for x in exp]
    assert_same(rows_of(got), rows_of(exp))


@pytest.mark.parametrize("seed", range(10 + _EXTRA // 10))
def test_fuzz_order_fast_route_shortcuts(hip, oracle, seed, monkeypatch):
    """ORDER BY one plain key of >= 2^20 rows with the shortcuts of round 3 forced at test size: key range from a sample
    (SQLRS_ORDER_SAMPLE=1), rows already in order, and — every other seed — a LimitExecutor above whose offset + limit is
    handed to the sort (SQLRS_ORDER_TOPK=1); random key types, ranges, directions, tie densities, nearly sorted inputs."""
    from sqlrs_amd.executor import LimitExecutor
    monkeypatch.setenv("SQLRS_ORDER_SAMPLE", "1")
    monkeypatch.setenv("SQLRS_ORDER_TOPK", "1")
    rng = np.random.default_rng(4500 + seed)
    n = int(rng.choice([1_100_000, 1_400_000]))
    kind = str(rng.choice(["i64", "i64_ties", "f64", "i32", "i64_offset"]))
    if kind == "i64":
        k = rng.integers(-(1 << 31), 1 << 31, n, dtype=np.int64) >> int(rng.integers(0, 12))
    elif kind == "i64_ties":
        k = rng.integers(0, int(rng.choice([3, 100, 40_000])), n, dtype=np.int64)
    elif kind == "f64":
        k = np.round(rng.standard_normal(n), int(rng.integers(1, 6))) + 0.0  # (+ 0.0: no negative zeros — their order against +0.0 is arrow's business)
    elif kind == "i32":
        k = rng.integers(-(1 << 24), 1 << 24, n).astype(np.int32)
    else:
        k = (1 << 45) + rng.integers(0, 1 << 28, n, dtype=np.int64)
    shape = str(rng.choice(["random", "random", "sorted", "sorted_desc", "nearly_sorted"]))
    if shape != "random":
        k = np.sort(k, kind="stable")
        if shape == "sorted_desc":
            k = k[::-1].copy()
        if shape == "nearly_sorted":
            i = int(rng.integers(1, n - 1))
            k[i], k[i - 1] = k[i - 1], k[i]
    asc = bool(rng.random() < 0.5)
    cols, names = [pa.array(k), pa.array(np.arange(n, dtype=np.int64))], ["k", "row"]
    if rng.random() < 0.5:
        cols.append(pa.array(rng.random(n), mask=rng.random(n) < 0.05))
        names.append("x")
    b = pa.RecordBatch.from_arrays(cols, names=names)
    bs = _split(rng, b)
    ob = [OrderBy(InputRef(0), asc)]
    if seed % 3 == 2:  # a second key: the top-k threshold is taken on the first key only, the fast route steps aside
        ob.append(OrderBy(InputRef(1), bool(rng.random() < 0.5)))
    if seed % 2:
        kk, off = int(rng.choice([1, 100, 20_000, n // 17])), int(rng.choice([0, 0, 13]))
        got = pa.Table.from_batches(list(LimitExecutor(hip, kk, off, OrderExecutor(hip, ob, bs, limit_hint=kk + off).execute()).execute()))
        exp = pa.Table.from_batches(list(LimitExecutor(oracle, kk, off, OrderExecutor(oracle, ob, bs).execute()).execute()))
    else:
        got = pa.Table.from_batches(list(OrderExecutor(hip, ob, bs).execute()))
        exp = pa.Table.from_batches(list(OrderExecutor(oracle, ob, bs).execute()))
    assert got.num_rows == exp.num_rows
    for i in range(len(names)):
        assert got.column(i).combine_chunks().equals(exp.column(i).combine_chunks()), (names[i], kind, shape, asc)


@pytest.mark.parametrize("seed", range(30 + _EXTRA))
def test_fuzz_order(hip, oracle, seed):
    rng = np.random.default_rng(4000 + seed)
    n = int(rng.choice([1, 2, 65, 5000, 70_000]))
    kinds = [str(rng.choice(["i64", "f64", "i32"])) for _ in range(int(rng.integers(1, 4)))]
    b = _rand_batch(rng, n, kinds, int(rng.choice([2, 50, 100_000])))
    b = b.append_column("rowid", pa.array(np.arange(n)))
    nkeys = int(rng.integers(1, len(kinds) + 1))
    ob = [OrderBy(InputRef(int(c)), bool(rng.random() < 0.5)) for c in rng.permutation(len(kinds))[:nkeys]]
    bs = _split(rng, b)
    assert_same(rows_of(OrderExecutor(hip, ob, bs).execute()), rows_of(OrderExecutor(oracle, ob, bs).execute()))


@pytest.mark.parametrize("seed", range(12 + _EXTRA // 10))
def test_fuzz_hash_agg_partition_route(hip, oracle, seed):
    """batches above 2^21 rows: LDS-partitioned pre-aggregation (packed and unpacked rows, NULLs,
    one or two value columns, skewed keys, several batches staged together)"""
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.choice([2_100_000, 2_600_000, 3_300_000]))
    card = int(rng.choice([5, 4000, 200_000, 1_500_000]))
    nulls = float(rng.choice([0.0, 0.0, 0.03]))
    lo = int(rng.choice([0, -card // 2, 1 << 35]))
    keys = rng.integers(lo, lo + card, n, dtype=np.int64)
    if rng.random() < 0.4:  # heavy hitters
        keys[rng.random(n) < 0.5] = lo + int(rng.integers(0, card))
    # stride 1: dense keys (range partition, direct-addressed tables); 3: a third of the range is
    # used (still dense enough); 11: sparse keys (hashed partition, probing tables)
    keys *= int(rng.choice([1, 1, 3, 11]))
    kmask = (rng.random(n) < nulls) if nulls else None
    v1 = pa.array(rng.random(n), mask=(rng.random(n) < nulls) if nulls else None)
    v2 = pa.array(rng.integers(-10**6, 10**6, n, dtype=np.int64), mask=(rng.random(n) < nulls) if nulls else None)
    b = pa.RecordBatch.from_arrays([pa.array(keys, mask=kmask), v1, v2], names=["k", "a", "b"])
    pool = [("count", 1, abi.INT64), ("sum", 1, abi.FLOAT64), ("min", 1, abi.FLOAT64), ("max", 1, abi.FLOAT64),
            ("count", 2, abi.INT64), ("sum", 2, abi.INT64), ("min", 2, abi.INT64), ("max", 2, abi.INT64)]
    two_cols = rng.random() < 0.5
    cand = pool if two_cols else pool[:4]
    picks = [cand[i] for i in rng.choice(len(cand), int(rng.integers(1, 4)), replace=False)]
    if seed % 5 == 4:  # a wide aggregate list: three argument columns (the key column is the third), run as one-column parts
        picks = [pool[i] for i in rng.choice(len(pool), int(rng.integers(4, 8)), replace=False)] + [("max", 0, abi.INT64), ("count", 0, abi.INT64)]
    aggs = [AggFunc(f, InputRef(c), t) for f, c, t in picks]
    fl = {1 + i for i, (f, c, t) in enumerate(picks) if f == "sum" and t == abi.FLOAT64}
    bs = [b] if rng.random() < 0.5 else [b.slice(0, n // 3), b.slice(n // 3)]
    assert_same(rows_of(HashAggExecutor(hip, aggs, [InputRef(0)], bs).execute()),
                rows_of(HashAggExecutor(oracle, aggs, [InputRef(0)], bs).execute()), float_cols=fl)


@pytest.mark.parametrize("seed", range(10 + _EXTRA // 10))
def test_fuzz_join_agg(hip, oracle, seed):
    """HashJoinAgg (fused where it applies, composed otherwise) vs HashAgg(HashJoin) on the oracle"""
    from sqlrs_amd.executor import HashJoinAggExecutor
    rng = np.random.default_rng(6000 + seed)
    nb = int(rng.choice([50, 3000, 40_000]))
    npb = int(rng.choice([70_000, 400_000, 2_300_000]))
    card = int(nb * rng.choice([1, 2, 10]))
    unique = rng.random() < 0.7
    lkeys = rng.permutation(card)[:nb].astype(np.int64) if unique else rng.integers(0, card, nb, dtype=np.int64)
    lb = pa.RecordBatch.from_arrays([pa.array(lkeys), pa.array(rng.integers(0, 9, nb, dtype=np.int64))], names=["k", "x"])
    nulls = float(rng.choice([0.0, 0.04]))
    pk = rng.integers(0, card, npb, dtype=np.int64)
    if rng.random() < 0.4:  # heavy hitters: skewed buckets are split and merged through the per-bucket global tables
        pk[rng.random(npb) < 0.6] = lkeys[int(rng.integers(0, nb))] if rng.random() < 0.7 else card + 5
        pk[rng.random(npb) < 0.1] = lkeys[int(rng.integers(0, nb))]
    rb = pa.RecordBatch.from_arrays([pa.array(pk, mask=(rng.random(npb) < nulls) if nulls else None),
                                     pa.array(rng.random(npb), mask=(rng.random(npb) < nulls) if nulls else None)], names=["k", "v"])
    cond = JoinCondition([(InputRef(0), InputRef(0))])
    sch = join_schema(lb, rb)
    aggs = [AggFunc("count", InputRef(3), abi.INT64), AggFunc("sum", InputRef(3), abi.FLOAT64)]
    if rng.random() < 0.3:
        aggs.append(AggFunc("sum", InputRef(1), abi.INT64))  # build-side argument: composed route
    gb = [InputRef(0)] if rng.random() < 0.5 else [InputRef(2)]
    if seed % 4 == 3:  # GROUP BY a build-side payload column (eager aggregation when the build keys are unique), or key + payload
        gb = [InputRef(1)] if rng.random() < 0.6 else [InputRef(0), InputRef(1)]
    rbs = _split(rng, rb)
    got = rows_of(HashJoinAggExecutor(hip, [lb], rbs, cond, sch, 2, aggs, gb).execute())
    join = HashJoinExecutor(oracle, [lb], rbs, "inner", cond, sch, 2)
    exp = rows_of(HashAggExecutor(oracle, aggs, gb, join.execute()).execute())
    assert_same(got, exp, float_cols={len(gb) + 1})


@pytest.mark.parametrize("seed", range(8 + _EXTRA // 20))
def test_fuzz_hash_agg_partition_route_composite_keys(hip, oracle, seed):
    """large batches whose group key is not one plain int64 column: two key columns (hash-only
    matching like the reference), f64 keys, i32 keys, a key expression"""
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([2_100_000, 2_500_000]))
    kind = str(rng.choice(["two_cols", "f64", "i32", "expr"]))
    card = int(rng.choice([50, 3000, 100_000]))
    nulls = float(rng.choice([0.0, 0.03]))
    m = lambda: (rng.random(n) < nulls) if nulls else None
    k1 = pa.array(rng.integers(0, card, n, dtype=np.int64), mask=m())
    k2 = pa.array(rng.integers(0, 7, n, dtype=np.int32), mask=m())
    kf = pa.array(np.floor(rng.random(n) * card) - card / 2, mask=m())
    v = pa.array(rng.random(n), mask=m())
    b = pa.RecordBatch.from_arrays([k1, k2, kf, v], names=["k1", "k2", "kf", "v"])
    gb = {"two_cols": [InputRef(0), InputRef(1)], "f64": [InputRef(2)], "i32": [InputRef(1)],
          "expr": [InputRef(0) + Constant(5, abi.INT64)]}[kind]
    aggs = [AggFunc("count", InputRef(3), abi.INT64), AggFunc("sum", InputRef(3), abi.FLOAT64)]
    if rng.random() < 0.5:
        aggs.append(AggFunc("max", InputRef(3), abi.FLOAT64))
    bs = [b] if rng.random() < 0.5 else [b.slice(0, n // 2), b.slice(n // 2)]
    nk = len(gb)
    assert_same(rows_of(HashAggExecutor(hip, aggs, gb, bs).execute()),
                rows_of(HashAggExecutor(oracle, aggs, gb, bs).execute()), float_cols={nk + 1})


@pytest.mark.parametrize("seed", range(8 + _EXTRA // 20))
def test_fuzz_filter_and_order_large(hip, oracle, seed):
    rng = np.random.default_rng(8000 + seed)
    n = int(rng.choice([200_000, 1_000_003]))
    nulls = float(rng.choice([0.0, 0.1]))
    a = pa.array(rng.integers(-(1 << int(rng.choice([8, 20, 40]))), 1 << int(rng.choice([8, 20, 40])), n, dtype=np.int64),
                 mask=(rng.random(n) < nulls) if nulls else None)
    f = pa.array(rng.normal(0, 100, n), mask=(rng.random(n) < nulls) if nulls else None)
    b = pa.RecordBatch.from_arrays([a, f, pa.array(np.arange(n))], names=["a", "f", "i"])
    thr = float(rng.normal(0, 80))
    e = BinaryOp(str(rng.choice([">", "<=", "!="])), InputRef(1), Constant(thr, abi.FLOAT64))
    bs = _split(rng, b)
    assert_same(rows_of(FilterExecutor(hip, e, bs).execute()), rows_of(FilterExecutor(oracle, e, bs).execute()))
    ob = [OrderBy(InputRef(0), bool(rng.random() < 0.5))]
    if rng.random() < 0.5:
        ob.append(OrderBy(InputRef(1), bool(rng.random() < 0.5)))
    sub = b.slice(0, 300_000)
    assert_same(rows_of(OrderExecutor(hip, ob, [sub]).execute()), rows_of(OrderExecutor(oracle, ob, [sub]).execute()))


@pytest.mark.parametrize("seed", range(6 + _EXTRA // 20))
def test_fuzz_hash_agg_distinct_and_utf8_keys(hip, oracle, seed):
    """DISTINCT aggregates (de-duplicating sub-aggregation) and Utf8 group keys (hash-only matching),
    medium to large inputs, several batches"""
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice([5_000, 120_000, 2_150_000]))
    card = int(rng.choice([3, 200, 20_000]))
    nulls = float(rng.choice([0.0, 0.05]))
    m = lambda: (rng.random(n) < nulls) if nulls else None
    ik = pa.array(rng.integers(0, card, n, dtype=np.int64), mask=m())
    sk = pa.array([f"k{v % 977}" for v in rng.integers(0, card, n)], type=pa.string(), mask=m()) if n <= 120_000 else None
    v = pa.array(rng.integers(0, 40, n, dtype=np.int64), mask=m())
    f = pa.array(np.round(rng.random(n) * 8), mask=m())
    use_utf8 = sk is not None and rng.random() < 0.5
    b = pa.RecordBatch.from_arrays([sk if use_utf8 else ik, v, f], names=["k", "v", "f"])
    aggs = [AggFunc("count", InputRef(1), abi.INT64, distinct=True), AggFunc("sum", InputRef(1), abi.INT64, distinct=True),
            AggFunc("count", InputRef(2), abi.INT64), AggFunc("sum", InputRef(2), abi.FLOAT64, distinct=bool(rng.random() < 0.5))]
    aggs = [aggs[i] for i in sorted(rng.choice(4, int(rng.integers(1, 5)), replace=False))]
    bs = _split(rng, b)
    fl = {1 + i for i, a in enumerate(aggs) if a.return_type == abi.FLOAT64}
    assert_same(rows_of(HashAggExecutor(hip, aggs, [InputRef(0)], bs).execute()),
                rows_of(HashAggExecutor(oracle, aggs, [InputRef(0)], bs).execute()), float_cols=fl)


@pytest.mark.gpu
@pytest.mark.hooked_rerun("early_flushes")
def test_fuzz_hash_agg_with_early_flushes():
    """SQLRS_STAGE_FLUSH_ROWS=600000: staged batches are aggregated several times before finish, so
    groups of an earlier flush are merged with later ones (deferred groups -> table, table growth);
    the large-batch fuzz families run under it (hook read once per process, hence the subprocess)."""
    from conftest import hooked_rerun
    hooked_rerun("early_flushes")
