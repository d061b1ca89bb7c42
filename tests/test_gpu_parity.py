"""Parity of the HIP path against the CPU oracle on seeded random inputs (MI355X, through
the C ABI).  Integer / index / order results must be bit-exact; SUM(double) within 1e-9
relative (north_star tolerance)."""
import math
import os

import numpy as np
import pyarrow as pa
import pytest

from sqlrs_amd import abi
from sqlrs_amd.executor import (FilterExecutor, HashAggExecutor, HashJoinExecutor, OrderExecutor,
                                eval_column)
from sqlrs_amd.expr import AggFunc, BinaryOp, Constant, InputRef, JoinCondition, OrderBy, TypeCast

pytestmark = pytest.mark.gpu

SIZES = [0, 1, 63, 64, 65, 4095, 4096, 4097, 100_003]
REL_TOL = 1e-9


def col(rng, n, kind, null_frac=0.0, lo=-50, hi=50):
    if kind == "i64":
        v = rng.integers(lo, hi, n, dtype=np.int64)
        t = pa.int64()
    elif kind == "i32":
        v = rng.integers(lo, hi, n, dtype=np.int32)
        t = pa.int32()
    elif kind == "f64":
        v = rng.random(n) * (hi - lo) + lo
        t = pa.float64()
    elif kind == "bool":
        v = rng.integers(0, 2, n).astype(bool)
        t = pa.bool_()
    else:
        raise ValueError(kind)
    mask = rng.random(n) < null_frac if null_frac else None
    return pa.array(v, type=t, mask=mask)


def batch(rng, n, spec):
    arrays = [col(rng, n, k, nf, *rest) for (k, nf, *rest) in spec]
    return pa.RecordBatch.from_arrays(arrays, names=[f"c{i}" for i in range(len(arrays))])


def rows_of(batches):
    out = []
    for b in batches:
        cols = [b.column(i).to_pylist() for i in range(b.num_columns)]
        out.extend(zip(*cols) if cols else [])
    return [tuple(r) for r in out]


def table_of(batches):
    """the batches concatenated into one pyarrow Table, for `assert_same_table` (tests with millions of groups: row tuples
    in Python were most of their run time)"""
    batches = list(batches)
    return pa.Table.from_batches(batches) if batches else None


def assert_same_table(got, exp, float_cols=()):
    if got is None or exp is None:
        assert (got is None or got.num_rows == 0) and (exp is None or exp.num_rows == 0)
        return
    assert got.num_rows == exp.num_rows, f"{got.num_rows} rows, expected {exp.num_rows}"
    assert got.num_columns == exp.num_columns
    for i in range(got.num_columns):
        g, e = got.column(i).combine_chunks(), exp.column(i).combine_chunks()
        if i in float_cols:
            assert g.is_null().equals(e.is_null()), i
            a, b = g.fill_null(0).to_numpy(zero_copy_only=False), e.fill_null(0).to_numpy(zero_copy_only=False)
            bad = np.abs(a - b) > REL_TOL * np.maximum(np.abs(b), 1e-300)
            assert not bad.any(), (i, int(bad.argmax()), a[bad.argmax()], b[bad.argmax()])  # SUM(double): 1e-9 relative
        else:
            assert g.equals(e), i


def assert_same(got, exp, float_cols=()):
    assert len(got) == len(exp)
    for g, e in zip(got, exp):
        assert len(g) == len(e)
        for i, (x, y) in enumerate(zip(g, e)):
            if i in float_cols and x is not None and y is not None:
                assert math.isclose(x, y, rel_tol=REL_TOL, abs_tol=1e-300), (g, e)
            else:
                assert x == y or (isinstance(x, float) and isinstance(y, float) and math.isnan(x) and math.isnan(y)), (g, e)


# ------------------------------------------------------------------------- filter --
@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("kind,const", [("i64", 3), ("f64", 0.5), ("i32", -7)])
@pytest.mark.parametrize("op", [">", "<", ">=", "<=", "=", "!="])
def test_filter_cmp_const(hip, oracle, n, kind, const, op):
    rng = np.random.default_rng(n * 7 + len(op))
    b = batch(rng, n, [(kind, 0.1), ("i64", 0.2), ("f64", 0.0), ("bool", 0.3), ("i32", 0.0)])
    dt = {"i64": abi.INT64, "f64": abi.FLOAT64, "i32": abi.INT32}[kind]
    e = BinaryOp(op, InputRef(0), Constant(const, dt))
    got = rows_of(FilterExecutor(hip, e, [b]).execute())
    exp = rows_of(FilterExecutor(oracle, e, [b]).execute())
    assert_same(got, exp)


@pytest.mark.parametrize("shape", ["two_terms", "three_terms_q6", "four_terms", "five_terms_general", "null_column_general", "or_general", "nothing_passes"])
def test_filter_conjunction_of_comparisons(hip, oracle, shape):
    """`a > x AND b < y [AND ...]` over int64 / float64 columns without NULLs: one kernel writes the selection mask of up to
    four terms (filter_conj_mask_kernel); more terms, a nullable column or an OR take the expression evaluator.  Same rows in
    the same order as the oracle either way; the class of the run says which route it was."""
    rng = np.random.default_rng(len(shape))
    n = 300_000
    nulls = 0.05 if shape == "null_column_general" else 0.0
    b = batch(rng, n, [("i64", 0.0, 0, 1000), ("f64", nulls, 0, 1), ("i64", 0.0, -50, 50), ("f64", 0.0, 0, 1), ("i32", 0.1, 0, 9)])
    t = [BinaryOp(">", InputRef(0), Constant(200, abi.INT64)), BinaryOp("<", InputRef(1), Constant(0.6, abi.FLOAT64)),
         BinaryOp(">=", InputRef(2), Constant(-20, abi.INT64)), BinaryOp("!=", InputRef(3), Constant(0.5, abi.FLOAT64)),
         BinaryOp("<=", InputRef(0), Constant(900, abi.INT64))]
    k = {"two_terms": 2, "three_terms_q6": 3, "four_terms": 4, "five_terms_general": 5, "null_column_general": 2, "or_general": 2,
         "nothing_passes": 2}[shape]
    e = t[0]
    for x in t[1:k]:
        e = BinaryOp("or" if shape == "or_general" else "and", e, x)
    if shape == "nothing_passes":
        e = BinaryOp("and", BinaryOp(">", InputRef(0), Constant(5000, abi.INT64)), t[1])
    hip.profile(True)
    got = table_of(FilterExecutor(hip, e, [b]).execute())
    prof = hip.profile_read()
    hip.profile(False)
    exp = table_of(FilterExecutor(oracle, e, [b]).execute())
    assert_same_table(got, exp)
    assert (prof.get("filter_conj_mask", (0, 0))[1] > 0) == (not shape.endswith("general")), prof


@pytest.mark.parametrize("n", [0, 1, 64, 1000, 70_001])
def test_filter_general_expr(hip, oracle, n):
    rng = np.random.default_rng(n)
    b = batch(rng, n, [("i64", 0.1), ("i64", 0.1), ("f64", 0.05), ("bool", 0.2), ("i32", 0.1)])
    exprs = [
        (InputRef(0) > InputRef(1)) & (InputRef(2) < Constant(10.0, abi.FLOAT64)),
        (InputRef(0) + InputRef(1) > Constant(5, abi.INT64)) | InputRef(3),
        BinaryOp("=", TypeCast(InputRef(4), abi.INT64), InputRef(0)),
        BinaryOp("!=", InputRef(0) * Constant(2, abi.INT64) - InputRef(1), Constant(None, abi.INT64)),
        InputRef(3) & Constant(None, abi.BOOLEAN),
        BinaryOp(">=", TypeCast(InputRef(0), abi.FLOAT64), InputRef(2)),
    ]
    for e in exprs:
        got = rows_of(FilterExecutor(hip, e, [b]).execute())
        exp = rows_of(FilterExecutor(oracle, e, [b]).execute())
        assert_same(got, exp)
        if n:
            g = eval_column(hip, e, b).column(0).to_pylist()
            x = eval_column(oracle, e, b).column(0).to_pylist()
            assert g == x


def test_filter_device_resident_chain(hip, oracle):
    rng = np.random.default_rng(5)
    b = batch(rng, 50_000, [("i64", 0.1), ("f64", 0.0)])
    e1 = InputRef(0) > Constant(-10, abi.INT64)
    e2 = InputRef(1) < Constant(20.0, abi.FLOAT64)
    dev = hip.to_device(b)
    mid = list(FilterExecutor(hip, e1, [dev], out_mem=abi.MEM_DEVICE).execute())
    got = rows_of(FilterExecutor(hip, e2, mid).execute())
    mid_o = list(FilterExecutor(oracle, e1, [b]).execute())
    exp = rows_of(FilterExecutor(oracle, e2, mid_o).execute())
    assert_same(got, exp)


def test_divide_by_zero_is_arrow_error(hip, oracle):
    b = pa.RecordBatch.from_arrays([pa.array([4, 6], pa.int64()), pa.array([2, 0], pa.int64())], names=["a", "b"])
    from sqlrs_amd import ExecutorError
    for be in (hip, oracle):
        with pytest.raises(ExecutorError) as ei:
            eval_column(be, InputRef(0) / InputRef(1), b)
        assert ei.value.status == abi.ERR_ARROW


# --------------------------------------------------------------------------- join --
def join_schema(lb, rb):
    return pa.schema([pa.field(f"l.{f.name}", f.type) for f in lb.schema] +
                     [pa.field(f"r.{f.name}", f.type) for f in rb.schema])


@pytest.mark.parametrize("jt", ["inner", "left", "right", "full"])
@pytest.mark.parametrize("nb,np_,keyrange,nulls", [
    (0, 10, 5, 0.0), (10, 0, 5, 0.0), (1, 1, 2, 0.0), (100, 1000, 50, 0.1), (1000, 5000, 2000, 0.05),
    (5000, 70_001, 5000, 0.0), (3000, 20_000, 100, 0.02)])
@pytest.mark.parametrize("with_filter", [False, True])
def test_hash_join(hip, oracle, jt, nb, np_, keyrange, nulls, with_filter):
    rng = np.random.default_rng(nb * 31 + np_)
    lb = batch(rng, nb, [("i64", nulls, 0, keyrange), ("i64", 0.1), ("f64", 0.0)])
    rb = batch(rng, np_, [("f64", 0.05), ("i64", nulls, 0, keyrange), ("i64", 0.0)])
    filt = (InputRef(1) > InputRef(5)) if with_filter else None
    cond = JoinCondition([(InputRef(0), InputRef(1))], filt)
    sch = join_schema(lb, rb)
    # build side in two batches, probe side in three
    lbs = [lb.slice(0, nb // 2), lb.slice(nb // 2)] if nb > 1 else [lb]
    rbs = [rb.slice(0, np_ // 3), rb.slice(np_ // 3, np_ // 3), rb.slice(2 * (np_ // 3))] if np_ > 3 else [rb]
    got = list(HashJoinExecutor(hip, lbs, rbs, jt, cond, sch, lb.num_columns).execute())
    exp = list(HashJoinExecutor(oracle, lbs, rbs, jt, cond, sch, lb.num_columns).execute())
    assert [b.num_rows for b in got] == [b.num_rows for b in exp]  # one batch per probe batch + tail
    assert_same(rows_of(got), rows_of(exp))


@pytest.mark.parametrize("jt", ["inner", "right"])
def test_hash_join_indices(hip, oracle, jt):
    rng = np.random.default_rng(11)
    lb = batch(rng, 2000, [("i64", 0.05, 0, 300)])
    rb = batch(rng, 9000, [("i64", 0.05, 0, 400)])
    cond = JoinCondition([(InputRef(0), InputRef(0))])
    sch = join_schema(lb, rb)
    got = list(HashJoinExecutor(hip, [lb], [rb], jt, cond, sch, 1).execute(indices_only=True))
    exp = list(HashJoinExecutor(oracle, [lb], [rb], jt, cond, sch, 1).execute(indices_only=True))
    assert_same(rows_of(got), rows_of(exp))


@pytest.mark.parametrize("shape", ["all_hit", "miss_in_the_last_row", "miss_in_the_first_row", "one_row_in_ten_misses", "hook_off",
                                   "null_build_key_and_max_plus_one"])
def test_hash_join_dense_probe_all_hit_attempt(hip, oracle, shape, monkeypatch):
    """Direct-address probe, Inner join, unique build keys: the optimistic all-hit kernel (pair i = (build row, i), no
    compaction) runs first and the compacting kernel only redoes the batch when a probe row had no partner.  Index pairs
    and joined batches against the oracle; a single miss anywhere must send the batch through the compacting kernel
    (one launch of the class per kernel: 1 + 1, of which the second is a no-op while the flag is clear)."""
    if shape == "hook_off":
        monkeypatch.setenv("SQLRS_PROBE_ALLHIT", "0")
    rng = np.random.default_rng(len(shape))
    nb, npr = 50_000, 400_000
    bk = rng.permutation(nb).astype(np.int64) + 1000
    pk = rng.integers(1000, 1000 + nb, npr, dtype=np.int64)
    if shape == "miss_in_the_last_row":
        pk[-1] = 5
    elif shape == "miss_in_the_first_row":
        pk[0] = 1000 + nb + 7
    elif shape == "one_row_in_ten_misses":
        pk[rng.random(npr) < 0.1] = -3
    bka = pa.array(bk)
    if shape == "null_build_key_and_max_plus_one":
        # one NULL build key: its row sits in the table slot right behind the key range.  A non-NULL probe key equal to
        # max build key + 1 addresses exactly that slot and must NOT match (a NULL key only matches a NULL key,
        # hash_utils.rs:91-104); every other probe row matches, so the all-hit attempt is what runs.
        mask = np.zeros(nb, dtype=bool)
        mask[17] = True  # (that row's key value disappears from the build side: no probe row may ask for it)
        gone = bk[17]
        bka = pa.array(bk, mask=mask)
        pk[pk == gone] = bk[18]
        pk[12345] = bk.max() + 1
        pk[npr - 2] = bk.max() + 1
    lb = pa.RecordBatch.from_arrays([bka, pa.array(bk * 3 + 1)], names=["k", "p"])
    rb = pa.RecordBatch.from_arrays([pa.array(pk), pa.array(rng.random(npr))], names=["k", "v"])
    cond = JoinCondition([(InputRef(0), InputRef(0))])
    sch = join_schema(lb, rb)
    for indices_only in (True, False):
        got = table_of(HashJoinExecutor(hip, [lb], [rb], "inner", cond, sch, 2).execute(indices_only=indices_only))
        exp = table_of(HashJoinExecutor(oracle, [lb], [rb], "inner", cond, sch, 2).execute(indices_only=indices_only))
        assert_same_table(got, exp)


@pytest.mark.parametrize("jt", ["inner", "left"])
@pytest.mark.parametrize("nb,npr,forced", [(700, 9_000, True), (40_000, 1_200_000, True), (300_000, 4_400_000, False)])
def test_hash_join_lds_tables_general_keys(hip, oracle, monkeypatch, jt, nb, npr, forced):
    """General (sparse 64-bit, f64) unique build keys probed through LDS tables over a blocked partition
    (lds_join_probe_kernel): probed in partitioned order, un-permuted per 2^15-row range through LDS, compacted in probe-row order.  The pairs — and the
    joined batches — must equal the oracle's bit for bit (hash_join.rs:207-292).  `forced`: SQLRS_LDS_JOIN=1 takes
    the route at test sizes; the last case takes it by its own size rule (one probe batch >= 2^22 rows)."""
    if forced:
        monkeypatch.setenv("SQLRS_LDS_JOIN", "1")
    rng = np.random.default_rng(nb + len(jt))
    A = np.int64(0x9E3779B97F4A7C15 - (1 << 64))
    with np.errstate(over="ignore"):
        bk = rng.permutation(nb + nb // 3)[:nb].astype(np.int64) * A + np.int64(77)     # sparse, unique
        pk = rng.integers(0, nb + nb // 2, npr, dtype=np.int64) * A + np.int64(77)      # ~1/3 of the probe keys miss
    lb = pa.RecordBatch.from_arrays([pa.array(bk), pa.array(rng.integers(0, 1000, nb, dtype=np.int64))], names=["c0", "c1"])
    rb = pa.RecordBatch.from_arrays([pa.array(pk), pa.array(rng.random(npr))], names=["c0", "c1"])
    rbs = [rb] if (npr < 100_000 or not forced) else [rb.slice(0, npr // 3), rb.slice(npr // 3)]
    cond = JoinCondition([(InputRef(0), InputRef(0))])
    sch = join_schema(lb, rb)
    hip.profile(True)
    got = list(HashJoinExecutor(hip, [lb], rbs, jt, cond, sch, 2).execute(indices_only=(jt == "inner")))
    prof = hip.profile_read()
    hip.profile(False)
    assert prof.get("join_probe_lds", (0, 0))[1] > 0 and prof.get("join_match_compact", (0, 0))[1] > 0, prof
    exp = list(HashJoinExecutor(oracle, [lb], rbs, jt, cond, sch, 2).execute(indices_only=(jt == "inner")))
    assert [b.num_rows for b in got] == [b.num_rows for b in exp]
    for g, e in zip(got, exp):
        for c in range(g.num_columns):
            assert g.column(c).equals(e.column(c)), c


def test_hash_join_lds_tables_f64_keys_and_fallbacks(hip, oracle, monkeypatch):
    """f64 keys compare by bit pattern on the LDS route too; NULL probe keys keep the global-table route; duplicate build
    keys and Right / Full joins take the LDS tables through the un-permuted {run, pairs} form (same results)."""
    monkeypatch.setenv("SQLRS_LDS_JOIN", "1")
    rng = np.random.default_rng(19)
    nb, npr = 5000, 60_000
    bk = rng.permutation(2 * nb)[:nb].astype(np.float64) * 0.37
    pk = rng.integers(0, 2 * nb, npr).astype(np.float64) * 0.37
    cond = JoinCondition([(InputRef(0), InputRef(0))])
    for variant in ("plain", "probe_nulls", "dup_build", "full"):
        b2, p2 = bk.copy(), pk
        lmask = None
        if variant == "dup_build":
            b2[10] = b2[11]
        lb = pa.RecordBatch.from_arrays([pa.array(b2), pa.array(np.arange(nb, dtype=np.int64))], names=["c0", "c1"])
        rb = pa.RecordBatch.from_arrays([pa.array(p2, mask=(rng.random(npr) < 0.05) if variant == "probe_nulls" else None),
                                         pa.array(rng.random(npr))], names=["c0", "c1"])
        sch = join_schema(lb, rb)
        jt = "full" if variant == "full" else "inner"
        hip.profile(True)
        got = list(HashJoinExecutor(hip, [lb], [rb], jt, cond, sch, 2).execute())
        prof = hip.profile_read()
        hip.profile(False)
        assert (prof.get("join_probe_lds", (0, 0))[1] > 0) == (variant != "probe_nulls"), (variant, prof)
        assert (prof.get("join_match_unpermute", (0, 0))[1] > 0) == (variant in ("dup_build", "full")), (variant, prof)
        exp = list(HashJoinExecutor(oracle, [lb], [rb], jt, cond, sch, 2).execute())
        assert_same(rows_of(got), rows_of(exp))


@pytest.mark.parametrize("jt", ["inner", "left", "right", "full"])
@pytest.mark.parametrize("shape", ["x4", "skewed", "int32_keys", "probe_nulls", "two_build_batches"])
def test_hash_join_duplicate_keys_over_a_dense_range(hip, oracle, jt, shape):
    """Round 6: build keys that REPEAT inside a dense range (a foreign key on the build side) take neither the general table
    nor the LDS tables: runs by key (count per key of the range, scan, rows stably sorted by key) and one direct-address
    {run, rows} lookup per probe row.  Probe keys outside the range, NULL probe keys, int32 keys, a build side in two batches,
    all four join types — batches equal the oracle's bit for bit (hash_join.rs:172-177, 225-248)."""
    rng = np.random.default_rng(len(jt) * 7 + len(shape))
    nb, npr = 50_000, 600_000
    if shape == "skewed":
        bk = rng.integers(100, 100 + nb // 2, nb, dtype=np.int64)
        bk[rng.random(nb) < 0.2] = 123
    else:
        bk = rng.integers(-500, -500 + nb // 4, nb, dtype=np.int64)
    pk = rng.integers(bk.min() - 2000, bk.max() + 2000, npr, dtype=np.int64)
    kt = np.int32 if shape == "int32_keys" else np.int64
    lb = pa.RecordBatch.from_arrays([pa.array(bk.astype(kt)), pa.array(np.arange(nb, dtype=np.int64))], names=["c0", "c1"])
    rb = pa.RecordBatch.from_arrays([pa.array(pk.astype(kt), mask=(rng.random(npr) < 0.1) if shape == "probe_nulls" else None),
                                     pa.array(rng.random(npr))], names=["c0", "c1"])
    lbs = [lb.slice(0, nb // 3), lb.slice(nb // 3)] if shape == "two_build_batches" else [lb]
    rbs = [rb.slice(0, 70_000), rb.slice(70_000, 1), rb.slice(70_001)]
    cond = JoinCondition([(InputRef(0), InputRef(0))])
    sch = join_schema(lb, rb)
    hip.profile(True)
    got = list(HashJoinExecutor(hip, lbs, rbs, jt, cond, sch, 2).execute())
    prof = hip.profile_read()
    hip.profile(False)
    assert prof.get("join_probe_count_dense_dup", (0, 0))[1] > 0 and prof.get("join_build", (0, 0))[1] == 0, prof
    exp = list(HashJoinExecutor(oracle, lbs, rbs, jt, cond, sch, 2).execute())
    assert [b.num_rows for b in got] == [b.num_rows for b in exp]
    for g, e in zip(got, exp):
        for c in range(g.num_columns):
            assert g.column(c).equals(e.column(c)), c


@pytest.mark.parametrize("jt", ["inner", "full"])
@pytest.mark.parametrize("lds", ["1", "0"])
def test_hash_join_per_row_counts_form(hip, oracle, monkeypatch, jt, lds):
    """build sides of >= 2^26 rows keep per-row pair counts + offsets (64 runs of that length overflow a 32-bit group sum): the
    form is forced here (SQLRS_JOIN_GROUPED=0) on both count passes — the LDS route's un-permute and the general table's probe"""
    monkeypatch.setenv("SQLRS_JOIN_GROUPED", "0")
    monkeypatch.setenv("SQLRS_LDS_JOIN", lds)
    rng = np.random.default_rng(5 + len(jt))
    nb, npr = 30_000, 400_001
    bk = rng.integers(0, nb // 3, nb, dtype=np.int64) * np.int64(1_000_003) + np.int64(9)
    pk = rng.integers(0, nb // 2, npr, dtype=np.int64) * np.int64(1_000_003) + np.int64(9)
    lb = pa.RecordBatch.from_arrays([pa.array(bk), pa.array(np.arange(nb, dtype=np.int64))], names=["c0", "c1"])
    rb = pa.RecordBatch.from_arrays([pa.array(pk), pa.array(rng.random(npr))], names=["c0", "c1"])
    cond = JoinCondition([(InputRef(0), InputRef(0))])
    sch = join_schema(lb, rb)
    got = list(HashJoinExecutor(hip, [lb], [rb.slice(0, 100_000), rb.slice(100_000)], jt, cond, sch, 2).execute())
    exp = list(HashJoinExecutor(oracle, [lb], [rb.slice(0, 100_000), rb.slice(100_000)], jt, cond, sch, 2).execute())
    assert [b.num_rows for b in got] == [b.num_rows for b in exp]
    for g, e in zip(got, exp):
        for c in range(g.num_columns):
            assert g.column(c).equals(e.column(c)), c


@pytest.mark.parametrize("jt", ["inner", "left", "right", "full"])
@pytest.mark.parametrize("shape", ["dup_x4", "dup_skewed", "unique", "empty_word_key"])
@pytest.mark.parametrize("nb,npr,forced", [(900, 11_000, True), (60_000, 700_000, True), (280_000, 4_300_000, False)])
def test_hash_join_lds_tables_duplicate_build_keys_and_outer_joins(hip, oracle, monkeypatch, jt, shape, nb, npr, forced):
    """Round 6: DUPLICATE build keys (a probe row emits the run of build rows with its key in build order, hash_join.rs:172-177,
    225-234) and Right / Full joins (an unmatched probe row emits (NULL, row), :235-248) matched on the LDS tables: the
    tables hold the DISTINCT keys of the general table, the range is un-permuted into {run, pairs} per probe row and the
    fill pass expands it.  Batches equal the oracle's bit for bit, over several probe batches."""
    if shape == "unique" and jt in ("inner", "left"):
        pytest.skip("the compacting form: test_hash_join_lds_tables_general_keys")
    if not forced and (shape not in ("dup_x4", "unique") or jt in ("left",)):
        pytest.skip("one large case per route")
    if forced:
        monkeypatch.setenv("SQLRS_LDS_JOIN", "1")
    rng = np.random.default_rng(nb + len(jt) + len(shape))
    A = np.int64(0x9E3779B97F4A7C15 - (1 << 64))
    with np.errstate(over="ignore"):
        if shape == "dup_x4":
            base = rng.permutation(nb)[: nb // 4].astype(np.int64)
            bk = np.concatenate([base] * 4)
            rng.shuffle(bk)
        elif shape == "dup_skewed":  # one key on a tenth of the build rows, the rest unique or doubled
            bk = rng.integers(0, nb, nb, dtype=np.int64)
            bk[rng.random(nb) < 0.1] = 5
        else:
            bk = rng.permutation(nb + nb // 3)[:nb].astype(np.int64)
        bk = bk * A + np.int64(77)
        pk = rng.integers(0, nb + nb // 2, npr, dtype=np.int64) * A + np.int64(77)
        if shape == "empty_word_key":  # the key whose value is the general table's "empty" word, on both sides
            bk[3] = bk[7] = np.int64(-1)
            pk[::97] = np.int64(-1)
    lb = pa.RecordBatch.from_arrays([pa.array(bk), pa.array(rng.integers(0, 1000, nb, dtype=np.int64))], names=["c0", "c1"])
    rb = pa.RecordBatch.from_arrays([pa.array(pk), pa.array(rng.random(npr))], names=["c0", "c1"])
    rbs = [rb] if (npr < 100_000 or not forced) else [rb.slice(0, npr // 3), rb.slice(npr // 3)]
    cond = JoinCondition([(InputRef(0), InputRef(0))])
    sch = join_schema(lb, rb)
    hip.profile(True)
    got = list(HashJoinExecutor(hip, [lb], rbs, jt, cond, sch, 2).execute())
    prof = hip.profile_read()
    hip.profile(False)
    assert prof.get("join_probe_lds", (0, 0))[1] > 0 and prof.get("join_match_unpermute", (0, 0))[1] > 0, prof
    assert prof.get("join_probe_count", (0, 0))[1] == 0, prof
    exp = list(HashJoinExecutor(oracle, [lb], rbs, jt, cond, sch, 2).execute())
    assert [b.num_rows for b in got] == [b.num_rows for b in exp]
    for g, e in zip(got, exp):
        for c in range(g.num_columns):
            assert g.column(c).equals(e.column(c)), c


@pytest.mark.parametrize("first", ["1", "0"])
def test_hash_join_lds_route_builds_no_global_table(hip, oracle, monkeypatch, first):
    """Round 6: a build side that is probed on LDS bucket tables takes `unique` from those tables (lds_join_unique_kernel) and
    never builds the global 16-byte-slot table — until a probe batch that cannot take the route (here: NULL probe keys)
    asks for it.  Same pairs as the oracle either way, and as with SQLRS_LDS_FIRST=0 (the table first, as before)."""
    monkeypatch.setenv("SQLRS_LDS_JOIN", "1")
    monkeypatch.setenv("SQLRS_LDS_FIRST", first)
    rng = np.random.default_rng(23)
    nb, npr = 30_000, 500_000
    A = np.int64(0x9E3779B97F4A7C15 - (1 << 64))
    with np.errstate(over="ignore"):
        bk = rng.permutation(2 * nb)[:nb].astype(np.int64) * A + np.int64(5)
        pk = rng.integers(0, 3 * nb, npr, dtype=np.int64) * A + np.int64(5)
    lb = pa.RecordBatch.from_arrays([pa.array(bk), pa.array(np.arange(nb, dtype=np.int64))], names=["c0", "c1"])
    plain = pa.RecordBatch.from_arrays([pa.array(pk), pa.array(rng.random(npr))], names=["c0", "c1"])
    nulls = pa.RecordBatch.from_arrays([pa.array(pk[:50_000], mask=rng.random(50_000) < 0.1), pa.array(rng.random(50_000))], names=["c0", "c1"])
    cond = JoinCondition([(InputRef(0), InputRef(0))])
    sch = join_schema(lb, plain)
    for rbs, table_builds in (([plain, plain.slice(1000, 200_000)], 0), ([plain, nulls, plain.slice(7, 100_000)], 1)):
        hip.profile(True)
        got = list(HashJoinExecutor(hip, [lb], rbs, "inner", cond, sch, 2).execute())
        prof = hip.profile_read()
        hip.profile(False)
        if first == "1":
            assert prof.get("join_build_lds_unique", (0, 0))[1] == 1, prof
            assert prof.get("join_build", (0, 0))[1] == table_builds, prof
        else:
            assert prof.get("join_build_lds_unique", (0, 0))[1] == 0 and prof.get("join_build", (0, 0))[1] == 1, prof
        assert prof.get("join_probe_lds", (0, 0))[1] == len(rbs) - table_builds, prof
        exp = list(HashJoinExecutor(oracle, [lb], rbs, "inner", cond, sch, 2).execute())
        assert [b.num_rows for b in got] == [b.num_rows for b in exp]
        for g, e in zip(got, exp):
            for c in range(g.num_columns):
                assert g.column(c).equals(e.column(c)), c


@pytest.mark.parametrize("hit", ["all", "some", "none"])
@pytest.mark.parametrize("batches", [1, 2])
@pytest.mark.parametrize("fill", ["full_range", "gaps"])
def test_hash_join_key_only_build_side(hip, oracle, hit, batches, fill):
    """Inner join whose build side is nothing but its unique dense key column (a dimension projected to its key):
    the probe is an existence test against a bitmap and the joined batch the probe batch restricted to the matching
    rows (all of them: shared outright) — same batches as the general route (hash_join.rs:207-292)."""
    rng = np.random.default_rng(len(hit) + batches)
    nb, npr = 50_000, 400_000
    # full_range: the keys fill their range — a probe key inside it has its partner, no bitmap is read; gaps: every other
    # value of the range is missing (the bitmap decides; "all" then means every probe key inside the range, half of them hit)
    bk = (rng.permutation(nb) + 1000).astype(np.int64) if fill == "full_range" else (rng.permutation(nb) * 2 + 1000).astype(np.int64)
    span = nb if fill == "full_range" else 2 * nb
    lo, hi = {"all": (1000, 1000 + span), "some": (0, 2000 + span), "none": (2 * span + 5000, 3 * span + 5000)}[hit]
    lb = pa.RecordBatch.from_arrays([pa.array(bk)], names=["c0"])
    rb = pa.RecordBatch.from_arrays([pa.array(rng.random(npr)), pa.array(rng.integers(lo, hi, npr, dtype=np.int64))], names=["c0", "c1"])
    rbs = [rb] if batches == 1 else [rb.slice(0, 150_000), rb.slice(150_000)]
    cond = JoinCondition([(InputRef(0), InputRef(1))])
    sch = join_schema(lb, rb)
    hip.profile(True)
    got = list(HashJoinExecutor(hip, [lb], rbs, "inner", cond, sch, 1).execute())
    prof = hip.profile_read()
    hip.profile(False)
    assert prof.get("join_semi_mask", (0, 0))[1] == batches, prof
    exp = list(HashJoinExecutor(oracle, [lb], rbs, "inner", cond, sch, 1).execute())
    assert [b.num_rows for b in got] == [b.num_rows for b in exp]
    for g, e in zip(got, exp):
        for c in range(g.num_columns):
            assert g.column(c).equals(e.column(c)), c


@pytest.mark.parametrize("kind", ["i32", "f64", "bool"])
def test_hash_join_key_types(hip, oracle, kind):
    rng = np.random.default_rng(3)
    lb = batch(rng, 500, [(kind, 0.1, 0, 40), ("i64", 0.0)])
    rb = batch(rng, 3000, [(kind, 0.1, 0, 60), ("i64", 0.0)])
    if kind == "f64":  # make float keys collide
        lb = lb.set_column(0, "c0", pa.array(np.floor(lb.column(0).to_numpy(zero_copy_only=False)), mask=np.array(lb.column(0).is_null())))
        rb = rb.set_column(0, "c0", pa.array(np.floor(rb.column(0).to_numpy(zero_copy_only=False)), mask=np.array(rb.column(0).is_null())))
    cond = JoinCondition([(InputRef(0), InputRef(0))])
    sch = join_schema(lb, rb)
    for jt in ("inner", "full"):
        got = list(HashJoinExecutor(hip, [lb], [rb], jt, cond, sch, 2).execute())
        exp = list(HashJoinExecutor(oracle, [lb], [rb], jt, cond, sch, 2).execute())
        assert_same(rows_of(got), rows_of(exp))


def test_hash_join_two_column_keys(hip, oracle):
    rng = np.random.default_rng(4)
    lb = batch(rng, 800, [("i64", 0.05, 0, 20), ("i32", 0.05, 0, 10), ("i64", 0.0)])
    rb = batch(rng, 4000, [("i64", 0.05, 0, 20), ("i32", 0.05, 0, 10), ("f64", 0.0)])
    cond = JoinCondition([(InputRef(0), InputRef(0)), (InputRef(1), InputRef(1))])
    sch = join_schema(lb, rb)
    for jt in ("inner", "left"):
        got = list(HashJoinExecutor(hip, [lb], [rb], jt, cond, sch, 3).execute())
        exp = list(HashJoinExecutor(oracle, [lb], [rb], jt, cond, sch, 3).execute())
        assert_same(rows_of(got), rows_of(exp))


def test_empty_build_side_emits_nothing(hip, oracle):
    rb = batch(np.random.default_rng(1), 10, [("i64", 0.0)])
    sch = pa.schema([pa.field("l.c0", pa.int64()), pa.field("r.c0", pa.int64())])
    cond = JoinCondition([(InputRef(0), InputRef(0))])
    for jt in ("inner", "right", "full"):
        assert list(HashJoinExecutor(hip, [], [rb], jt, cond, sch, 1).execute()) == []
        assert list(HashJoinExecutor(oracle, [], [rb], jt, cond, sch, 1).execute()) == []


# ---------------------------------------------------------------------------- agg --
AGGS = [AggFunc("count", InputRef(1), abi.INT64), AggFunc("sum", InputRef(1), abi.INT64),
        AggFunc("sum", InputRef(2), abi.FLOAT64), AggFunc("count", InputRef(2), abi.INT64),
        AggFunc("min", InputRef(1), abi.INT64), AggFunc("max", InputRef(2), abi.FLOAT64),
        AggFunc("max", InputRef(1), abi.INT64), AggFunc("min", InputRef(2), abi.FLOAT64)]


@pytest.mark.parametrize("n,groups,nulls", [(1, 1, 0.0), (10, 3, 0.3), (1000, 10, 0.1), (5000, 5000, 0.1),
                                            (100_003, 1000, 0.05), (200_000, 150_000, 0.0)])
def test_hash_agg_single_batch(hip, oracle, n, groups, nulls):
    rng = np.random.default_rng(n + groups)
    b = batch(rng, n, [("i64", nulls, 0, groups), ("i64", nulls, -1000, 1000), ("f64", nulls, 0, 1)])
    got = rows_of(HashAggExecutor(hip, AGGS, [InputRef(0)], [b]).execute())
    exp = rows_of(HashAggExecutor(oracle, AGGS, [InputRef(0)], [b]).execute())
    assert_same(got, exp, float_cols={3, 6, 8})


def test_hash_agg_multi_batch_accumulates(hip, oracle):
    """COUNT accumulates across batches on the HIP path (SQL semantics); the reference assigns
    (count.rs:22).  SUM/MIN/MAX must agree with the oracle either way; COUNT agrees with the
    oracle in its default (accumulating) mode."""
    rng = np.random.default_rng(9)
    bs = [batch(rng, n, [("i64", 0.1, 0, 200), ("i64", 0.2, -5, 5), ("f64", 0.0 if i < 2 else 0.3, 0, 1)])
          for i, n in enumerate([1000, 1, 5000, 0, 3000])]
    got = rows_of(HashAggExecutor(hip, AGGS, [InputRef(0)], bs).execute())
    exp = rows_of(HashAggExecutor(oracle, AGGS, [InputRef(0)], bs).execute())
    assert_same(got, exp, float_cols={3, 6, 8})


def test_hash_agg_count_compat_switch_documents_reference_quirk(oracle, oracle_compat):
    """CPU only in effect: the oracle's compat switch reproduces count.rs:22 (last batch wins)."""
    b1 = pa.RecordBatch.from_arrays([pa.array([1, 1, 2], pa.int64()), pa.array([1, 1, 1], pa.int64())], names=["k", "v"])
    b2 = pa.RecordBatch.from_arrays([pa.array([1], pa.int64()), pa.array([1], pa.int64())], names=["k", "v"])
    aggs = [AggFunc("count", InputRef(1), abi.INT64)]
    assert rows_of(HashAggExecutor(oracle, aggs, [InputRef(0)], [b1, b2]).execute()) == [(1, 3), (2, 1)]
    assert rows_of(HashAggExecutor(oracle_compat, aggs, [InputRef(0)], [b1, b2]).execute()) == [(1, 1), (2, 1)]


@pytest.mark.parametrize("kind", ["i32", "f64", "bool"])
def test_hash_agg_key_types(hip, oracle, kind):
    rng = np.random.default_rng(12)
    b = batch(rng, 20_000, [(kind, 0.1, 0, 30), ("i64", 0.1), ("f64", 0.1, 0, 1)])
    if kind == "f64":
        b = b.set_column(0, "c0", pa.array(np.floor(b.column(0).to_numpy(zero_copy_only=False)), mask=np.array(b.column(0).is_null())))
    got = rows_of(HashAggExecutor(hip, AGGS, [InputRef(0)], [b]).execute())
    exp = rows_of(HashAggExecutor(oracle, AGGS, [InputRef(0)], [b]).execute())
    assert_same(got, exp, float_cols={3, 6, 8})


def test_hash_agg_two_keys_and_exprs(hip, oracle):
    rng = np.random.default_rng(13)
    b = batch(rng, 30_000, [("i64", 0.05, 0, 20), ("i64", 0.1), ("f64", 0.1, 0, 1), ("i32", 0.05, 0, 7)])
    aggs = [AggFunc("sum", InputRef(1) + Constant(1, abi.INT64), abi.INT64),
            AggFunc("sum", TypeCast(InputRef(3), abi.INT64), abi.INT64),
            AggFunc("count", InputRef(2), abi.INT64)]
    got = rows_of(HashAggExecutor(hip, aggs, [InputRef(0), InputRef(3)], [b]).execute())
    exp = rows_of(HashAggExecutor(oracle, aggs, [InputRef(0), InputRef(3)], [b]).execute())
    assert_same(got, exp)


def test_hash_agg_without_input_is_internal_error(hip, oracle):
    from sqlrs_amd import ExecutorError
    for be in (hip, oracle):
        with pytest.raises(ExecutorError) as ei:
            list(HashAggExecutor(be, AGGS[:1], [InputRef(0)], []).execute())
        assert ei.value.status == abi.ERR_INTERNAL


# -------------------------------------------------------------------------- order --
@pytest.mark.parametrize("n", [1, 2, 64, 4097, 50_001])
def test_order(hip, oracle, n):
    rng = np.random.default_rng(n)
    b = batch(rng, n, [("i64", 0.1, -20, 20), ("f64", 0.1, -3, 3), ("i32", 0.0, 0, 5), ("bool", 0.2), ("i64", 0.0)])
    bs = [b.slice(0, n // 2), b.slice(n // 2)] if n > 1 else [b]
    for ob in ([OrderBy(InputRef(0), True)], [OrderBy(InputRef(0), False)], [OrderBy(InputRef(1), False)],
               [OrderBy(InputRef(2), True), OrderBy(InputRef(0), False)],
               [OrderBy(InputRef(3), False), OrderBy(InputRef(2), True), OrderBy(InputRef(1), True)]):
        got = rows_of(OrderExecutor(hip, ob, bs).execute())
        exp = rows_of(OrderExecutor(oracle, ob, bs).execute())
        assert_same(got, exp)


# ----------------------------------------------------- partition route of HashAgg --
@pytest.fixture
def force_partition_route(monkeypatch):
    # the library reads this once per process on the first push; GPU tests run it in a
    # subprocess-free way by using batches above the default threshold instead
    yield


@pytest.mark.parametrize("n,groups,nulls", [(2_200_000, 1000, 0.05), (2_200_000, 300_000, 0.0),
                                            (2_300_000, 2_000_000, 0.02), (4_500_000, 50_000, 0.1)])
def test_hash_agg_partition_route(hip, oracle, n, groups, nulls):
    """Batches >= 2^21 rows take the LDS-partitioned pre-aggregation; results (incl. first-seen
    group order) must equal the oracle's."""
    rng = np.random.default_rng(n + groups)
    b = batch(rng, n, [("i64", nulls, 0, groups), ("i64", nulls, -1000, 1000), ("f64", nulls, 0, 1)])
    aggs = [AggFunc("count", InputRef(2), abi.INT64), AggFunc("sum", InputRef(2), abi.FLOAT64),
            AggFunc("sum", InputRef(1), abi.INT64), AggFunc("min", InputRef(1), abi.INT64),
            AggFunc("max", InputRef(2), abi.FLOAT64)]
    got = table_of(HashAggExecutor(hip, aggs, [InputRef(0)], [b]).execute())
    exp = table_of(HashAggExecutor(oracle, aggs, [InputRef(0)], [b]).execute())
    assert_same_table(got, exp, float_cols={2, 5})


@pytest.mark.parametrize("shape", ["one_batch", "batches", "with_filter", "small_batches_filter_on_an_expression", "device_batch_filter_on_an_expression"])
def test_hash_agg_wide_aggregate_list(hip, oracle, shape):
    """More than two argument columns / more accumulator cells than one partition-route operator carries (TPC-H Q1's
    SUM(a), SUM(b), SUM(c), AVG parts, COUNT(*) ...): the operator runs as several parts with the same GROUP BY and
    disjoint aggregates, whose columns line up by first-seen order — no row of it takes the row route's global
    atomics.  Against the oracle, column for column in the order the aggregates were given."""
    rng = np.random.default_rng(len(shape))
    n, groups = 2_400_000, 40_000
    spec = [("i64", 0.02, 0, groups), ("f64", 0.05, 0, 1), ("i64", 0.0, -1000, 1000), ("f64", 0.0, -5, 5), ("i64", 0.1, 0, 50)]
    aggs = [AggFunc("sum", InputRef(1), abi.FLOAT64), AggFunc("count", InputRef(4), abi.INT64), AggFunc("sum", InputRef(2), abi.INT64),
            AggFunc("min", InputRef(3), abi.FLOAT64), AggFunc("max", InputRef(3), abi.FLOAT64), AggFunc("sum", InputRef(3), abi.FLOAT64),
            AggFunc("count", InputRef(1), abi.INT64), AggFunc("max", InputRef(4), abi.INT64), AggFunc("min", InputRef(2), abi.INT64),
            AggFunc("sum", InputRef(4), abi.INT64)]
    if shape == "batches":
        bs = [batch(rng, 3000, spec), batch(rng, n, spec), batch(rng, 70_000, spec)]
    elif shape == "small_batches_filter_on_an_expression":   # staged once in the operator, filtered once, then the parts
        whole = batch(rng, 200_000, spec)
        bs = [whole.slice(o, 1000) for o in range(0, 200_000, 1000)]
    else:
        bs = [batch(rng, n, spec)]
    pf = (InputRef(3) > Constant(-1.0, abi.FLOAT64)) if shape == "with_filter" else None
    if shape.endswith("filter_on_an_expression"):            # not `column OP constant`: the Filter operator runs, once
        pf = (InputRef(3) + InputRef(1)) > Constant(-1.0, abi.FLOAT64)
    child = [hip.to_device(bs[0])] if shape.startswith("device_batch") else bs
    hip.profile(True)
    ex = HashAggExecutor(hip, aggs, [InputRef(0)], child, child_filter=pf)
    got = table_of(ex.execute())
    prof = hip.profile_read()
    hip.profile(False)
    for c in child:
        if c is not bs[0] and hasattr(c, "release"):
            c.release()
    if os.environ.get("SQLRS_AGG_SPLIT") != "0" and shape == "one_batch":  # (the other shapes stage / filter below the route's size)
        assert prof.get("lds_agg", (0, 0))[1] >= 2 and prof.get("agg_update", (0, 0))[1] == 0, prof
    kept = list(FilterExecutor(oracle, pf, bs).execute()) if pf is not None else bs
    exp = table_of(HashAggExecutor(oracle, aggs, [InputRef(0)], kept).execute())
    assert_same_table(got, exp, float_cols={1, 4, 5, 6})


def test_host_batches_cross_the_pinned_staging_threshold(hip, oracle):
    """150 HOST batches of 1000 rows (the reference's CSV batch shape) into the blocking operators: the staging area starts as
    a vector and moves to pinned memory when 2^16 rows have arrived (host_stage.hpp) — values, NULL bitmaps and row order
    must survive the move; HashAgg, Order and the join's build side against the oracle."""
    rng = np.random.default_rng(150)
    n = 150_000
    b = batch(rng, n, [("i64", 0.03, 0, 5000), ("f64", 0.05, 0, 1), ("i64", 0.0, -100, 100)])
    bs = [b.slice(o, 1000) for o in range(0, n, 1000)]
    aggs = [AggFunc("count", InputRef(1), abi.INT64), AggFunc("sum", InputRef(1), abi.FLOAT64), AggFunc("max", InputRef(2), abi.INT64)]
    assert_same_table(table_of(HashAggExecutor(hip, aggs, [InputRef(0)], bs).execute()),
                      table_of(HashAggExecutor(oracle, aggs, [InputRef(0)], bs).execute()), float_cols={2})
    ob = [OrderBy(InputRef(2), asc=False), OrderBy(InputRef(0), asc=True)]
    assert_same_table(table_of(OrderExecutor(hip, ob, bs).execute()), table_of(OrderExecutor(oracle, ob, bs).execute()))
    probe = batch(rng, 20_000, [("i64", 0.0, 0, 6000), ("f64", 0.0, 0, 1)])
    cond = JoinCondition([(InputRef(0), InputRef(0))])
    sch = join_schema(b, probe)
    assert_same_table(table_of(HashJoinExecutor(hip, bs, [probe], "inner", cond, sch, 3).execute()),
                      table_of(HashJoinExecutor(oracle, bs, [probe], "inner", cond, sch, 3).execute()))


def test_hash_agg_mixed_routes_multi_batch(hip, oracle):
    rng = np.random.default_rng(77)
    spec = [("i64", 0.05, 0, 5000), ("f64", 0.0, 0, 1)]
    bs = [batch(rng, 1000, spec), batch(rng, 2_200_000, spec), batch(rng, 50_000, [("i64", 0.05, 0, 9000), ("f64", 0.2, 0, 1)]),
          batch(rng, 2_100_000, [("i64", 0.0, 4000, 12000), ("f64", 0.1, 0, 1)])]
    aggs = [AggFunc("count", InputRef(1), abi.INT64), AggFunc("sum", InputRef(1), abi.FLOAT64),
            AggFunc("min", InputRef(1), abi.FLOAT64)]
    got = rows_of(HashAggExecutor(hip, aggs, [InputRef(0)], bs).execute())
    exp = rows_of(HashAggExecutor(oracle, aggs, [InputRef(0)], bs).execute())
    assert_same(got, exp, float_cols={2, 3})


# -------------------------------------------------------------- exchange primitive --
@pytest.mark.parametrize("parts", [1, 2, 8])
def test_hash_partition_matches_host_restatement(hip, parts):
    from sqlrs_amd import distributed as D
    rng = np.random.default_rng(parts)
    n = 100_003
    keys = rng.integers(-10**12, 10**12, n, dtype=np.int64)
    vals = rng.random(n)
    mask = rng.random(n) < 0.05
    b = pa.RecordBatch.from_arrays([pa.array(keys, mask=mask), pa.array(vals)], names=["k", "v"])
    out, offs = hip.hash_partition(b, InputRef(0), parts, abi.MEM_DEVICE)
    got = hip.to_host(out).to_arrow(["k", "v"])
    exp_cols, exp_offs = D.partition_numpy([keys, vals], parts, valid=~mask)
    assert offs == exp_offs
    gk = got.column(0).to_numpy(zero_copy_only=False)
    ek = np.where(~mask, keys, 0)[np.argsort(D.partition_of(keys, parts, ~mask), kind="stable")]
    assert (np.nan_to_num(gk.astype(float), nan=0.0) == ek.astype(float)).all()
    assert (got.column(1).to_numpy() == exp_cols[1]).all()


# ---------------------------------------------------------------- DISTINCT aggregates --
@pytest.mark.parametrize("n,groups,nulls", [(10, 3, 0.3), (5000, 40, 0.1), (60_000, 3000, 0.05)])
def test_hash_agg_distinct(hip, oracle, n, groups, nulls):
    """count(distinct x) / sum(distinct x) (count.rs:31-58, sum.rs:99-132), mixed with plain
    aggregates; a NULL counts as one distinct value for COUNT, is skipped by SUM."""
    rng = np.random.default_rng(n)
    b = batch(rng, n, [("i64", nulls, 0, groups), ("i64", nulls, 0, 12), ("f64", nulls, 0, 1)])
    bs = [b.slice(0, n // 2), b.slice(n // 2)]
    aggs = [AggFunc("count", InputRef(1), abi.INT64, distinct=True), AggFunc("sum", InputRef(2), abi.FLOAT64),
            AggFunc("sum", InputRef(1), abi.INT64, distinct=True), AggFunc("count", InputRef(2), abi.INT64)]
    got = rows_of(HashAggExecutor(hip, aggs, [InputRef(0)], bs).execute())
    exp = rows_of(HashAggExecutor(oracle, aggs, [InputRef(0)], bs).execute())
    assert_same(got, exp, float_cols={2})


# ------------------------------------------------------------ fused HashJoin + HashAgg --
def _join_agg_reference(oracle, lbs, rbs, cond, sch, nleft, aggs, gb, as_table=False):
    join = HashJoinExecutor(oracle, lbs, rbs, "inner", cond, sch, nleft)
    return (table_of if as_table else rows_of)(HashAggExecutor(oracle, aggs, gb, join.execute()).execute())


@pytest.mark.parametrize("nb,np_,keyrange,nulls,group_on_left", [
    (1000, 140_000, 1500, 0.0, True), (5000, 200_000, 5000, 0.05, False), (300, 2_200_000, 600, 0.02, True),
    (50_000, 300_000, 80_000, 0.0, True)])
def test_join_agg_fused_route(hip, oracle, nb, np_, keyrange, nulls, group_on_left):
    """Unique build keys, group by the join key, arguments from the probe side: the joined batch is
    never materialised; result must equal HashAgg(HashJoin(..)) on the oracle incl. group order."""
    from sqlrs_amd.executor import HashJoinAggExecutor
    rng = np.random.default_rng(nb + np_)
    lkeys = rng.permutation(keyrange)[:nb].astype(np.int64)
    lmask = np.zeros(nb, bool)
    if nulls:
        lmask[0] = True  # one NULL build key: NULL = NULL matches (hash_utils.rs:91-104)
    lb = pa.RecordBatch.from_arrays([pa.array(lkeys, mask=lmask), pa.array(rng.random(nb))], names=["c0", "c1"])
    rb = batch(rng, np_, [("f64", nulls, 0, 1), ("i64", nulls, 0, keyrange), ("i64", nulls, -100, 100)])
    cond = JoinCondition([(InputRef(0), InputRef(1))])
    sch = join_schema(lb, rb)
    gb = [InputRef(0)] if group_on_left else [InputRef(2 + 1)]
    aggs = [AggFunc("count", InputRef(2), abi.INT64), AggFunc("sum", InputRef(2), abi.FLOAT64),
            AggFunc("max", InputRef(4), abi.INT64), AggFunc("min", InputRef(2), abi.FLOAT64)]
    rbs = [rb.slice(0, np_ // 2), rb.slice(np_ // 2)]
    ex = HashJoinAggExecutor(hip, [lb], rbs, cond, sch, 2, aggs, gb)
    got = rows_of(ex.execute())
    assert ex.fused_batches >= 1  # staged probe batches are processed together
    exp = _join_agg_reference(oracle, [lb], rbs, cond, sch, 2, aggs, gb)
    assert_same(got, exp, float_cols={2, 4})


def test_join_agg_composed_route(hip, oracle):
    """Duplicate build keys / arguments from the build side: the library composes join + agg."""
    from sqlrs_amd.executor import HashJoinAggExecutor
    rng = np.random.default_rng(5)
    lb = batch(rng, 3000, [("i64", 0.02, 0, 500), ("i64", 0.0, 0, 9)])
    rb = batch(rng, 100_000, [("i64", 0.02, 0, 700), ("f64", 0.05, 0, 1)])
    cond = JoinCondition([(InputRef(0), InputRef(0))])
    sch = join_schema(lb, rb)
    aggs = [AggFunc("count", InputRef(3), abi.INT64), AggFunc("sum", InputRef(1), abi.INT64), AggFunc("sum", InputRef(3), abi.FLOAT64)]
    ex = HashJoinAggExecutor(hip, [lb], [rb], cond, sch, 2, aggs, [InputRef(0)])
    got = rows_of(ex.execute())
    assert ex.fused_batches == 0
    exp = _join_agg_reference(oracle, [lb], [rb], cond, sch, 2, aggs, [InputRef(0)])
    assert_same(got, exp, float_cols={3})


@pytest.mark.parametrize("build", ["unique", "one_duplicate", "two_nulls", "many_duplicates"])
def test_join_agg_deferred_hash_table(hip, oracle, build):
    """Sparse 64-bit build keys (no direct-address table): a join owned by HashJoinAgg defers its hash table, the
    fused bucket pass inserts the build keys into its LDS tables and is the one that notices a key twice (or two
    NULL keys) — the attempt is dropped, the table is built after all and the operators are composed; unique
    keys never build it.  Fused-eligible shape in every case: group by the join key, arguments from the probe side."""
    from sqlrs_amd.executor import HashJoinAggExecutor
    rng = np.random.default_rng(len(build))
    nb, np_ = 40_000, 300_000
    A = np.int64(0x9E3779B97F4A7C15 - (1 << 64))
    with np.errstate(over="ignore"):
        lkeys = rng.permutation(nb).astype(np.int64) * A + np.int64(17)
        pkeys = rng.integers(0, int(nb * 1.1), np_, dtype=np.int64) * A + np.int64(17)
    lmask = np.zeros(nb, bool)
    if build == "one_duplicate":
        lkeys[nb - 1] = lkeys[3]
    elif build == "many_duplicates":
        lkeys[nb // 2:] = lkeys[:nb - nb // 2]
    elif build == "two_nulls":
        lmask[[5, 777]] = True
    pmask = rng.random(np_) < (0.01 if build == "two_nulls" else 0.0)
    lb = pa.RecordBatch.from_arrays([pa.array(lkeys, mask=lmask), pa.array(rng.random(nb))], names=["c0", "c1"])
    rb = pa.RecordBatch.from_arrays([pa.array(pkeys, mask=pmask), pa.array(rng.random(np_))], names=["c0", "c1"])
    cond = JoinCondition([(InputRef(0), InputRef(0))])
    sch = join_schema(lb, rb)
    aggs = [AggFunc("count", InputRef(3), abi.INT64), AggFunc("sum", InputRef(3), abi.FLOAT64)]
    ex = HashJoinAggExecutor(hip, [lb], [rb], cond, sch, 2, aggs, [InputRef(0)])
    got = rows_of(ex.execute())
    assert (ex.fused_batches >= 1) == (build == "unique")
    exp = _join_agg_reference(oracle, [lb], [rb], cond, sch, 2, aggs, [InputRef(0)])
    assert_same(got, exp, float_cols={2})


# ------------------------------------------------------------------------- Utf8 paths --
def _strings(rng, n, null_frac=0.1):
    alphabet = ["a", "ab", "abc", "b", "", "zz", "abcdefgh", "abcdefghi", "abcdefgh\x00", "Colorado", "CA", "CO", "été"]
    vals = [alphabet[i] + ("" if rng.random() < 0.7 else str(rng.integers(0, 50))) for i in rng.integers(0, len(alphabet), n)]
    mask = rng.random(n) < null_frac
    return pa.array(vals, type=pa.string(), mask=mask)


@pytest.mark.parametrize("n", [1, 65, 5000])
def test_utf8_comparisons_and_filter(hip, oracle, n):
    rng = np.random.default_rng(n)
    b = pa.RecordBatch.from_arrays([_strings(rng, n), _strings(rng, n), pa.array(rng.integers(0, 9, n))], names=["s", "t", "v"])
    for e in (BinaryOp("!=", InputRef(0), Constant("CO", abi.UTF8)), BinaryOp("=", InputRef(0), Constant("abcdefgh", abi.UTF8)),
              BinaryOp(">", InputRef(0), InputRef(1)), BinaryOp("<=", InputRef(0), InputRef(1)),
              BinaryOp(">=", InputRef(0), Constant("ab", abi.UTF8)), BinaryOp("<", InputRef(1), Constant(None, abi.UTF8))):
        assert eval_column(hip, e, b).column(0).to_pylist() == eval_column(oracle, e, b).column(0).to_pylist()
        got = rows_of(FilterExecutor(hip, e, [b]).execute())
        assert got == rows_of(FilterExecutor(oracle, e, [b]).execute())


@pytest.mark.parametrize("n", [2, 300, 20_000])
def test_order_by_utf8(hip, oracle, n):
    rng = np.random.default_rng(n)
    b = pa.RecordBatch.from_arrays([_strings(rng, n), pa.array(rng.integers(0, 5, n)), pa.array(np.arange(n))], names=["s", "k", "i"])
    for ob in ([OrderBy(InputRef(0), True)], [OrderBy(InputRef(0), False)],
               [OrderBy(InputRef(1), True), OrderBy(InputRef(0), False)], [OrderBy(InputRef(0), True), OrderBy(InputRef(1), False)]):
        assert rows_of(OrderExecutor(hip, ob, [b]).execute()) == rows_of(OrderExecutor(oracle, ob, [b]).execute())


def test_utf8_keys_join_and_agg(hip, oracle):
    rng = np.random.default_rng(8)
    lb = pa.RecordBatch.from_arrays([_strings(rng, 400, 0.05), pa.array(rng.integers(0, 100, 400))], names=["s", "x"])
    rb = pa.RecordBatch.from_arrays([pa.array(rng.random(3000)), _strings(rng, 3000, 0.05)], names=["v", "s"])
    cond = JoinCondition([(InputRef(0), InputRef(1))])
    sch = join_schema(lb, rb)
    for jt in ("inner", "full"):
        got = rows_of(HashJoinExecutor(hip, [lb], [rb], jt, cond, sch, 2).execute())
        assert_same(got, rows_of(HashJoinExecutor(oracle, [lb], [rb], jt, cond, sch, 2).execute()))
    aggs = [AggFunc("count", InputRef(0), abi.INT64), AggFunc("sum", InputRef(0), abi.FLOAT64)]
    got = rows_of(HashAggExecutor(hip, aggs, [InputRef(1)], [rb]).execute())
    assert_same(got, rows_of(HashAggExecutor(oracle, aggs, [InputRef(1)], [rb]).execute()), float_cols={2})


def test_hash_agg_partition_route_with_key_skew(hip, oracle):
    """A heavy-hitter key fills one LDS bucket beyond the chunk limit: the bucket is split over
    several workgroups and the duplicate partial groups are merged (agg_partition.hip)."""
    rng = np.random.default_rng(21)
    n = 2_400_000
    keys = rng.integers(0, 20_000, n, dtype=np.int64)
    keys[rng.random(n) < 0.6] = 7  # 60 % of the rows share one key
    b = pa.RecordBatch.from_arrays([pa.array(keys), pa.array(rng.random(n)), pa.array(rng.integers(-9, 9, n, dtype=np.int64))],
                                   names=["k", "v", "w"])
    aggs = [AggFunc("count", InputRef(1), abi.INT64), AggFunc("sum", InputRef(1), abi.FLOAT64),
            AggFunc("min", InputRef(2), abi.INT64), AggFunc("sum", InputRef(2), abi.INT64)]
    got = rows_of(HashAggExecutor(hip, aggs, [InputRef(0)], [b]).execute())
    exp = rows_of(HashAggExecutor(oracle, aggs, [InputRef(0)], [b]).execute())
    assert_same(got, exp, float_cols={2})


@pytest.mark.gpu
@pytest.mark.hooked_rerun("forced_table_overflow")
@pytest.mark.parametrize("scale", ["0.05"])
def test_hash_agg_partition_route_with_forced_table_overflow(scale):
    """SQLRS_EST_SCALE shrinks the HyperLogLog group estimate, so the per-bucket LDS tables
    overflow and rows take the overflow -> global-table path; group order and values must not
    change (the hook is read once per process, hence the subprocess)."""
    from conftest import hooked_rerun
    hooked_rerun("forced_table_overflow")


# ------------------------------------------- code paths added with the round-1 kernel rework --
@pytest.mark.parametrize("keys_kind", ["negative", "offset_2p40", "with_minus_one", "extremes", "wide_random"])
@pytest.mark.parametrize("aggs_kind", ["count_sum_f64", "sum_i64_count", "min_max_f64", "min_i64", "count", "max_i64_generic3"])
def test_hash_agg_partition_route_packed_and_specialised(hip, oracle, keys_kind, aggs_kind):
    """No NULLs + one value column: (key,row) packing when the key range allows it, and the
    compile-time accumulator signatures of lds_agg_kernel; key sets chosen around the packing
    conditions (negative keys, large offsets, the key whose bit pattern is the LDS EMPTY marker,
    INT64 extremes = range too wide to pack, random 64-bit keys)."""
    n = 2_150_000
    rng = np.random.default_rng(hash((keys_kind, aggs_kind)) % (2**32))
    if keys_kind == "negative":
        keys = rng.integers(-40_000, 40_000, n, dtype=np.int64)
    elif keys_kind == "offset_2p40":
        keys = (1 << 40) + rng.integers(0, 200_000, n, dtype=np.int64) * 3
    elif keys_kind == "with_minus_one":
        keys = rng.integers(-3, 5_000, n, dtype=np.int64)  # -1 == ~0ull, the EMPTY marker
    elif keys_kind == "extremes":
        keys = rng.integers(0, 1000, n, dtype=np.int64)
        keys[rng.random(n) < 0.01] = np.iinfo(np.int64).min
        keys[rng.random(n) < 0.01] = np.iinfo(np.int64).max
    else:
        keys = rng.integers(np.iinfo(np.int64).min, np.iinfo(np.int64).max, 60_000, dtype=np.int64)[rng.integers(0, 60_000, n)]
    vi = rng.integers(-10**6, 10**6, n, dtype=np.int64)
    vf = rng.random(n) * 2 - 1
    if aggs_kind == "count_sum_f64":
        vals, aggs, fl = vf, [AggFunc("count", InputRef(1), abi.INT64), AggFunc("sum", InputRef(1), abi.FLOAT64)], {2}
    elif aggs_kind == "sum_i64_count":
        vals, aggs, fl = vi, [AggFunc("sum", InputRef(1), abi.INT64), AggFunc("count", InputRef(1), abi.INT64)], set()
    elif aggs_kind == "min_max_f64":
        vals, aggs, fl = vf, [AggFunc("min", InputRef(1), abi.FLOAT64), AggFunc("max", InputRef(1), abi.FLOAT64)], set()
    elif aggs_kind == "min_i64":
        vals, aggs, fl = vi, [AggFunc("min", InputRef(1), abi.INT64)], set()
    elif aggs_kind == "count":
        vals, aggs, fl = vf, [AggFunc("count", InputRef(1), abi.INT64)], set()
    else:  # three accumulators: the interpreted (generic) kernel on the packed path
        vals, aggs, fl = vi, [AggFunc("max", InputRef(1), abi.INT64), AggFunc("count", InputRef(1), abi.INT64),
                              AggFunc("sum", InputRef(1), abi.INT64)], set()
    b = pa.RecordBatch.from_arrays([pa.array(keys), pa.array(vals)], names=["k", "v"])
    got = rows_of(HashAggExecutor(hip, aggs, [InputRef(0)], [b]).execute())
    exp = rows_of(HashAggExecutor(oracle, aggs, [InputRef(0)], [b]).execute())
    assert_same(got, exp, float_cols=fl)


@pytest.mark.parametrize("n", [16_383, 16_384, 16_385, 5 * 16_384 + 17, 300_001])
@pytest.mark.parametrize("nulls", [0.0, 0.1])
def test_filter_large_tiles(hip, oracle, n, nulls):
    """filter_cmp_const: 16384-row tiles (8 worker waves + scan wave), ragged ends, validity words."""
    rng = np.random.default_rng(n)
    b = batch(rng, n, [("f64", nulls, 0, 1), ("i64", 0.05, -5, 5), ("i32", 0.0, 0, 100)])
    for e in (InputRef(0) > Constant(0.5, abi.FLOAT64), InputRef(0) <= Constant(0.01, abi.FLOAT64),
              BinaryOp("!=", InputRef(2), Constant(7, abi.INT32))):
        got = rows_of(FilterExecutor(hip, e, [b]).execute())
        exp = rows_of(FilterExecutor(oracle, e, [b]).execute())
        assert_same(got, exp)


@pytest.mark.parametrize("jt", ["inner", "left"])
@pytest.mark.parametrize("np_", [16_383, 16_385, 70_000])
@pytest.mark.parametrize("build", ["dense_unique", "dense_unique_null", "dense_dup", "dense_two_nulls"])
def test_hash_join_direct_address_table(hip, oracle, jt, np_, build):
    """Build keys in a small integer range: direct-address table when unique (incl. one NULL
    build key, NULL = NULL matches), hash table when the verify pass finds a duplicate (a repeated
    key or two NULL keys); probe through the 16384-row tile kernel."""
    rng = np.random.default_rng(np_ + len(build))
    nb = 3000
    keys = rng.permutation(4000)[:nb].astype(np.int64) - 500
    mask = np.zeros(nb, dtype=bool)
    if build == "dense_dup":
        keys[17] = keys[2900]
    if build in ("dense_unique_null", "dense_two_nulls"):
        mask[5] = True
    if build == "dense_two_nulls":
        mask[77] = True
    lb = pa.RecordBatch.from_arrays([pa.array(keys, mask=mask), pa.array(rng.random(nb))], names=["k", "x"])
    pk = rng.integers(-600, 3600, np_, dtype=np.int64)
    pmask = rng.random(np_) < 0.03
    rb = pa.RecordBatch.from_arrays([pa.array(rng.random(np_)), pa.array(pk, mask=pmask)], names=["v", "k"])
    cond = JoinCondition([(InputRef(0), InputRef(1))])
    sch = join_schema(lb, rb)
    got = list(HashJoinExecutor(hip, [lb], [rb], jt, cond, sch, lb.num_columns).execute())
    exp = list(HashJoinExecutor(oracle, [lb], [rb], jt, cond, sch, lb.num_columns).execute())
    assert [b.num_rows for b in got] == [b.num_rows for b in exp]
    assert_same(rows_of(got), rows_of(exp))


@pytest.mark.parametrize("kind", ["i64_31bit", "i64_negative", "f64", "i64_const"])
@pytest.mark.parametrize("asc", [True, False])
def test_order_large_with_ties(hip, oracle, kind, asc):
    """300k rows: radix passes over constant bytes are skipped, the leading key column is rebuilt
    from the sorted keys, ties keep their input order (stable, like the oracle)."""
    n = 300_007
    rng = np.random.default_rng(5)
    if kind == "i64_31bit":
        k = pa.array(rng.integers(0, 1 << 31, n, dtype=np.int64) & ~0xff)  # low byte constant too
    elif kind == "i64_negative":
        k = pa.array(rng.integers(-1000, 1000, n, dtype=np.int64))
    elif kind == "f64":
        v = np.round(rng.random(n) * 200 - 100, 1)
        v[::97] = -0.0
        v[::89] = 0.0
        k = pa.array(v)
    else:
        k = pa.array(np.full(n, 42, dtype=np.int64))
    b = pa.RecordBatch.from_arrays([k, pa.array(np.arange(n)), pa.array(rng.random(n))], names=["k", "i", "v"])
    ob = [OrderBy(InputRef(0), asc)]
    got = rows_of(OrderExecutor(hip, ob, [b.slice(0, 100_000), b.slice(100_000)]).execute())
    exp = rows_of(OrderExecutor(oracle, ob, [b.slice(0, 100_000), b.slice(100_000)]).execute())
    assert [math.copysign(1, r[0]) if isinstance(r[0], float) else 0 for r in got] == \
           [math.copysign(1, r[0]) if isinstance(r[0], float) else 0 for r in exp]  # -0.0 vs 0.0 kept apart
    assert_same(got, exp)


def test_group_order_any_keeps_groups_and_values(hip, oracle):
    """SQLRS_GROUP_ORDER_ANY (partial aggregates for the exchange): same groups, same values, only
    the order is unspecified; DISTINCT aggregates refuse it."""
    import ctypes as C
    rng = np.random.default_rng(31)
    n = 2_200_000
    b = pa.RecordBatch.from_arrays([pa.array(rng.integers(0, 70_000, n, dtype=np.int64)), pa.array(rng.random(n))],
                                   names=["k", "v"])
    aggs = [AggFunc("count", InputRef(1), abi.INT64), AggFunc("sum", InputRef(1), abi.FLOAT64)]
    exp = rows_of(HashAggExecutor(oracle, aggs, [InputRef(0)], [b]).execute())
    keep = []
    arr = (abi.AggFunc * 2)(*[a.abi_struct(keep) for a in aggs])
    gb, _k = abi.pack_exprs([InputRef(0)])
    a = C.c_void_p()
    hip.check(hip.fn("hash_agg_create")(hip.ctx, 1, gb, 2, arr, C.byref(a)))
    hip.check(hip.fn("hash_agg_set_group_order")(a, abi.GROUP_ORDER_ANY))
    hb = abi.as_batch(b)
    hip.check(hip.fn("hash_agg_push")(a, hb.ptr))
    out = C.POINTER(abi.Batch)()
    hip.check(hip.fn("hash_agg_finish")(a, abi.MEM_HOST, C.byref(out)))
    got = rows_of([hip.wrap(out).to_arrow(["k", "c", "s"])])
    hip.fn("hash_agg_destroy")(a)
    assert len(got) == len(exp)
    assert_same(sorted(got), sorted(exp), float_cols={2})
    d = C.c_void_p()
    darr = (abi.AggFunc * 1)(AggFunc("count", InputRef(1), abi.INT64, distinct=True).abi_struct(keep))
    hip.check(hip.fn("hash_agg_create")(hip.ctx, 1, gb, 1, darr, C.byref(d)))
    assert hip.fn("hash_agg_set_group_order")(d, abi.GROUP_ORDER_ANY) != 0
    hip.fn("hash_agg_destroy")(d)


@pytest.mark.parametrize("n,groups,nulls", [(1, 1, 0.0), (50, 4, 0.3), (3000, 40, 0.1), (40_000, 3000, 0.05)])
def test_hash_agg_min_max_utf8(hip, oracle, n, groups, nulls):
    """MIN / MAX over Utf8 (min_max.rs:21-29): byte-wise lexicographic, NULLs skipped, all-NULL
    group -> NULL; several batches, mixed with numeric aggregates."""
    rng = np.random.default_rng(n)
    keys = pa.array(rng.integers(0, groups, n, dtype=np.int64))
    strs = _strings(rng, n, nulls)
    nums = pa.array(rng.integers(-100, 100, n, dtype=np.int64))
    b = pa.RecordBatch.from_arrays([keys, strs, nums], names=["k", "s", "x"])
    bs = [b.slice(0, n // 3), b.slice(n // 3)] if n > 3 else [b]
    aggs = [AggFunc("min", InputRef(1), abi.UTF8), AggFunc("count", InputRef(1), abi.INT64),
            AggFunc("max", InputRef(1), abi.UTF8), AggFunc("sum", InputRef(2), abi.INT64)]
    got = rows_of(HashAggExecutor(hip, aggs, [InputRef(0)], bs).execute())
    exp = rows_of(HashAggExecutor(oracle, aggs, [InputRef(0)], bs).execute())
    assert_same(got, exp)


def test_hash_agg_keeps_nothing_borrowed(hip, oracle):
    """Device-resident input batches are released (and their memory reused) right after each push:
    the staged copies must be private (NULLs in keys and values, Utf8 MIN included)."""
    import ctypes as C
    rng = np.random.default_rng(3)
    n = 40_000
    b = pa.RecordBatch.from_arrays([pa.array(rng.integers(0, 300, n, dtype=np.int64), mask=rng.random(n) < 0.03),
                                    pa.array(rng.random(n), mask=rng.random(n) < 0.05), _strings(rng, n, 0.1)],
                                   names=["k", "v", "s"])
    aggs = [AggFunc("count", InputRef(1), abi.INT64), AggFunc("sum", InputRef(1), abi.FLOAT64),
            AggFunc("min", InputRef(2), abi.UTF8)]
    chunks = [b.slice(i, 10_000) for i in range(0, n, 10_000)]
    exp = rows_of(HashAggExecutor(oracle, aggs, [InputRef(0)], chunks).execute())
    keep = []
    arr = (abi.AggFunc * 3)(*[a.abi_struct(keep) for a in aggs])
    gb, _k = abi.pack_exprs([InputRef(0)])
    a = C.c_void_p()
    hip.check(hip.fn("hash_agg_create")(hip.ctx, 1, gb, 3, arr, C.byref(a)))
    for ch in chunks:
        dev = hip.to_device(ch)
        hip.check(hip.fn("hash_agg_push")(a, dev.ptr))
        dev.release()
        # reuse the freed blocks: an unrelated operator with buffers of the same sizes
        junk = pa.RecordBatch.from_arrays([pa.array(np.full(len(ch), -7, dtype=np.int64)), pa.array(np.full(len(ch), 1e30))], names=["a", "b"])
        list(FilterExecutor(hip, InputRef(0) < Constant(0, abi.INT64), [hip.to_device(junk)], out_mem=abi.MEM_DEVICE).execute())
    out = C.POINTER(abi.Batch)()
    hip.check(hip.fn("hash_agg_finish")(a, abi.MEM_HOST, C.byref(out)))
    got = rows_of([hip.wrap(out).to_arrow(["k", "c", "s", "m"])])
    hip.fn("hash_agg_destroy")(a)
    assert_same(got, exp, float_cols={2})


@pytest.mark.gpu
@pytest.mark.hooked_rerun("without_staging")
def test_agg_paths_without_staging():
    """SQLRS_STAGE_DIRECT_ROWS=0 aggregates every pushed batch on its own (the path a first batch of
    >= 2^26 rows takes): per-batch pre-aggregation, deferred groups and merges through the table
    stay covered (the hook is read once per process, hence the subprocess)."""
    from conftest import hooked_rerun
    hooked_rerun("without_staging")


@pytest.mark.gpu
def test_lookback_ticket_fallback_paths():
    """SQLRS_FORCE_TICKET=1: the single-pass kernels (filter, hash-join probe) use their ticketed
    fallback launch, which otherwise only runs after a look-back timeout (hook read once per process,
    hence the subprocess)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, SQLRS_FORCE_TICKET="1")
    here = os.path.abspath(__file__)
    r = subprocess.run([sys.executable, "-m", "pytest", here, "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "(test_filter_cmp_const or test_filter_large_tiles or test_filter_general_expr or test_hash_join_indices or "
                              "test_hash_join_direct_address_table or test_hash_join_lds_tables_general_keys or (test_hash_join and inner)) "
                              "and not fallback"],
                       env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_f64_keys_compare_by_bit_pattern(hip, oracle):
    """hash_utils.rs:124-131: f64 keys are hashed by their bits, so -0.0 and +0.0 are different keys
    and NaNs with different payloads are different keys — group-by and join."""
    import struct
    nan1 = struct.unpack("<d", struct.pack("<Q", 0x7FF8000000000001))[0]
    nan2 = struct.unpack("<d", struct.pack("<Q", 0x7FF8000000000002))[0]
    vals = [0.0, -0.0, nan1, nan2, 1.5, -0.0, nan1, 0.0, nan2, nan2]
    k = pa.array(np.array(vals, dtype=np.float64))
    b = pa.RecordBatch.from_arrays([k, pa.array(np.arange(len(vals), dtype=np.int64))], names=["k", "v"])
    aggs = [AggFunc("count", InputRef(1), abi.INT64), AggFunc("sum", InputRef(1), abi.INT64)]
    got = list(HashAggExecutor(hip, aggs, [InputRef(0)], [b]).execute())[0]
    exp = list(HashAggExecutor(oracle, aggs, [InputRef(0)], [b]).execute())[0]
    bits = lambda t: [struct.unpack("<Q", struct.pack("<d", x))[0] for x in t.column(0).to_pylist()]
    assert bits(got) == bits(exp) and len(bits(got)) == 5
    assert got.column(1).to_pylist() == exp.column(1).to_pylist() == [2, 2, 2, 3, 1]
    assert got.column(2).to_pylist() == exp.column(2).to_pylist()
    lb = pa.RecordBatch.from_arrays([pa.array(np.array([0.0, nan1, 2.0])), pa.array([10, 20, 30])], names=["k", "x"])
    cond = JoinCondition([(InputRef(0), InputRef(0))])
    sch = join_schema(lb, b)
    gj = rows_of(HashJoinExecutor(hip, [lb], [b], "inner", cond, sch, 2).execute())
    ej = rows_of(HashJoinExecutor(oracle, [lb], [b], "inner", cond, sch, 2).execute())
    assert [(r[1], r[3]) for r in gj] == [(r[1], r[3]) for r in ej] == [(10, 0), (20, 2), (20, 6), (10, 7)]  # +0.0 and nan1 rows only


@pytest.mark.parametrize("parts", [2, 8, 200])
@pytest.mark.parametrize("ncols,kcol", [(1, 0), (2, 0), (3, 1), (3, 2)])
def test_hash_partition_fast_path(hip, parts, ncols, kcol):
    """up to three 8-byte columns without NULLs: the stable LDS-staged multi-split; partitions, their
    sizes and the row order inside each must equal the host restatement (stable argsort)."""
    from sqlrs_amd import distributed as D
    rng = np.random.default_rng(parts * 10 + ncols + kcol)
    n = 300_017
    cols = [rng.integers(-10**12, 10**12, n, dtype=np.int64) if c != 1 else rng.random(n) for c in range(ncols)]
    if kcol == 1:
        cols[1] = rng.integers(0, 10**6, n, dtype=np.int64)
    b = pa.RecordBatch.from_arrays([pa.array(c) for c in cols], names=[f"c{i}" for i in range(ncols)])
    out, offs = hip.hash_partition(b, InputRef(kcol), parts, abi.MEM_DEVICE)
    got = hip.to_host(out).to_arrow([f"c{i}" for i in range(ncols)])
    keys = cols[kcol].astype(np.int64)
    pid = D.partition_of(keys, parts, np.ones(n, dtype=bool))
    order = np.argsort(pid, kind="stable")
    exp_offs = [0] + list(np.cumsum(np.bincount(pid, minlength=parts)))
    assert [int(x) for x in offs] == [int(x) for x in exp_offs]
    for c in range(ncols):
        assert (got.column(c).to_numpy() == cols[c][order]).all()


PART_PREDS = {
    "none": (None, lambda c: np.ones(len(c[0]), dtype=bool)),
    "val_gt": (lambda: InputRef(1) > Constant(0.5, abi.FLOAT64), lambda c: c[1] > 0.5),             # the C5 shape
    "key_le": (lambda: InputRef(0) <= Constant(0, abi.INT64), lambda c: c[0] <= 0),                 # predicate on the key
    "other_ne": (lambda: BinaryOp("!=", InputRef(2), Constant(3, abi.INT64)), lambda c: c[2] != 3),
    "general": (lambda: (InputRef(1) > Constant(0.25, abi.FLOAT64)) & (InputRef(0) > Constant(0, abi.INT64)),
                lambda c: (c[1] > 0.25) & (c[0] > 0)),                                              # not fusable: composed
}


@pytest.mark.parametrize("parts", [1, 2, 8, 200])
@pytest.mark.parametrize("pred", ["none", "val_gt", "key_le", "other_ne", "general"])
@pytest.mark.parametrize("n", [300_017, 70_000, 1000])
def test_hash_partition_filter(hip, parts, pred, n):
    """Filter + partition in one pass (sqlrs_hash_partition_filter): partition p of the output must hold exactly
    the kept rows whose key hashes to p (any order inside a partition), columns aligned row by row; small
    batches, general predicates and nullable columns take the composed path with the same contract."""
    from sqlrs_amd import distributed as D
    if n != 300_017 and (parts == 200 or pred in ("key_le", "other_ne")):
        pytest.skip("small batches (composed path) are crossed with the main predicates only")
    rng = np.random.default_rng(parts * 100 + len(pred) + n)
    cols = [rng.integers(-10**12, 10**12, n, dtype=np.int64), rng.random(n), rng.integers(0, 9, n, dtype=np.int64)]
    ncols = 3 if pred == "other_ne" else 2
    cols = cols[:ncols]
    names = [f"c{i}" for i in range(ncols)]
    b = pa.RecordBatch.from_arrays([pa.array(c) for c in cols], names=names)
    mk, keep_of = PART_PREDS[pred]
    out, starts, rows = hip.hash_partition_filter(b, InputRef(0), mk() if mk else None, parts, abi.MEM_DEVICE)
    got = hip.to_host(out).to_arrow(names)
    g = [got.column(c).to_numpy() for c in range(ncols)]
    keep = keep_of(cols)
    pid = D.partition_of(cols[0], parts)
    assert sum(rows) == int(keep.sum())
    for p in range(parts):
        lo, hi = starts[p], starts[p] + rows[p]
        assert p == 0 or lo >= starts[p - 1] + rows[p - 1]  # regions do not overlap
        sel = keep & (pid == p)
        assert rows[p] == int(sel.sum())
        o1 = np.argsort(g[0][lo:hi], kind="stable")  # keys are distinct with overwhelming probability
        o2 = np.argsort(cols[0][sel], kind="stable")
        for c in range(ncols):
            assert (g[c][lo:hi][o1] == cols[c][sel][o2]).all()


def test_hash_partition_filter_nullable_predicate_column(hip):
    """a NULL in the predicate column is a NULL mask row: dropped (filter.rs:13-25 / arrow filter semantics)"""
    from sqlrs_amd import distributed as D
    rng = np.random.default_rng(5)
    n = 200_000
    k, v = rng.integers(0, 10**9, n, dtype=np.int64), rng.random(n)
    mask = rng.random(n) < 0.1
    b = pa.RecordBatch.from_arrays([pa.array(k), pa.array(v, mask=mask)], names=["k", "v"])
    out, starts, rows = hip.hash_partition_filter(b, InputRef(0), InputRef(1) > Constant(0.5, abi.FLOAT64), 4, abi.MEM_DEVICE)
    got = hip.to_host(out).to_arrow(["k", "v"])
    keep = (~mask) & (v > 0.5)
    pid = D.partition_of(k, 4)
    gk = got.column(0).to_numpy()
    for p in range(4):
        assert sorted(gk[starts[p]:starts[p] + rows[p]].tolist()) == sorted(k[keep & (pid == p)].tolist())


@pytest.mark.parametrize("npb", [150_000, 2_300_000])
def test_join_agg_probe_keys_outside_build_range(hip, oracle, npb):
    """Fused route with (key, row) packing: probe keys below / above every build key are stored as the
    sentinel key; every pass must place such a row by that same key (a mismatch between the first
    level's rank and its store once overwrote neighbouring rows)."""
    from sqlrs_amd.executor import HashJoinAggExecutor
    rng = np.random.default_rng(npb)
    lkeys = (1000 + rng.permutation(28_000)[:3000]).astype(np.int64)
    lb = pa.RecordBatch.from_arrays([pa.array(lkeys), pa.array(rng.integers(0, 9, 3000, dtype=np.int64))], names=["k", "x"])
    rb = pa.RecordBatch.from_arrays([pa.array(rng.integers(-500, 30_500, npb, dtype=np.int64)), pa.array(rng.random(npb))], names=["k", "v"])
    cond = JoinCondition([(InputRef(0), InputRef(0))])
    sch = join_schema(lb, rb)
    aggs = [AggFunc("count", InputRef(3), abi.INT64), AggFunc("sum", InputRef(3), abi.FLOAT64)]
    ex = HashJoinAggExecutor(hip, [lb], [rb], cond, sch, 2, aggs, [InputRef(0)])
    got = rows_of(ex.execute())
    assert ex.fused_batches == 1
    exp = _join_agg_reference(oracle, [lb], [rb], cond, sch, 2, aggs, [InputRef(0)])
    assert_same(got, exp, float_cols={2})


# ----------------------------------------- range partition + direct-addressed bucket tables --
@pytest.mark.parametrize("shape", ["uniform", "sparse_quarter", "clustered", "zipf", "one_hot_bucket", "multi_level"])
@pytest.mark.parametrize("aggs_kind", ["count_sum_f64", "min_max_i64", "count_only", "no_aggregates"])
def test_hash_agg_dense_key_route(hip, oracle, shape, aggs_kind):
    """Integer keys that fill most of their range are partitioned by key range and aggregated in
    direct-addressed LDS tables (agg_partition.hip, lds_agg_dense_kernel): uniform keys, a range
    filled to a quarter, keys clustered in few buckets and heavy hitters (bucket chunks merged
    through the direct-addressed split tables), and a range wide enough for two partition levels."""
    n = 2_200_000
    rng = np.random.default_rng(hash((shape, aggs_kind)) % (2**32))
    if shape == "uniform":
        keys = rng.integers(-70_000, 130_000, n, dtype=np.int64)
    elif shape == "sparse_quarter":
        keys = (1 << 33) + rng.choice(400_000, 110_000, replace=False)[rng.integers(0, 110_000, n)]
    elif shape == "clustered":
        keys = np.where(rng.random(n) < 0.9, rng.integers(5_000, 5_300, n), rng.integers(0, 900_000, n)).astype(np.int64)
    elif shape == "zipf":
        keys = np.minimum(rng.zipf(1.1, n), 600_000).astype(np.int64)
    elif shape == "one_hot_bucket":
        keys = np.where(rng.random(n) < 0.6, 77_777, rng.integers(0, 300_000, n)).astype(np.int64)
    else:
        keys = rng.integers(0, 1_050_000, n, dtype=np.int64) - 7
    if aggs_kind == "count_sum_f64":
        v = pa.array(rng.random(n))
        aggs = [AggFunc("count", InputRef(1), abi.INT64), AggFunc("sum", InputRef(1), abi.FLOAT64)]
        fl = {2}
    elif aggs_kind == "min_max_i64":
        v = pa.array(rng.integers(-10**12, 10**12, n, dtype=np.int64))
        aggs = [AggFunc("min", InputRef(1), abi.INT64), AggFunc("max", InputRef(1), abi.INT64)]
        fl = set()
    elif aggs_kind == "count_only":
        v = pa.array(rng.integers(0, 5, n, dtype=np.int64))
        aggs = [AggFunc("count", InputRef(1), abi.INT64)]
        fl = set()
    else:  # SELECT DISTINCT k (planner/select.rs:29-32): group-by without aggregate functions
        v = pa.array(rng.integers(0, 5, n, dtype=np.int64))
        aggs = []
        fl = set()
    b = pa.RecordBatch.from_arrays([pa.array(keys), v], names=["k", "v"])
    got = table_of(HashAggExecutor(hip, aggs, [InputRef(0)], [b]).execute())
    exp = table_of(HashAggExecutor(oracle, aggs, [InputRef(0)], [b]).execute())
    assert_same_table(got, exp, float_cols=fl)


@pytest.mark.parametrize("nb,base", [(40_000, 0), (300_000, -1234), (1_100_000, 1 << 35)])
@pytest.mark.parametrize("hot", [False, True])
def test_join_agg_dense_build_keys(hip, oracle, nb, base, hot):
    """Fused join + group-by whose build keys are a permutation of a full integer range (a dimension
    table's primary key): direct-addressed bucket tables, no build-side partition.  Probe keys reach
    below and above the range (no partner), optionally with heavy hitters."""
    from sqlrs_amd.executor import HashJoinAggExecutor
    npb = 2_300_000
    rng = np.random.default_rng(nb + hot)
    lkeys = (base + rng.permutation(nb)).astype(np.int64)
    lb = pa.RecordBatch.from_arrays([pa.array(lkeys), pa.array(rng.integers(0, 9, nb, dtype=np.int64))], names=["k", "x"])
    pk = rng.integers(base - nb // 10, base + nb + nb // 10, npb, dtype=np.int64)
    if hot:
        pk[rng.random(npb) < 0.5] = base + nb // 3
        pk[rng.random(npb) < 0.1] = base - 1      # heavy hitter without partner
    rb = pa.RecordBatch.from_arrays([pa.array(pk), pa.array(rng.random(npb))], names=["k", "v"])
    cond = JoinCondition([(InputRef(0), InputRef(0))])
    sch = join_schema(lb, rb)
    aggs = [AggFunc("count", InputRef(3), abi.INT64), AggFunc("sum", InputRef(3), abi.FLOAT64)]
    ex = HashJoinAggExecutor(hip, [lb], [rb], cond, sch, 2, aggs, [InputRef(0)])
    got = table_of(ex.execute())
    assert ex.fused_batches == 1
    exp = _join_agg_reference(oracle, [lb], [rb], cond, sch, 2, aggs, [InputRef(0)], as_table=True)
    assert_same_table(got, exp, float_cols={2})


@pytest.mark.parametrize("shape", ["dense_dim_region", "sparse_dim_two_columns", "nullable", "key_and_attribute", "probe_key_and_attribute",
                                   "small_batches", "duplicate_build_keys", "no_match"])
def test_join_agg_group_by_build_columns(hip, oracle, shape):
    """`... FROM fact JOIN dim ON fact.k = dim.k [WHERE ...] GROUP BY dim.region`: eager aggregation — the probe rows are
    grouped by JOIN KEY first (non-materialising route), every distinct key finds its build row once, the partial rows are
    re-aggregated by the build-side columns.  Same groups in the same first-seen order and the same aggregates as
    HashAgg(HashJoin(..)) on the oracle; duplicate build keys (a key no longer determines its build row) and batches
    below the route's size keep the composed route."""
    from sqlrs_amd.executor import HashJoinAggExecutor
    rng = np.random.default_rng(len(shape))
    nb, npb = 300_000, 900_000
    if shape == "sparse_dim_two_columns":
        lkeys = (rng.permutation(nb).astype(np.int64) * 1_000_003 + 17)
    else:
        lkeys = rng.permutation(nb).astype(np.int64) - 5000
    if shape == "duplicate_build_keys":
        lkeys[:1000] = lkeys[1000:2000]
    region = rng.integers(0, 37, nb, dtype=np.int64)
    tier = rng.integers(0, 3, nb).astype(np.int32)
    rmask = (rng.random(nb) < 0.03) if shape == "nullable" else None
    lb = pa.RecordBatch.from_arrays([pa.array(lkeys), pa.array(region, mask=rmask), pa.array(tier)], names=["k", "region", "tier"])
    pk = lkeys[rng.integers(0, nb, npb)]
    miss = rng.random(npb) < (1.0 if shape == "no_match" else 0.1)
    pk = np.where(miss, np.int64(-7_000_000) - rng.integers(0, 1000, npb), pk)  # rows without partner
    vmask = (rng.random(npb) < 0.05) if shape == "nullable" else None
    rb = pa.RecordBatch.from_arrays([pa.array(rng.random(npb), mask=vmask), pa.array(pk), pa.array(rng.integers(-50, 50, npb, dtype=np.int64))],
                                    names=["v", "k", "w"])
    cond = JoinCondition([(InputRef(0), InputRef(1))])
    sch = join_schema(lb, rb)
    gb = {"sparse_dim_two_columns": [InputRef(1), InputRef(2)], "key_and_attribute": [InputRef(0), InputRef(1)],
          "probe_key_and_attribute": [InputRef(2), InputRef(3 + 1)]}.get(shape, [InputRef(1)])  # (InputRef(4) = f.k, the probe-side join key)
    aggs = [AggFunc("count", InputRef(3), abi.INT64), AggFunc("sum", InputRef(3), abi.FLOAT64),
            AggFunc("sum", InputRef(5), abi.INT64), AggFunc("min", InputRef(3), abi.FLOAT64), AggFunc("max", InputRef(5), abi.INT64)]
    if shape == "small_batches":
        rbs = [rb.slice(o, 30_000) for o in range(0, 240_000, 30_000)]  # staged, flushed as one batch below the route's size
    else:
        rbs = [rb.slice(0, npb // 3), rb.slice(npb // 3)]
    pf = (InputRef(0) > Constant(0.25, abi.FLOAT64)) if shape in ("dense_dim_region", "key_and_attribute") else None
    ex = HashJoinAggExecutor(hip, [lb], rbs, cond, sch, 3, aggs, gb, probe_filter=pf)
    got = table_of(ex.execute())
    if shape == "duplicate_build_keys":
        assert ex.eager_groups == 0
    elif shape not in ("small_batches", "no_match") and os.environ.get("SQLRS_EAGER_AGG") != "0":
        assert ex.eager_groups > 0 and ex.fused_batches >= 1
    kept = list(FilterExecutor(oracle, pf, rbs).execute()) if pf is not None else rbs
    exp = _join_agg_reference(oracle, [lb], kept, cond, sch, 3, aggs, gb, as_table=True)
    assert_same_table(got, exp, float_cols={len(gb) + 1, len(gb) + 3})


@pytest.mark.parametrize("shape", ["pairs", "mixed_multiplicities_with_gaps", "hot", "with_filter", "sparse_range"])
def test_join_agg_duplicate_build_keys(hip, oracle, shape):
    """Join + group-by on the join key with DUPLICATE build keys over a dense range: every probe row stands for m joined rows
    (m = build rows with its key), so the direct-addressed route aggregates the probe rows as for unique keys and multiplies
    COUNT / SUM cells by m when a slot is emitted (MIN / MAX unaffected, first-seen order = the probe rows').  Must equal
    HashAgg(HashJoin(..)) on the oracle.  `sparse_range` (1 key in 40 of the range): no dense range, composed route."""
    from sqlrs_amd.executor import HashJoinAggExecutor
    rng = np.random.default_rng(len(shape))
    nkeys, npb, base = 150_000, 700_000, 1 << 33
    if shape == "pairs":
        mult = np.full(nkeys, 2)
    else:
        mult = rng.integers(0, 4, nkeys)  # 0 = a gap in the range
        mult[0] = mult[-1] = 1
    stride = 40 if shape == "sparse_range" else 1
    lkeys = rng.permutation(np.repeat(base + np.arange(nkeys, dtype=np.int64) * stride, mult))
    lb = pa.RecordBatch.from_arrays([pa.array(lkeys), pa.array(rng.integers(0, 9, len(lkeys), dtype=np.int64))], names=["k", "x"])
    pk = base + rng.integers(-500, nkeys + 500, npb, dtype=np.int64) * stride
    if shape == "hot":
        pk[rng.random(npb) < 0.4] = lkeys[3]
        pk[rng.random(npb) < 0.1] = base + int(np.nonzero(mult == 0)[0][2])  # heavy hitter inside a gap
    rb = pa.RecordBatch.from_arrays([pa.array(pk), pa.array(rng.random(npb)), pa.array(rng.integers(-9, 9, npb, dtype=np.int64))], names=["k", "v", "w"])
    cond = JoinCondition([(InputRef(0), InputRef(0))])
    sch = join_schema(lb, rb)
    aggs = [AggFunc("count", InputRef(3), abi.INT64), AggFunc("sum", InputRef(3), abi.FLOAT64), AggFunc("min", InputRef(3), abi.FLOAT64)]
    aggs2 = [AggFunc("sum", InputRef(4), abi.INT64), AggFunc("max", InputRef(4), abi.INT64)]
    pf = (InputRef(1) > Constant(0.3, abi.FLOAT64)) if shape == "with_filter" else None
    kept = list(FilterExecutor(oracle, pf, [rb]).execute()) if pf is not None else [rb]
    for ag, fl in ((aggs, {2, 3}), (aggs2, set())):
        ex = HashJoinAggExecutor(hip, [lb], [rb], cond, sch, 2, ag, [InputRef(0)], probe_filter=pf)
        got = table_of(ex.execute())
        if os.environ.get("SQLRS_DENSE_AGG") != "0":
            assert ex.fused_batches == (0 if shape == "sparse_range" else 1)
        exp = _join_agg_reference(oracle, [lb], kept, cond, sch, 2, ag, [InputRef(0)], as_table=True)
        assert_same_table(got, exp, float_cols=fl)


@pytest.mark.parametrize("keep_one_in", [2, 8, 24])
@pytest.mark.parametrize("hot", [False, True])
def test_join_agg_build_keys_with_gaps(hip, oracle, keep_one_in, hot):
    """Fused join + group-by whose unique build keys cover only PART of their range — a filtered dimension, or the
    hash-partitioned shard of one that a rank of the multi-GPU plan receives: up to 16 slots per key the join keeps
    its direct-address table and the group-by its direct-addressed bucket tables, and a slot is a group only where
    the existence bitmap of the range has its bit (probe keys in a gap have no partner).  1 key in 24: hashed
    buckets, as before.  `hot`: heavy hitters with and without partner (split buckets store their chunk tables)."""
    from sqlrs_amd.executor import HashJoinAggExecutor
    nrange, npb, base = 1_600_000, 800_000, -777
    rng = np.random.default_rng(keep_one_in + hot)
    present = rng.random(nrange) < 1.0 / keep_one_in
    present[0] = present[-1] = True
    lkeys = rng.permutation(base + np.nonzero(present)[0]).astype(np.int64)
    lb = pa.RecordBatch.from_arrays([pa.array(lkeys), pa.array(rng.integers(0, 9, len(lkeys), dtype=np.int64))], names=["k", "x"])
    pk = rng.integers(base - 1000, base + nrange + 1000, npb, dtype=np.int64)
    if hot:
        pk[rng.random(npb) < 0.4] = lkeys[7]                                   # heavy hitter with a partner
        pk[rng.random(npb) < 0.2] = base + int(np.nonzero(~present)[0][5])     # ... and one inside a gap
    rb = pa.RecordBatch.from_arrays([pa.array(pk), pa.array(rng.random(npb))], names=["k", "v"])
    cond = JoinCondition([(InputRef(0), InputRef(0))])
    sch = join_schema(lb, rb)
    aggs = [AggFunc("count", InputRef(3), abi.INT64), AggFunc("sum", InputRef(3), abi.FLOAT64)]
    hip.profile(True)
    ex = HashJoinAggExecutor(hip, [lb], [rb], cond, sch, 2, aggs, [InputRef(0)])
    got = table_of(ex.execute())
    prof = hip.profile_read()
    hip.profile(False)
    assert ex.fused_batches == 1
    direct = prof.get("join_build_dense", (0, 0))[1] > 0 and prof.get("rp_scatter_build", (0, 0))[1] == 0
    if os.environ.get("SQLRS_DENSE_AGG") != "0":
        assert direct == (keep_one_in <= 16), prof
    exp = _join_agg_reference(oracle, [lb], [rb], cond, sch, 2, aggs, [InputRef(0)], as_table=True)
    assert_same_table(got, exp, float_cols={2})


@pytest.mark.gpu
@pytest.mark.hooked_rerun("without_dense_tables")
def test_agg_paths_without_dense_tables():
    """SQLRS_DENSE_AGG=0: dense integer keys go through the hashed partition and the probing LDS
    tables like any other key set, so that path stays covered by the same inputs (hook read once
    per process, hence the subprocess)."""
    from conftest import hooked_rerun
    hooked_rerun("without_dense_tables")


@pytest.mark.parametrize("shape", ["int64_7_groups", "int64_sparse_40_groups", "utf8_10_groups", "two_columns_12_groups"])
def test_hash_agg_partition_route_few_groups(hip, oracle, shape):
    """Few groups on the partition route: the batch is one or a handful of buckets, every bucket is
    cut into many chunks whose LDS tables merge through the per-bucket global tables (a single
    workgroup used to stream the whole batch)."""
    n = 3_000_000
    rng = np.random.default_rng(len(shape))
    v = pa.array(rng.random(n))
    if shape == "int64_7_groups":
        kc, gb = [pa.array(rng.integers(-3, 4, n, dtype=np.int64))], [InputRef(0)]
    elif shape == "int64_sparse_40_groups":
        kc, gb = [pa.array(rng.integers(0, 40, n, dtype=np.int64) * 1_000_003 - 17)], [InputRef(0)]
    elif shape == "utf8_10_groups":
        st = ["CA", "CO", "NY", "TX", "WA", "Colorado State", "California State", "", "zz", "abcdefghij"]
        kc = [pa.DictionaryArray.from_arrays(pa.array(rng.integers(0, len(st), n, dtype=np.int32)), pa.array(st)).cast(pa.string())]
        gb = [InputRef(0)]
    else:
        kc = [pa.array(rng.integers(0, 3, n, dtype=np.int64)), pa.array(rng.integers(10, 14, n, dtype=np.int64))]
        gb = [InputRef(0), InputRef(1)]
    b = pa.RecordBatch.from_arrays(kc + [v], names=[f"k{i}" for i in range(len(kc))] + ["v"])
    vi = len(kc)
    aggs = [AggFunc("count", InputRef(vi), abi.INT64), AggFunc("sum", InputRef(vi), abi.FLOAT64),
            AggFunc("min", InputRef(vi), abi.FLOAT64)]
    got = rows_of(HashAggExecutor(hip, aggs, gb, [b]).execute())
    exp = rows_of(HashAggExecutor(oracle, aggs, gb, [b]).execute())
    assert_same(got, exp, float_cols={len(kc) + 1})


@pytest.mark.parametrize("dtype", ["i32", "i64", "f64"])
def test_filter_fast_path_large_batch_stays_on_its_first_launch(hip, oracle, dtype):
    """The persistent filter kernel needs all its workgroups resident.  The int32 variant (96 registers)
    was launched with two 9-wave workgroups per CU on the occupancy API's word, half of them were not
    resident, and every call sat out the look-back timeout (1.4 s) before the ticketed rerun: results
    were right, 15 000x late.  Checked by result and by a (very generous) time bound."""
    import time
    n = 6_000_000
    rng = np.random.default_rng(11)
    if dtype == "i32":
        col, k = pa.array(rng.integers(0, 100, n).astype(np.int32)), Constant(7, abi.INT32)
    elif dtype == "i64":
        col, k = pa.array(rng.integers(0, 100, n, dtype=np.int64)), Constant(7, abi.INT64)
    else:
        col, k = pa.array(rng.random(n)), Constant(0.25, abi.FLOAT64)
    b = pa.RecordBatch.from_arrays([col, pa.array(np.arange(n, dtype=np.int64))], names=["v", "i"])
    dev = hip.to_device(b)
    e = BinaryOp(">", InputRef(0), k)
    for _ in range(2):  # warm-up: code objects, pool
        for o in FilterExecutor(hip, e, [dev], out_mem=abi.MEM_DEVICE).execute():
            o.release()
    hip.synchronize()
    t = time.perf_counter()
    for o in FilterExecutor(hip, e, [dev], out_mem=abi.MEM_DEVICE).execute():
        o.release()
    hip.synchronize()
    dt = time.perf_counter() - t
    got = rows_of(FilterExecutor(hip, e, [dev]).execute())
    assert got == rows_of(FilterExecutor(oracle, e, [b]).execute())
    assert dt < 0.25, f"filter over {n} rows took {dt * 1e3:.1f} ms: look-back timeout + ticketed rerun?"


@pytest.mark.parametrize("shape", ["col_cmp_const", "general_expr", "with_nulls", "utf8_falls_back"])
def test_filter_push_many_yields_the_batches_of_push(hip, oracle, shape):
    """sqlrs_filter_push_many: a group of small HOST batches (the reference's 1024-row CSV batches, csv.rs:105) uploaded and
    filtered together must come back as exactly the batches sqlrs_filter_push yields one by one — one output batch per
    input batch, in order, empty ones included (filter.rs:15-24) — across every batch boundary; against the oracle's
    per-batch Filter."""
    rng = np.random.default_rng(len(shape))
    sizes = [0, 1, 63, 64, 65, 1024, 1024, 1, 0, 4095, 4096, 4097, 1024, 7, 20_000, 1024, 0, 3]
    batches = []
    for n in sizes:
        v = rng.integers(-50, 50, n, dtype=np.int64)
        w = rng.random(n)
        if shape == "with_nulls":
            cols = [pa.array(v, mask=rng.random(n) < 0.2), pa.array(w, mask=rng.random(n) < 0.1)]
        elif shape == "utf8_falls_back":
            cols = [pa.array(v), pa.array([f"s{int(x * 100)}" for x in w], type=pa.string())]
        else:
            cols = [pa.array(v), pa.array(w)]
        batches.append(pa.RecordBatch.from_arrays(cols, names=["v", "w"]))
    if shape == "general_expr":
        expr = ((InputRef(0) > Constant(3, abi.INT64)) & (InputRef(1) < Constant(0.7, abi.FLOAT64))) | BinaryOp("=", InputRef(0), Constant(-7, abi.INT64))
    else:
        expr = InputRef(0) > Constant(3, abi.INT64)
    exp = list(FilterExecutor(oracle, expr, batches).execute())
    for many in (5, 64):
        got = list(FilterExecutor(hip, expr, batches, many=many).execute())
        assert [b.num_rows for b in got] == [b.num_rows for b in exp]
        for g, e in zip(got, exp):
            assert g.equals(e), (many, g.num_rows)


@pytest.mark.parametrize("shape", ["arith", "with_nulls", "compare_is_boolean", "utf8_falls_back", "constant_column"])
def test_project_push_many_yields_the_batches_of_push(hip, oracle, shape):
    """sqlrs_project_push_many: a group of small HOST batches projected by one launch sequence must come back as exactly the
    batches sqlrs_project_push yields one by one — one output batch per input batch, in order (project.rs:15-27); results
    that are not fixed-width columns (Boolean, Utf8, a constant) and Utf8 inputs fall back to batch by batch; against the
    oracle's per-batch Project."""
    from sqlrs_amd.executor import ProjectExecutor
    rng = np.random.default_rng(len(shape))
    sizes = [0, 1, 63, 64, 65, 1024, 1024, 1, 0, 4095, 4097, 1024, 7, 20_000, 0, 3]
    batches = []
    for n in sizes:
        v = rng.integers(-50, 50, n, dtype=np.int64)
        w = rng.random(n)
        if shape == "with_nulls":
            cols = [pa.array(v, mask=rng.random(n) < 0.2), pa.array(w, mask=rng.random(n) < 0.1)]
        elif shape == "utf8_falls_back":
            cols = [pa.array(v), pa.array([f"s{int(x * 100)}" for x in w], type=pa.string())]
        else:
            cols = [pa.array(v), pa.array(w)]
        batches.append(pa.RecordBatch.from_arrays(cols, names=["v", "w"]))
    if shape == "compare_is_boolean":
        exprs = [InputRef(0) > Constant(3, abi.INT64), InputRef(1)]
    elif shape == "utf8_falls_back":
        exprs = [InputRef(1), InputRef(0) + Constant(1, abi.INT64)]
    elif shape == "constant_column":
        exprs = [Constant(7, abi.INT64), InputRef(0)]
    else:
        exprs = [InputRef(1), InputRef(0) * Constant(3, abi.INT64) + InputRef(0), InputRef(0)]
    exp = list(ProjectExecutor(oracle, exprs, batches).execute())
    for many in (5, 64):
        got = list(ProjectExecutor(hip, exprs, batches, many=many).execute())
        assert [b.num_rows for b in got] == [b.num_rows for b in exp]
        for g, e in zip(got, exp):
            assert g.equals(e), (many, g.num_rows)


@pytest.mark.parametrize("jt", ["inner", "left", "right"])
@pytest.mark.parametrize("build", ["unique_all_hit", "unique_some_miss", "duplicates", "with_filter"])
def test_hash_join_probe_push_many_yields_the_batches_of_probe_push(hip, oracle, jt, build):
    """sqlrs_hash_join_probe_push_many: small HOST probe batches (csv.rs:105) probed together must come back as exactly
    the joined batches sqlrs_hash_join_probe_push yields one by one (hash_join.rs:207-292), batch boundaries included —
    Inner / Left take the grouped route, Right joins and join filters fall back to batch by batch; against the oracle."""
    rng = np.random.default_rng(len(jt) * 7 + len(build))
    nb = 3000
    bk = rng.permutation(nb).astype(np.int64) if build != "duplicates" else rng.integers(0, nb // 3, nb, dtype=np.int64)
    lb = pa.RecordBatch.from_arrays([pa.array(bk), pa.array(bk * 3 + 1)], names=["k", "p"])
    sizes = [0, 1, 63, 64, 65, 1024, 1024, 1, 0, 4095, 4097, 1024, 7, 20_000, 1024, 0, 3]
    hi = nb if build == "unique_all_hit" else int(nb * 1.3)
    rbs = [pa.RecordBatch.from_arrays([pa.array(rng.integers(0, hi, n, dtype=np.int64)), pa.array(rng.random(n))], names=["k", "v"])
           for n in sizes]
    filt = (InputRef(1) > InputRef(2)) if build == "with_filter" else None
    cond = JoinCondition([(InputRef(0), InputRef(0))], filt)
    sch = join_schema(lb, rbs[0])
    exp = list(HashJoinExecutor(oracle, [lb], rbs, jt, cond, sch, 2).execute())
    for many in (4, 64):
        got = list(HashJoinExecutor(hip, [lb], rbs, jt, cond, sch, 2, many=many).execute())
        assert [b.num_rows for b in got] == [b.num_rows for b in exp]
        for g, e in zip(got, exp):
            assert g.equals(e), (many, g.num_rows)
