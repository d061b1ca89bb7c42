"""HashAgg(HashJoin(left, Filter(right))) through sqlrs_join_agg_set_probe_filter, and the chunked
(histogram-free) first partition level that evaluates the filter — against the oracle running the
three operators one after the other (filter.rs:13-25 -> hash_join.rs:146-323 -> hash_agg.rs:32-150).

The chunked level is only taken by large two-level partitions; `test_chunked_first_level_forced`
re-runs the `chunked` cases in a subprocess with SQLRS_RP_CHUNKED=1 / SQLRS_STAGE_DIRECT_ROWS=1 (both
hooks are read once per process), where they also assert that the filter really was fused."""
import os
import subprocess
import sys
import zlib

import numpy as np
import pyarrow as pa
import pytest

from sqlrs_amd import abi
from sqlrs_amd.executor import FilterExecutor, HashAggExecutor, HashJoinAggExecutor, HashJoinExecutor
from sqlrs_amd.expr import AggFunc, BinaryOp, Constant, InputRef, JoinCondition

pytestmark = pytest.mark.gpu
FORCED = os.environ.get("SQLRS_RP_CHUNKED") == "1"


def rows_of(batches):
    """the batches concatenated into one pyarrow Table (column-wise comparison: these tests carry millions of groups,
    row tuples in Python were most of their run time)"""
    batches = list(batches)
    return pa.Table.from_batches(batches) if batches else None


def assert_same(got, exp, float_cols=()):
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
            bad = np.abs(a - b) > 1e-9 * np.maximum(np.abs(b), 1e-300)
            assert not bad.any(), (i, int(bad.argmax()), a[bad.argmax()], b[bad.argmax()])  # SUM(double): 1e-9 relative
        else:
            assert g.equals(e), i


def reference(oracle, lb, rbs, cond, sch, nleft, aggs, gb, pred):
    filt = FilterExecutor(oracle, pred, rbs)
    join = HashJoinExecutor(oracle, [lb], filt.execute(), "inner", cond, sch, nleft)
    return rows_of(HashAggExecutor(oracle, aggs, gb, join.execute()).execute())


def tables(rng, nb, np_, sparse, probe_extra=0.1):
    """build: nb unique keys (dense permutation or sparse 64-bit); probe: [val f64, key, other i64]"""
    bkeys = rng.permutation(nb).astype(np.int64)
    pk = rng.integers(0, int(nb * (1 + probe_extra)), np_, dtype=np.int64)  # some keys without partner
    if sparse:
        A = np.int64(0x9E3779B97F4A7C15 - (1 << 64))
        with np.errstate(over="ignore"):
            bkeys, pk = bkeys * A + np.int64(99), pk * A + np.int64(99)
    lb = pa.RecordBatch.from_arrays([pa.array(bkeys), pa.array(rng.random(nb))], names=["c0", "c1"])
    rb = pa.RecordBatch.from_arrays([pa.array(rng.random(np_)), pa.array(pk), pa.array(rng.integers(-100, 100, np_, dtype=np.int64))],
                                    names=["c0", "c1", "c2"])
    sch = pa.schema([("l.c0", pa.int64()), ("l.c1", pa.float64()), ("r.c0", pa.float64()), ("r.c1", pa.int64()), ("r.c2", pa.int64())])
    return lb, rb, sch


PREDS = {
    "val_gt_half": InputRef(0) > Constant(0.5, abi.FLOAT64),          # the C5 shape: predicate on the aggregated column
    "val_le_all": InputRef(0) <= Constant(2.0, abi.FLOAT64),           # keeps everything
    "val_lt_none": InputRef(0) < Constant(-1.0, abi.FLOAT64),          # keeps nothing
    "other_ne": BinaryOp("!=", InputRef(2), Constant(7, abi.INT64)),   # predicate on a column the aggregates do not read
    "other_eq": BinaryOp("=", InputRef(2), Constant(-3, abi.INT64)),   # ~0.5 % selectivity
    "key_ge": InputRef(1) >= Constant(1000, abi.INT64),                # predicate on the join key itself
    "general": (InputRef(0) > Constant(0.25, abi.FLOAT64)) & (InputRef(2) < Constant(50, abi.INT64)),  # not fusable
}
AGGS = {
    "count_sum": [AggFunc("count", InputRef(2), abi.INT64), AggFunc("sum", InputRef(2), abi.FLOAT64)],
    "none": [],
    "two_columns": [AggFunc("sum", InputRef(2), abi.FLOAT64), AggFunc("max", InputRef(4), abi.INT64), AggFunc("count", InputRef(4), abi.INT64)],
}


# The oracle is a row-at-a-time restatement of the reference: 4-6 s for a 2.5e6-row Filter -> HashJoin -> HashAgg.  The
# two-level cases below therefore share ONE pair of tables per key shape and ONE oracle result per (key shape,
# predicate, aggregate list) — every variant of the HIP side (hooks, record forms, levels) is checked against the same
# expectation instead of paying for its own.
_TABLES, _REF = {}, {}


def shared_tables(sparse):
    if sparse not in _TABLES:
        _TABLES[sparse] = tables(np.random.default_rng(1234 + int(sparse)), 2_400_000 if not sparse else 400_000, 2_500_000, sparse)
    return _TABLES[sparse]


def shared_reference(oracle, sparse, pred, aggs):
    key = (sparse, pred, aggs)
    if key not in _REF:
        lb, rb, sch = shared_tables(sparse)
        _REF[key] = reference(oracle, lb, [rb], JoinCondition([(InputRef(0), InputRef(1))]), sch, 2, AGGS[aggs], [InputRef(0)], PREDS[pred])
    return _REF[key]


@pytest.mark.parametrize("pred", ["val_gt_half", "other_ne", "general"])
@pytest.mark.parametrize("batches", [1, 3])
def test_probe_filter_small_batches(hip, oracle, pred, batches):
    """small probe batches: filtered on arrival, staged, processed together"""
    rng = np.random.default_rng(len(pred) + batches)
    lb, rb, sch = tables(rng, 4000, 150_000, sparse=False)
    rbs = [rb.slice(i * (rb.num_rows // batches), rb.num_rows // batches if i + 1 < batches else rb.num_rows) for i in range(batches)]
    cond = JoinCondition([(InputRef(0), InputRef(1))])
    ex = HashJoinAggExecutor(hip, [lb], rbs, cond, sch, 2, AGGS["count_sum"], [InputRef(0)], probe_filter=PREDS[pred])
    got = rows_of(ex.execute())
    exp = reference(oracle, lb, rbs, cond, sch, 2, AGGS["count_sum"], [InputRef(0)], PREDS[pred])
    assert_same(got, exp, float_cols={2})


@pytest.mark.parametrize("sparse", [False, True])
@pytest.mark.parametrize("pred", ["val_gt_half", "val_le_all", "val_lt_none", "other_ne", "other_eq", "key_ge", "general"])
@pytest.mark.parametrize("aggs", ["count_sum", "none", "two_columns"])
def test_probe_filter_chunked(hip, oracle, sparse, pred, aggs):
    """two-level partitions (dense: 2.4e6 build keys -> 586 range buckets of 4096 slots; sparse: hashed buckets)"""
    if pred == "key_ge" and sparse:
        pytest.skip("the threshold is meant for the dense key range")
    if aggs != "count_sum" and pred not in ("val_gt_half", "other_ne"):
        pytest.skip("aggregate lists are crossed with two predicates only")
    # (every case costs an oracle run of 4-9 s: the hashed-bucket shape keeps the predicates that differ in HOW the filter
    #  reaches the partition pass — on the aggregated column, on another column, not fusable — and the COUNT + SUM list)
    if sparse and (pred not in ("val_gt_half", "other_ne", "general") or (aggs != "count_sum" and pred != "val_gt_half")):
        pytest.skip("the sparse key shape is crossed with three predicates / one predicate per other aggregate list")
    lb, rb, sch = shared_tables(sparse)
    cond = JoinCondition([(InputRef(0), InputRef(1))])
    ex = HashJoinAggExecutor(hip, [lb], [rb], cond, sch, 2, AGGS[aggs], [InputRef(0)], probe_filter=PREDS[pred])
    got = rows_of(ex.execute())
    if FORCED and aggs != "none":  # (without accumulators a bucket table holds 8x the keys: one level, nothing chunked)
        assert ex.fused_batches == 1
        assert ex.filter_fused_batches == (0 if pred == "general" else 1)
    exp = shared_reference(oracle, sparse, pred, aggs)
    assert_same(got, exp, float_cols={2} if aggs == "count_sum" else ({1} if aggs == "two_columns" else ()))


@pytest.mark.parametrize("threshold", [0.3, -1.0])
def test_probe_filter_chunked_hot_digit(hip, oracle, threshold):
    """chunk close / spill logic at test size: every probe key falls into ONE level-1 digit, so each workgroup's chunk
    of that digit overflows after one or two tiles (a tile's run is split between the closing chunk and the next
    one; with `val > -1` every tile fills a chunk exactly and the next tile closes it with no room left) — the
    per-chunk counts of the level-2 digits that level 1 hands to level 2 must follow every one of those moves"""
    rng = np.random.default_rng(int(threshold * 10) + 50)
    nb, np_ = 2_400_000, 6_000_000
    bkeys = rng.permutation(nb).astype(np.int64)
    pk = rng.integers(1000, 1000 + nb // 80, np_, dtype=np.int64)  # ~30 K distinct keys: inside the first level-1 digit
    lb = pa.RecordBatch.from_arrays([pa.array(bkeys), pa.array(rng.random(nb))], names=["c0", "c1"])
    rb = pa.RecordBatch.from_arrays([pa.array(rng.random(np_)), pa.array(pk), pa.array(rng.integers(-100, 100, np_, dtype=np.int64))],
                                    names=["c0", "c1", "c2"])
    sch = pa.schema([("l.c0", pa.int64()), ("l.c1", pa.float64()), ("r.c0", pa.float64()), ("r.c1", pa.int64()), ("r.c2", pa.int64())])
    cond = JoinCondition([(InputRef(0), InputRef(1))])
    pred = InputRef(0) > Constant(threshold, abi.FLOAT64)
    ex = HashJoinAggExecutor(hip, [lb], [rb], cond, sch, 2, AGGS["count_sum"], [InputRef(0)], probe_filter=pred)
    got = rows_of(ex.execute())
    if FORCED:
        assert ex.fused_batches == 1 and ex.filter_fused_batches == 1
    exp = reference(oracle, lb, [rb], cond, sch, 2, AGGS["count_sum"], [InputRef(0)], pred)
    assert_same(got, exp, float_cols={2})


@pytest.mark.parametrize("np_", [150_000, 2_500_000])
def test_probe_filter_guards_a_raising_argument(hip, oracle, np_):
    """SUM(1000 / r.c2) ... WHERE r.c2 <> 0 (int64): the reference filters first (filter.rs:13-25), so the division
    never sees the zero rows.  The fused filter drops rows inside the first partition pass, AFTER the aggregate
    arguments were evaluated over the whole batch — such plans must take the Filter operator first instead of
    failing with "Divide by zero error" (small batch: filtered on arrival; large batch: the in-place route)."""
    rng = np.random.default_rng(77)
    lb, rb, sch = tables(rng, 2_400_000 if np_ > 1_000_000 else 4000, np_, sparse=False)
    assert 0 in rb.column(2).to_pylist()[:5000]
    cond = JoinCondition([(InputRef(0), InputRef(1))])
    aggs = [AggFunc("sum", Constant(1000, abi.INT64) / InputRef(4), abi.INT64), AggFunc("count", InputRef(4), abi.INT64)]
    pred = BinaryOp("!=", InputRef(2), Constant(0, abi.INT64))
    ex = HashJoinAggExecutor(hip, [lb], [rb], cond, sch, 2, aggs, [InputRef(0)], probe_filter=pred)
    got = rows_of(ex.execute())
    assert ex.filter_fused_batches == 0
    exp = reference(oracle, lb, [rb], cond, sch, 2, aggs, [InputRef(0)], pred)
    assert_same(got, exp)
    # without the guard the same plan is an Arrow error on both backends (array_compute.rs:70-90 -> divide)
    with pytest.raises(Exception, match="Divide by zero"):
        rows_of(HashJoinAggExecutor(hip, [lb], [rb], cond, sch, 2, aggs, [InputRef(0)]).execute())


@pytest.mark.parametrize("keys", ["dense", "sparse"])
def test_hash_agg_chunked(hip, oracle, keys):
    """plain HashAgg whose partition needs two levels (2.6e6 groups, > 512 tables of 4096 slots): chunked first level, no filter"""
    rng = np.random.default_rng(11)
    n, G = 6_000_000, 2_600_000
    k = rng.integers(0, G, n, dtype=np.int64)
    if keys == "sparse":
        with np.errstate(over="ignore"):
            k = k * np.int64(0x9E3779B97F4A7C15 - (1 << 64)) + np.int64(5)
    b = pa.RecordBatch.from_arrays([pa.array(k), pa.array(rng.random(n))], names=["k", "v"])
    aggs = [AggFunc("count", InputRef(1), abi.INT64), AggFunc("sum", InputRef(1), abi.FLOAT64), AggFunc("min", InputRef(1), abi.FLOAT64)]
    got = rows_of(HashAggExecutor(hip, aggs, [InputRef(0)], [b]).execute())
    exp = rows_of(HashAggExecutor(oracle, aggs, [InputRef(0)], [b]).execute())
    assert_same(got, exp, float_cols={2})


@pytest.mark.parametrize("outliers", ["none", "unsampled_chunk", "just_outside", "last_chunk"])
def test_hash_agg_optimistic_key_range(hip, outliers):
    """Large batches take their key statistics from a block SAMPLE (every eighth 6144-row chunk + the first and the
    last): the rows are packed with the widened sampled range, and a key outside it — here 1e12 among keys below 5e4,
    hidden in a chunk the sample skips — must be noticed by the first partition pass (KeyPack::oob) and the batch
    re-run with exact statistics.  Checked against numpy (counts bit-exact, sums 1e-9, groups in first-seen order)."""
    rng = np.random.default_rng(31)
    n, G = (1 << 24) + 12345, 50_000
    k = rng.integers(0, G, n, dtype=np.int64)
    if outliers == "unsampled_chunk":
        k[3 * 6144 + 17] = 10**12      # chunk 3: not a multiple of 8
        k[5 * 6144 + 4000] = -(10**12)
    elif outliers == "just_outside":
        # beyond the widened sampled range [-65536, 5e4 + 65536) but below the next power of two (the packed word's
        # sentinel offset 2^18 - 1): representable, yet no bucket of the range partition holds it
        k[3 * 6144 + 17] = 150_000
        k[11 * 6144 + 1] = 150_001
    elif outliers == "last_chunk":
        k[n - 5] = 10**12              # the last chunk is always sampled: the range simply becomes wide (no dense tables)
    v = rng.random(n)
    b = pa.RecordBatch.from_arrays([pa.array(k), pa.array(v)], names=["k", "v"])
    aggs = [AggFunc("count", InputRef(1), abi.INT64), AggFunc("sum", InputRef(1), abi.FLOAT64)]
    got = HashAggExecutor(hip, aggs, [InputRef(0)], [b]).execute()
    (out,) = list(got)
    uk, first, inv = np.unique(k, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")
    exp_k = uk[order]
    exp_c = np.bincount(inv, minlength=len(uk))[order]
    exp_s = np.bincount(inv, weights=v, minlength=len(uk))[order]
    assert out.num_rows == len(uk)
    assert (out.column(0).to_numpy() == exp_k).all()
    assert (out.column(1).to_numpy() == exp_c).all()
    gs = out.column(2).to_numpy()
    assert (np.abs(gs - exp_s) <= 1e-9 * np.abs(exp_s)).all()


@pytest.mark.parametrize("dense", ["1", "0"])
def test_hash_agg_column_form_hook(hip, oracle, dense, monkeypatch):
    """SQLRS_RP_REC=0 (read per call): the final partition level writes key / value columns instead of 16-byte
    records and the bucket pass reads them — the form in-process A/B runs compare against (direct-addressed and
    probing bucket tables)"""
    monkeypatch.setenv("SQLRS_RP_REC", "0")
    monkeypatch.setenv("SQLRS_DENSE_AGG", dense)
    rng = np.random.default_rng(12)
    n, G = 3_000_000, 400_000
    b = pa.RecordBatch.from_arrays([pa.array(rng.integers(0, G, n, dtype=np.int64)), pa.array(rng.random(n))], names=["k", "v"])
    aggs = [AggFunc("count", InputRef(1), abi.INT64), AggFunc("sum", InputRef(1), abi.FLOAT64)]
    got = rows_of(HashAggExecutor(hip, aggs, [InputRef(0)], [b]).execute())
    exp = rows_of(HashAggExecutor(oracle, aggs, [InputRef(0)], [b]).execute())
    assert_same(got, exp, float_cols={2})


@pytest.mark.parametrize("hooks,pred", [("default", "val_gt_half"), ("default", "other_ne"), ("default", "key_ge"),
                                        ("long_tile_ranges_early_closes", "val_gt_half"), ("long_tile_ranges_early_closes", "key_ge"),
                                        ("one_workgroup", "other_ne"), ("off", "val_gt_half"), ("late_starts", "val_gt_half"),
                                        ("off_late_starts", "key_ge")])
def test_chunked_slim_records(hip, oracle, hooks, pred, monkeypatch):
    """Slim records (radix_part.hip): 12-byte rows {value, slot | row-in-tile | tile delta} through both partition levels
    and the bucket pass, row ids rebuilt from the chunk / run tables — the groups' first-seen order (hash_agg.rs:87-99)
    must be the oracle's bit for bit.  Only where the chunked level runs (the forced run).  Hooks (read per call):
    few workgroups = long tile ranges per workgroup, a small largest tile delta = chunks closed early; off = the
    16-byte form on the same input."""
    if not FORCED:
        pytest.skip("needs the chunked first level (SQLRS_RP_CHUNKED=1: the forced run)")
    env = {"default": {}, "long_tile_ranges_early_closes": {"SQLRS_RP_CHUNK_WGS": "3", "SQLRS_RP_SLIM_DELTA": "5"},
           "one_workgroup": {"SQLRS_RP_CHUNK_WGS": "1", "SQLRS_RP_SLIM_DELTA": "127"}, "off": {"SQLRS_RP_SLIM": "0"},
           # (round 6: the bucket starts of level 2 are fetched AHEAD of its scatter by default; 0 = behind the level as before —
           #  slim and 16-byte rows take different call sites of the same helper)
           "late_starts": {"SQLRS_RP_EARLY_STARTS": "0"}, "off_late_starts": {"SQLRS_RP_SLIM": "0", "SQLRS_RP_EARLY_STARTS": "0"}}[hooks]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    lb, rb, sch = shared_tables(False)
    cond = JoinCondition([(InputRef(0), InputRef(1))])
    hip.profile(True)
    ex = HashJoinAggExecutor(hip, [lb], [rb], cond, sch, 2, AGGS["count_sum"], [InputRef(0)], probe_filter=PREDS[pred])
    got = rows_of(ex.execute())
    prof = hip.profile_read()
    hip.profile(False)
    assert ex.fused_batches == 1 and ex.filter_fused_batches == 1
    assert (prof.get("rp_slim_runs", (0, 0))[1] > 0) == (not hooks.startswith("off")), prof
    assert_same(got, shared_reference(oracle, False, pred, "count_sum"), float_cols={2})


@pytest.mark.hooked_rerun("chunked_first_level")
def test_chunked_first_level_forced():
    """the `chunked` cases again in a child pytest with SQLRS_RP_CHUNKED=1 / SQLRS_STAGE_DIRECT_ROWS=1 (read once per process;
    conftest.HOOKED_RERUNS), where they also assert that the filter really was fused.  The same level with SQLRS_RP_H2=0 —
    level 2 running its own histogram pass — is part of that run (test_chunked_level_without_chunk_histograms: read per call)"""
    if FORCED:
        pytest.skip("already inside the forced run")
    from conftest import hooked_rerun
    hooked_rerun("chunked_first_level")


@pytest.mark.parametrize("sparse", [False, True])
def test_chunked_level_without_chunk_histograms(hip, oracle, sparse, monkeypatch):
    """SQLRS_RP_H2=0 (read per call): level 2 runs its own histogram pass instead of taking the chunk histograms the
    first level counted on the way; only meaningful where the chunked level runs (the forced run, or full-size batches)"""
    monkeypatch.setenv("SQLRS_RP_H2", "0")
    test_probe_filter_chunked(hip, oracle, sparse, "val_gt_half", "count_sum")


def _claimed_case(hip, oracle, k, v, expect_claimed, aggs=None):
    b = pa.RecordBatch.from_arrays([pa.array(k), pa.array(v)], names=["k", "v"])
    aggs = aggs or [AggFunc("count", InputRef(1), abi.INT64), AggFunc("sum", InputRef(1), abi.FLOAT64)]
    hip.profile(True)
    got = rows_of(HashAggExecutor(hip, aggs, [InputRef(0)], [b]).execute())
    prof = hip.profile_read()
    hip.profile(False)
    exp = rows_of(HashAggExecutor(oracle, aggs, [InputRef(0)], [b]).execute())
    assert_same(got, exp, float_cols={i + 1 for i, a in enumerate(aggs) if a.return_type == abi.FLOAT64})
    claimed = prof.get("rp_claim_scatter", (0, 0))[1] > 0
    counted = prof.get("rp_scatter", (0, 0))[1] > 0
    assert claimed, prof
    if expect_claimed is not None:
        assert counted == (not expect_claimed), prof  # an overflowed region: the counting level redoes the batch


@pytest.mark.parametrize("shape", ["uniform", "zipf", "few_rows_per_bucket", "one_hot_bucket", "more_than_256_buckets"])
def test_hash_agg_claimed_single_level(hip, oracle, shape, monkeypatch):
    """One-level range partitions without a histogram pass (rp_claim_scatter_kernel): bucket regions sized from a
    sample, blocks claimed with one atomic per (workgroup, digit, block), sentinel rows in what is left of a block.
    Forced at test size (SQLRS_RP_CLAIM=1, read per call); against the oracle, groups in first-seen order."""
    monkeypatch.setenv("SQLRS_RP_CLAIM", "1")
    rng = np.random.default_rng(len(shape))
    n, G = 3_000_000, 400_000
    if shape == "uniform":
        k = rng.integers(0, G, n, dtype=np.int64)
    elif shape == "zipf":
        k = (np.minimum(rng.zipf(1.2, n), G) - 1).astype(np.int64)
        k = (k * 7919 + 13) % G
    elif shape == "more_than_256_buckets":  # 1.5e6 keys = 367 buckets of 4096 slots: digits beyond 255 (768-thread workgroups: no power-of-two masks)
        n, G = 3_000_000, 1_500_000
        k = rng.integers(0, G, n, dtype=np.int64)
    elif shape == "few_rows_per_bucket":
        n, G = 2_200_000, 2_000_000  # 489 buckets, ~17 rows per (workgroup, bucket): 16-row blocks
        k = rng.integers(0, G, n, dtype=np.int64)
    else:  # nine rows in ten fall into the key range of one bucket: its region takes most of the batch
        k = np.where(rng.random(n) < 0.9, rng.integers(200_000, 203_000, n), rng.integers(0, G, n)).astype(np.int64)
    _claimed_case(hip, oracle, k, rng.random(n), expect_claimed=True)


@pytest.mark.parametrize("shape", ["uniform", "one_hot_bucket"])
def test_hash_agg_claimed_level_16_byte_form(hip, oracle, shape, monkeypatch):
    """SQLRS_RP_SLIM=0 (read per call): the claimed level writes 16-byte {key|row, value} records instead of the slim
    value + 32-bit word rows (rp_claim_scatter_kernel / lds_agg_dense_kernel: the route of lists the slim form does not take)"""
    monkeypatch.setenv("SQLRS_RP_SLIM", "0")
    test_hash_agg_claimed_single_level(hip, oracle, shape, monkeypatch)


@pytest.mark.parametrize("delta", [1, 3])
@pytest.mark.parametrize("shape", ["uniform", "rare_buckets"])
def test_hash_agg_claimed_slim_abandoned_blocks(hip, oracle, shape, delta, monkeypatch):
    """Slim rows of the claimed level carry a 7-bit tile delta against the tile that claimed their BLOCK; a digit whose open
    block is older than that abandons it (sentinel rows).  SQLRS_RP_SLIM_DELTA (read per call) makes that happen after 1 / 3
    tiles: first-seen order (row ids rebuilt from block base tile + delta, hash_agg.rs:87-99) against the oracle.
    rare_buckets: most digits see a row only every few tiles."""
    monkeypatch.setenv("SQLRS_RP_CLAIM", "1")
    monkeypatch.setenv("SQLRS_RP_SLIM_DELTA", str(delta))
    monkeypatch.setenv("SQLRS_RP_CHUNK_WGS", "40")  # (read per call: ~11 tiles per workgroup instead of 2)
    rng = np.random.default_rng(11 * delta + len(shape))
    n, G = 2_500_000, 400_000
    if shape == "uniform":
        k = rng.integers(0, G, n, dtype=np.int64)
    else:  # 88 % of the rows on 2 % of the key range, the rest spread thin over all of it (still most keys of the range: dense)
        k = np.where(rng.random(n) < 0.88, rng.integers(100_000, 108_000, n), rng.integers(0, G, n)).astype(np.int64)
    # (rare_buckets: blocks abandoned every few tiles can use up a region's slack — the counting level then redoes the batch;
    #  either way the result is checked)
    _claimed_case(hip, oracle, k, rng.random(n), expect_claimed=True if shape == "uniform" else None)


def test_hash_agg_claimed_level_natural_size(hip, oracle):
    """>= 2^22 rows take the claimed level on their own (no hook); MIN / MAX over an int64 column"""
    rng = np.random.default_rng(77)
    n, G = (1 << 22) + 4321, 600_000
    k = rng.integers(-300_000, G - 300_000, n, dtype=np.int64)
    b = pa.RecordBatch.from_arrays([pa.array(k), pa.array(rng.integers(-10**9, 10**9, n, dtype=np.int64))], names=["k", "v"])
    aggs = [AggFunc("min", InputRef(1), abi.INT64), AggFunc("max", InputRef(1), abi.INT64)]
    hip.profile(True)
    got = rows_of(HashAggExecutor(hip, aggs, [InputRef(0)], [b]).execute())
    prof = hip.profile_read()
    hip.profile(False)
    assert prof.get("rp_claim_scatter", (0, 0))[1] > 0, prof
    assert_same(got, rows_of(HashAggExecutor(oracle, aggs, [InputRef(0)], [b]).execute()))


def test_hash_agg_claimed_level_overflow_falls_back(hip, oracle, monkeypatch):
    """The sample reads the (5 t mod 8)-th eighth of tile t.  Here exactly those rows carry keys of the lower half of
    the range and every other row a key of the upper half: the regions of the upper buckets get the minimum size,
    overflow, and the counting level must redo the batch (same result, both scatter kernels in the profile)."""
    monkeypatch.setenv("SQLRS_RP_CLAIM", "1")
    rng = np.random.default_rng(5)
    n, G, tile = 3_000_000, 400_000, 6144
    r = np.arange(n)
    t, o = r // tile, r % tile
    sampled = (o // (tile // 8)) == ((t * 5) % 8)
    k = np.where(sampled, rng.integers(0, G // 2, n), rng.integers(G // 2, G, n)).astype(np.int64)
    _claimed_case(hip, oracle, k, rng.random(n), expect_claimed=False)


AGG_PREDS = {
    "val_gt_half": InputRef(1) > Constant(0.5, abi.FLOAT64),           # predicate on the aggregated column
    "other_ne": BinaryOp("!=", InputRef(2), Constant(7, abi.INT64)),    # on a column the aggregates do not read
    "key_ge": InputRef(0) >= Constant(1000, abi.INT64),                 # on the group key itself
    "val_lt_none": InputRef(1) < Constant(-1.0, abi.FLOAT64),           # keeps nothing
    "general": (InputRef(1) > Constant(0.25, abi.FLOAT64)) & (InputRef(2) < Constant(50, abi.INT64)),  # not fusable
}


@pytest.mark.parametrize("shape", ["dense_one_level", "dense_two_level", "sparse"])
@pytest.mark.parametrize("pred", ["val_gt_half", "other_ne", "key_ge", "val_lt_none", "general"])
def test_hash_agg_child_filter(hip, oracle, shape, pred, monkeypatch):
    """HashAgg(Filter(child)) with the filter handed to the aggregate (sqlrs_hash_agg_set_filter) against the oracle
    running FilterExecutor -> HashAggExecutor (filter.rs:13-25 feeding hash_agg.rs:44); groups in first-seen order of
    the FILTERED rows.  At test size the library runs its Filter operator first; in the forced run (batches aggregated
    in place, see test_chunked_first_level_forced) `col OP constant` predicates must be evaluated by the first
    partition pass: the claimed single level (range partitions of <= 512 buckets) or the chunked first level."""
    if shape != "dense_one_level" and pred in ("key_ge", "val_lt_none"):
        pytest.skip("crossed with the one-level shape only")
    monkeypatch.setenv("SQLRS_RP_CLAIM", "1")
    rng = np.random.default_rng(zlib.crc32(f"{shape}{pred}".encode()))
    n = 3_000_000
    G = {"dense_one_level": 400_000, "dense_two_level": 2_600_000, "sparse": 300_000}[shape]
    k = rng.integers(0, G, n, dtype=np.int64)
    if shape == "sparse":
        with np.errstate(over="ignore"):
            k = k * np.int64(0x9E3779B97F4A7C15 - (1 << 64)) + np.int64(99)
    b = pa.RecordBatch.from_arrays([pa.array(k), pa.array(rng.random(n)), pa.array(rng.integers(-100, 100, n, dtype=np.int64))],
                                   names=["k", "v", "c"])
    aggs = [AggFunc("count", InputRef(1), abi.INT64), AggFunc("sum", InputRef(1), abi.FLOAT64)]
    ex = HashAggExecutor(hip, aggs, [InputRef(0)], [b], child_filter=AGG_PREDS[pred])
    got = rows_of(ex.execute())
    exp = rows_of(HashAggExecutor(oracle, aggs, [InputRef(0)], FilterExecutor(oracle, AGG_PREDS[pred], [b]).execute()).execute())
    if exp is not None and exp.num_rows == 0:
        exp = None
    if got is not None and got.num_rows == 0:
        got = None
    assert_same(got, exp, float_cols={2})
    if FORCED:
        # (sparse: 3e5 hashed groups need 612 probing tables — two levels, the chunked first level takes the filter)
        assert ex.filter_fused_batches == (1 if pred != "general" else 0)


def test_hash_agg_child_filter_batches_and_raising_argument(hip, oracle):
    """several small batches (filtered on arrival, staged, aggregated together) and an aggregate argument that
    divides by a column which is zero only in rows the filter removes: must not raise (the argument is evaluated
    behind the filter, as in the reference), DISTINCT aggregate included"""
    rng = np.random.default_rng(3)
    n = 200_000
    c = rng.integers(0, 5, n, dtype=np.int64)
    b = pa.RecordBatch.from_arrays([pa.array(rng.integers(0, 1000, n, dtype=np.int64)), pa.array(rng.integers(1, 100, n, dtype=np.int64)), pa.array(c)],
                                   names=["k", "v", "c"])
    bs = [b.slice(i * 50_000, 50_000) for i in range(4)]
    pred = InputRef(2) > Constant(0, abi.INT64)
    aggs = [AggFunc("sum", BinaryOp("/", InputRef(1), InputRef(2)), abi.INT64), AggFunc("count", InputRef(1), abi.INT64, distinct=True)]
    got = rows_of(HashAggExecutor(hip, aggs, [InputRef(0)], bs, child_filter=pred).execute())
    exp = rows_of(HashAggExecutor(oracle, aggs, [InputRef(0)], FilterExecutor(oracle, pred, bs).execute()).execute())
    assert_same(got, exp)
