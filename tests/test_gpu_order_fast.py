"""ORDER BY one key column on the fast route of order_fast.hip (rows travel through <= 2 HBM passes on the
top key bits + an in-LDS finish) against the oracle (order.rs:15-67), including the shapes that must fall
back to the general path.  A row-id column makes tie order (stable, like the general path) visible."""
import os
import numpy as np
import pyarrow as pa
import pytest

from sqlrs_amd import abi
from sqlrs_amd.executor import LimitExecutor, OrderExecutor
from sqlrs_amd.expr import InputRef, OrderBy

pytestmark = pytest.mark.gpu
N = 1_300_000


def keys_of(rng, shape):
    if shape == "i64_31bit":
        return rng.integers(0, 1 << 31, N, dtype=np.int64)
    if shape == "i64_negative_20bit":
        return rng.integers(-(1 << 19), 1 << 19, N, dtype=np.int64)
    if shape == "i64_10bit_many_ties":       # all key bits sorted in HBM (rbits = 0), groups of ~1300 equal keys
        return rng.integers(-500, 500, N, dtype=np.int64)
    if shape == "i64_offset_2p40":
        return (1 << 40) + rng.integers(0, 1 << 24, N, dtype=np.int64)
    if shape == "i64_wide":                  # > 32 varying bits: general path
        return rng.integers(-(1 << 62), 1 << 62, N, dtype=np.int64)
    if shape == "i64_heavy_groups":          # 50 distinct keys spread over 2^26: a group exceeds the LDS finish -> general path
        return rng.choice(rng.integers(0, 1 << 26, 50, dtype=np.int64), N)
    if shape == "i64_constant":
        return np.full(N, 7, dtype=np.int64)
    if shape == "f64_unit":                  # doubles in [0, 1): 52 varying bits -> general path
        return rng.random(N)
    if shape == "f64_few":                   # doubles from a small set over a narrow bit range
        return (rng.integers(0, 4096, N) / 4096.0 + 1.0)
    if shape == "i32":
        return rng.integers(-(1 << 20), 1 << 20, N).astype(np.int32)
    raise ValueError(shape)


@pytest.mark.parametrize("shape", ["i64_31bit", "i64_negative_20bit", "i64_10bit_many_ties", "i64_offset_2p40", "i64_wide",
                                   "i64_heavy_groups", "i64_constant", "f64_unit", "f64_few", "i32"])
@pytest.mark.parametrize("asc", [True, False])
@pytest.mark.parametrize("extra", ["none", "carry", "carry_and_more"])
def test_order_fast_route(hip, oracle, shape, asc, extra):
    if extra != "carry" and shape not in ("i64_31bit", "i64_10bit_many_ties", "i32", "f64_few"):
        pytest.skip("column mixes are crossed with four key shapes only")
    rng = np.random.default_rng(abs(hash_seed(shape, asc, extra)))
    k = keys_of(rng, shape)
    cols, names = [pa.array(k)], ["k"]
    if extra != "none":
        cols.append(pa.array(np.arange(N, dtype=np.int64)))          # row id: carried, shows the tie order
        names.append("row")
    if extra == "carry_and_more":
        cols += [pa.array(rng.random(N), mask=rng.random(N) < 0.1), pa.array(rng.integers(0, 100, N).astype(np.int32)),
                 pa.array([None if i % 11 == 0 else f"s{i % 97}" for i in range(N)])]
        names += ["f", "i", "s"]
    b = pa.RecordBatch.from_arrays(cols, names=names)
    bs = [b.slice(0, N // 3), b.slice(N // 3)]
    (got,) = list(OrderExecutor(hip, [OrderBy(InputRef(0), asc=asc)], bs).execute())
    (exp,) = list(OrderExecutor(oracle, [OrderBy(InputRef(0), asc=asc)], bs).execute())
    for i in range(b.num_columns):
        assert got.column(i).equals(exp.column(i)), names[i]


def hash_seed(*parts):
    import zlib
    return zlib.crc32("|".join(str(p) for p in parts).encode())


@pytest.mark.parametrize("hooks", [{"SQLRS_ORDER_TILED": "0"}, {"SQLRS_ORDER_REC": "0"}, {"SQLRS_ORDER_REC1": "0"}, {"SQLRS_ORDER_WIDE_REC1": "0"},
                                   {"SQLRS_ORDER_LB": "0"}, {"SQLRS_ORDER_LB_TEST_FAIL": "1"}, {"SQLRS_ORDER_LB_TEST_FAIL": "2"}, {"SQLRS_ORDER_SLIM": "0"}, {"SQLRS_ORDER_FINISH_COUNT": "0"},
                                   {"SQLRS_ORDER_LB": "0", "TEST_WIDE_KEYS": "1"}, {"SQLRS_ORDER_LB_TEST_FAIL": "1", "TEST_WIDE_KEYS": "1"}])
def test_order_fast_route_ab_hooks(hip, oracle, hooks, monkeypatch):
    """the A/B hooks of the fast route (read per call) keep the older forms alive: plain 4096-row blocks with a
    boundary scan of the sorted words, key / value columns instead of 16-byte records into the finish, and (SQLRS_ORDER_SLIM=0) the
    16-byte {word, value} records of round 5 instead of the 12-byte {key offset, value} ones the look-back form moves when nobody
    asks for the permutation"""
    for k_, v_ in hooks.items():
        monkeypatch.setenv(k_, v_)
    rng = np.random.default_rng(77)
    wide = "SQLRS_ORDER_WIDE_REC1" in hooks or "TEST_WIDE_KEYS" in hooks   # (TEST_WIDE_KEYS: no hook of the library, only this choice)
    k = rng.integers(0, 1 << 31, N, dtype=np.int64) if not wide else rng.integers(-(1 << 62), 1 << 62, N, dtype=np.int64)
    b = pa.RecordBatch.from_arrays([pa.array(k), pa.array(np.arange(N, dtype=np.int64))], names=["k", "row"])
    (got,) = list(OrderExecutor(hip, [OrderBy(InputRef(0), asc=True)], [b]).execute())
    (exp,) = list(OrderExecutor(oracle, [OrderBy(InputRef(0), asc=True)], [b]).execute())
    assert got.equals(exp)


@pytest.mark.parametrize("lb", ["1", "0"])
@pytest.mark.parametrize("asc", [True, False])
def test_order_split_passes_with_empty_segments(hip, oracle, asc, lb, monkeypatch):
    """26-bit keys whose first-pass digit (bits 16..23) takes 64 of its 256 values and whose second-pass digit two bits: 192
    segments of the tiled pass have no tile, so the look-back form's group table must pair every segment's bounds with those
    of the NEXT SEGMENT THAT HAS ROWS (ow_group_table_lb_kernel); 256 groups of ~5000 rows each reach the in-LDS finish.
    Both forms of the split passes (SQLRS_ORDER_LB, read per call)."""
    monkeypatch.setenv("SQLRS_ORDER_LB", lb)
    rng = np.random.default_rng(260 + asc)
    k = (rng.integers(0, 4, N, dtype=np.int64) << 24) | ((rng.integers(0, 64, N, dtype=np.int64) * 4) << 16) | rng.integers(0, 1 << 16, N, dtype=np.int64)
    b = pa.RecordBatch.from_arrays([pa.array(k), pa.array(np.arange(N, dtype=np.int64))], names=["k", "row"])
    hip.profile(True)
    (got,) = list(OrderExecutor(hip, [OrderBy(InputRef(0), asc=asc)], [b]).execute())
    prof = hip.profile_read()
    hip.profile(False)
    (exp,) = list(OrderExecutor(oracle, [OrderBy(InputRef(0), asc=asc)], [b]).execute())
    assert got.column(0).equals(exp.column(0)) and got.column(1).equals(exp.column(1))
    # (no second attempt: the groups fit the finish; the look-back form is one scope, the counting form two)
    assert prof.get("order_split", (0, 0))[1] == (1 if lb == "1" else 2) and prof.get("order_finish", (0, 0))[1] == 1, prof


@pytest.mark.parametrize("shape", ["sample_holds", "outlier_low", "outlier_high", "sorted"])
@pytest.mark.parametrize("asc", [True, False])
def test_order_fast_optimistic_key_range(hip, oracle, shape, asc, monkeypatch):
    """Columns of >= 2^24 rows take their key range from a SAMPLE (every 16th chunk of 2048 rows + the last rows) and the
    first split pass tests every key against it; forced at test size by SQLRS_ORDER_SAMPLE=1 (read per call).  An outlier
    in a chunk the sample does not read must send the call through the exact pass once (two order_minmax launches)."""
    monkeypatch.setenv("SQLRS_ORDER_SAMPLE", "1")
    rng = np.random.default_rng(len(shape) + asc)
    k = rng.integers(1 << 20, 1 << 30, N, dtype=np.int64)
    if shape == "outlier_low":
        k[2048 * 3 + 7] = -5
    elif shape == "outlier_high":
        k[2048 * 21 + 100] = (1 << 31) + 12345
    elif shape == "sorted":
        k = np.sort(k)
    b = pa.RecordBatch.from_arrays([pa.array(k), pa.array(np.arange(N, dtype=np.int64))], names=["k", "row"])
    hip.profile(True)
    (got,) = list(OrderExecutor(hip, [OrderBy(InputRef(0), asc=asc)], [b]).execute())
    prof = hip.profile_read()
    hip.profile(False)
    (exp,) = list(OrderExecutor(oracle, [OrderBy(InputRef(0), asc=asc)], [b]).execute())
    assert got.column(0).equals(exp.column(0)) and got.column(1).equals(exp.column(1))
    # (ascending keys ordered ascending: the sampled neighbour pairs show no inversion, the full test — a second scope of
    #  the class — finds the rows in order and nothing is sorted: test_order_rows_already_in_order)
    assert prof.get("order_minmax", (0, 0))[1] == (2 if shape.startswith("outlier") or (shape == "sorted" and asc) else 1), prof


@pytest.mark.parametrize("shape", ["ascending_with_ties", "descending", "one_inversion_at_the_end", "one_inversion_between_samples"])
def test_order_rows_already_in_order(hip, oracle, shape):
    """ORDER BY over rows that arrive in the requested order is the identity (ties included): sampled neighbour pairs, then
    every pair; a single inversion anywhere must send the rows through the sort (no order_split launch otherwise)."""
    rng = np.random.default_rng(len(shape))
    k = np.sort(rng.integers(0, 1 << 20, N, dtype=np.int64))   # ~1.3 rows per value: ties
    asc = shape != "descending"
    if not asc:
        k = k[::-1].copy()
    if shape == "one_inversion_at_the_end":
        k[-1] = k[-2] - 1
    elif shape == "one_inversion_between_samples":
        k[700_003], k[700_004] = k[700_004] + 5, k[700_003]
    b = pa.RecordBatch.from_arrays([pa.array(k), pa.array(np.arange(N, dtype=np.int64)), pa.array(rng.random(N))], names=["k", "row", "x"])
    hip.profile(True)
    (got,) = list(OrderExecutor(hip, [OrderBy(InputRef(0), asc=asc)], [b.slice(0, N // 2), b.slice(N // 2)]).execute())
    prof = hip.profile_read()
    hip.profile(False)
    (exp,) = list(OrderExecutor(oracle, [OrderBy(InputRef(0), asc=asc)], [b]).execute())
    for i in range(3):
        assert got.column(i).equals(exp.column(i)), i
    sorted_anyway = prof.get("order_split", (0, 0))[1] > 0 or prof.get("radix_sort", (0, 0))[1] > 0
    assert sorted_anyway == shape.startswith("one_inversion"), prof


@pytest.mark.parametrize("shape", ["i64_random", "i64_many_ties", "f64", "i32_desc", "threshold_too_low", "two_payload_columns_desc", "two_keys"])
def test_order_by_limit_sorts_only_the_candidates(hip, oracle, shape, monkeypatch):
    """PhysicalLimit(PhysicalOrder(child)): with sqlrs_order_set_limit(offset + limit) the operator keeps the rows that can
    be among the first k (threshold from a sorted sample, ties included), sorts those and returns a prefix of the full
    result; LimitExecutor slices it.  Same rows as Limit(Order(..)) on the oracle, ties in input order.  `threshold_too_low`:
    the first rows of the column dominate the sample badly enough that fewer than k rows pass — the hint must be ignored."""
    monkeypatch.setenv("SQLRS_ORDER_TOPK", "1")  # (the size rule wants >= 2^20 rows and k <= rows / 16: forced at test size)
    rng = np.random.default_rng(len(shape))
    n, k, off = N, 1000, 37
    asc = "desc" not in shape
    if shape == "i64_many_ties":
        key = rng.integers(0, 50, n, dtype=np.int64)          # k-th row deep inside a run of equal keys
    elif shape == "f64":
        key = rng.standard_normal(n)
    elif shape == "i32_desc":
        key = rng.integers(-(1 << 30), 1 << 30, n).astype(np.int32)
    elif shape == "threshold_too_low":
        key = rng.integers(1 << 20, 1 << 30, n, dtype=np.int64)
        key[::max(1, n // 65536)] = -5                         # every sampled row is tiny, almost no other row is
        k = 200_000
    else:
        key = rng.integers(-(1 << 40), 1 << 40, n, dtype=np.int64)
    cols, names = [pa.array(key), pa.array(np.arange(n, dtype=np.int64))], ["k", "row"]
    if shape == "two_payload_columns_desc":
        cols.append(pa.array(rng.random(n), mask=rng.random(n) < 0.1))
        names.append("x")
    b = pa.RecordBatch.from_arrays(cols, names=names)
    ob = [OrderBy(InputRef(0), asc=asc)]
    if shape == "two_keys":  # ORDER BY k, row DESC LIMIT ..: threshold on the first key, the candidates sorted on both
        key = rng.integers(0, 2000, n, dtype=np.int64)
        b = pa.RecordBatch.from_arrays([pa.array(key), pa.array(np.arange(n, dtype=np.int64))], names=names)
        ob = [OrderBy(InputRef(0), asc=True), OrderBy(InputRef(1), asc=False)]
    ex = OrderExecutor(hip, ob, [b.slice(0, n // 3), b.slice(n // 3)], limit_hint=off + k)
    got = pa.Table.from_batches(list(LimitExecutor(hip, k, off, ex.execute()).execute()))
    exp = pa.Table.from_batches(list(LimitExecutor(oracle, k, off, OrderExecutor(oracle, ob, [b]).execute()).execute()))
    assert got.num_rows == k
    for i in range(len(names)):
        assert got.column(i).combine_chunks().equals(exp.column(i).combine_chunks()), names[i]
    if shape == "threshold_too_low":
        assert ex.topk_candidates == 0
    elif shape != "i64_many_ties":
        assert 0 < ex.topk_candidates < n // 4, ex.topk_candidates


def wide_keys_of(rng, shape, n):
    if shape == "i64_random":                 # 63 varying bits, about one row per value
        return rng.integers(-(1 << 62), 1 << 62, n, dtype=np.int64)
    if shape == "i64_33bit":                  # just beyond what the 32-bit word holds
        k = rng.integers(0, (1 << 32) + 5, n, dtype=np.int64) - 17
        k[n // 2], k[n // 3] = -17, (1 << 32) + 4 - 17
        return k
    if shape == "i64_ties":                   # 200 000 distinct wide values, ~6 rows each: runs ranked by position
        return rng.choice(rng.integers(-(1 << 60), 1 << 60, 200_000, dtype=np.int64), n)
    if shape == "f64_unit":                   # the image of a double is its exponent first: fixed bits would pile half the rows into one binade
        return rng.random(n)
    if shape == "f64_normal":
        return rng.normal(0.0, 1e3, n)
    if shape == "f64_lognormal_signed":       # 40 orders of magnitude, both signs, zeros of both signs
        k = np.exp(rng.normal(0.0, 15.0, n)) * rng.choice([-1.0, 1.0], n)
        k[::1000] = 0.0
        k[1::1000] = -0.0
        return k
    if shape == "f64_clusters":               # ~1000 tight clusters: runs of > OWK_WALK rows inside a group -> LSD over all bits
        c = rng.choice(rng.normal(0.0, 1e6, 1000), n)
        return c + rng.integers(0, 1 << 12, n) * 2.0 ** -30
    if shape == "i64_one_heavy_value":        # a value repeated far beyond the LDS finish: a run of equal splitters, a group of its own
        k = rng.integers(-(1 << 62), 1 << 62, n, dtype=np.int64)
        k[rng.random(n) < 0.02] = 123456789012345
        return k
    if shape == "f64_many_zeros":             # 30 % zeros of both signs (two heavy values next to each other in the order), the smallest key heavy too
        k = rng.normal(0.0, 1.0, n)
        u = rng.random(n)
        k[u < 0.2] = 0.0
        k[(u >= 0.2) & (u < 0.3)] = -0.0
        k[(u >= 0.3) & (u < 0.35)] = k.min()
        return k
    if shape == "i32_heavy_values":           # an int32 key (4-byte values out) whose heavy values the probe notices: the splitter route
        k = rng.integers(-(1 << 30), 1 << 30, n).astype(np.int32)
        u = rng.random(n)
        k[u < 0.3] = 12345
        k[(u >= 0.3) & (u < 0.4)] = -(1 << 30)
        return k
    if shape == "i64_50_distinct_wide":       # every group holds one value
        return rng.choice(rng.integers(-(1 << 62), 1 << 62, 50, dtype=np.int64), n)
    if shape == "i64_heavy_values_and_neighbours":   # heavy values whose neighbours v + 1, v + 2 exist too: the run's last group
        base = rng.integers(-(1 << 60), 1 << 60, 20, dtype=np.int64)
        k = rng.choice(base, n) + rng.choice(np.array([0, 0, 0, 0, 0, 0, 1, 2], dtype=np.int64), n)
        fill = rng.random(n) < 0.3              # ... among 30 % random keys
        k[fill] = rng.integers(-(1 << 62), 1 << 62, int(fill.sum()), dtype=np.int64)
        return k
    raise ValueError(shape)


@pytest.mark.parametrize("shape", ["i64_random", "i64_33bit", "i64_ties", "f64_unit", "f64_normal", "f64_lognormal_signed",
                                   "f64_clusters", "i64_one_heavy_value", "f64_many_zeros", "i64_50_distinct_wide",
                                   "i64_heavy_values_and_neighbours", "i32_heavy_values"])
@pytest.mark.parametrize("asc", [True, False])
@pytest.mark.parametrize("extra", ["none", "carry", "carry_and_more"])
def test_order_wide_keys(hip, oracle, shape, asc, extra):
    """keys with more than 32 varying bits: splitters from a sorted sample, two multi-split passes, in-LDS finish with the
    counting step (order_fast.hip, order_wide) — against the oracle, ties in input order; `order_knots` shows the route"""
    if extra != "carry" and shape not in ("i64_random", "f64_unit", "i64_ties", "f64_clusters"):
        pytest.skip("column mixes are crossed with four key shapes only")
    n = N if shape != "f64_normal" else 5_000_000
    rng = np.random.default_rng(hash_seed("wide", shape, asc, extra))
    k = wide_keys_of(rng, shape, n)
    cols, names = [pa.array(k)], ["k"]
    if extra != "none":
        cols.append(pa.array(np.arange(n, dtype=np.int64)))
        names.append("row")
    if extra == "carry_and_more":
        cols += [pa.array(rng.random(n), mask=rng.random(n) < 0.1), pa.array(rng.integers(0, 100, n).astype(np.int32))]
        names += ["f", "i"]
    b = pa.RecordBatch.from_arrays(cols, names=names)
    hip.profile(True)
    (got,) = list(OrderExecutor(hip, [OrderBy(InputRef(0), asc=asc)], [b.slice(0, n // 3), b.slice(n // 3)]).execute())
    prof = hip.profile_read()
    hip.profile(False)
    (exp,) = list(OrderExecutor(oracle, [OrderBy(InputRef(0), asc=asc)], [b]).execute())
    for i in range(b.num_columns):
        assert got.column(i).equals(exp.column(i)), names[i]
    assert prof.get("order_knots", (0, 0))[1] == 1, prof
    assert prof.get("order_finish", (0, 0))[1] == 1 and prof.get("gather", (0, 0))[1] <= (3 if extra == "carry_and_more" else 0), prof  # (the general path gathers every column)


@pytest.mark.parametrize("shape", ["i64_random", "f64_unit", "f64_lognormal_signed", "f64_clusters", "i64_33bit"])
@pytest.mark.parametrize("asc", [True, False])
def test_order_wide_keys_with_sampled_range(hip, oracle, shape, asc, monkeypatch):
    """columns of >= 2^24 rows sample their key range (forced here by SQLRS_ORDER_SAMPLE=1): a sampled range of more than
    32 bits goes to the wide route at once, with offsets from 0 over all 64 bits — no exact min / max pass (one
    order_minmax launch); a sampled range that fits 32 bits while the keys do not (i64_33bit) is found out by the first
    split pass and takes the exact pass first (two launches)"""
    monkeypatch.setenv("SQLRS_ORDER_SAMPLE", "1")
    rng = np.random.default_rng(hash_seed("wide-sampled", shape, asc))
    k = wide_keys_of(rng, shape, N)
    if shape == "i64_33bit":          # the two extremes sit in chunks the sample skips (it reads chunks 0, 16, 32, ... and the last rows)
        k[N // 2], k[N // 3] = 5, 6
        k[2048 * 3 + 1], k[2048 * 21 + 9] = -17, (1 << 32) + 4 - 17
    b = pa.RecordBatch.from_arrays([pa.array(k), pa.array(np.arange(N, dtype=np.int64))], names=["k", "row"])
    hip.profile(True)
    (got,) = list(OrderExecutor(hip, [OrderBy(InputRef(0), asc=asc)], [b]).execute())
    prof = hip.profile_read()
    hip.profile(False)
    (exp,) = list(OrderExecutor(oracle, [OrderBy(InputRef(0), asc=asc)], [b]).execute())
    assert got.column(0).equals(exp.column(0)) and got.column(1).equals(exp.column(1))
    assert prof.get("order_knots", (0, 0))[1] == 1, prof
    assert prof.get("order_minmax", (0, 0))[1] == (2 if shape == "i64_33bit" else 1), prof


@pytest.mark.parametrize("asc", [True, False])
@pytest.mark.parametrize("probe", [True, False])
def test_order_few_distinct_keys_spread_over_many_bits(hip, oracle, asc, probe, monkeypatch):
    """50 distinct keys over 2^26: a group of equal top bits holds tens of thousands of rows, more than the in-LDS finish
    takes.  The heavy-value probe (2048 sampled keys counted in LDS, travels with the key range) sends the call to the
    splitter route, where every value gets a group of its own that is copied (`order_knots`, two `order_split` launches).
    Without the probe (SQLRS_ORDER_HEAVY_PROBE=0, or a share too small for it) the attempt is thrown away after both split
    passes and the call is redone with every key bit through HBM passes (four here, then a streaming unpack): six
    `order_split` launches, no general path (`gather`) either way"""
    if not probe:
        monkeypatch.setenv("SQLRS_ORDER_HEAVY_PROBE", "0")
    rng = np.random.default_rng(50 + asc)
    k = keys_of(rng, "i64_heavy_groups")
    b = pa.RecordBatch.from_arrays([pa.array(k), pa.array(np.arange(N, dtype=np.int64)), pa.array(rng.random(N))], names=["k", "row", "x"])
    hip.profile(True)
    (got,) = list(OrderExecutor(hip, [OrderBy(InputRef(0), asc=asc)], [b]).execute())
    prof = hip.profile_read()
    hip.profile(False)
    (exp,) = list(OrderExecutor(oracle, [OrderBy(InputRef(0), asc=asc)], [b]).execute())
    for i in range(3):
        assert got.column(i).equals(exp.column(i)), i
    # (gathered by the permutation: the third column; on the splitter route the row ids travel, so the second one too — the
    #  two together as packed rows)
    # (the thrown-away attempt is ONE scope in the look-back form of the two split passes, two in the counting form)
    wasted = 2 if os.environ.get("SQLRS_ORDER_LB") == "0" else 1
    assert prof.get("order_knots", (0, 0))[1] == (1 if probe else 0) and prof.get("order_split", (0, 0))[1] == (2 if probe else wasted + 4), prof


def composite_case(rng, shape, n):
    """-> (key arrays most significant first, asc flags, composite route expected)"""
    if shape == "two_small_ranges":          # 10 + 14 bits: the narrow route
        return [rng.integers(0, 1000, n, dtype=np.int64), rng.integers(-5000, 5000, n, dtype=np.int64)], [True, True], True
    if shape == "int32_then_wide_int64":     # 7 + 40 bits: the splitter route
        return [rng.integers(-50, 50, n).astype(np.int32), rng.integers(0, 1 << 40, n, dtype=np.int64)], [False, True], True
    if shape == "three_keys_mixed_directions":
        return [rng.integers(0, 12, n).astype(np.int32), rng.integers(1990, 2025, n, dtype=np.int64),
                rng.integers(-(1 << 30), 1 << 30, n, dtype=np.int64)], [True, False, False], True
    if shape == "four_keys_with_a_constant":
        return [rng.integers(0, 3, n, dtype=np.int64), np.full(n, -7, dtype=np.int64), rng.integers(0, 200, n).astype(np.int32),
                rng.integers(0, 5, n, dtype=np.int64)], [True, True, False, True], True
    if shape == "ties_on_every_key":         # 40 distinct (a, b) pairs: the row-id column shows the input order inside a pair
        return [rng.integers(0, 8, n, dtype=np.int64), rng.integers(0, 5, n, dtype=np.int64)], [False, True], True
    if shape == "ranges_beyond_64_bits":     # 63 + 63 bits: general path
        return [rng.integers(-(1 << 62), 1 << 62, n, dtype=np.int64), rng.integers(-(1 << 62), 1 << 62, n, dtype=np.int64)], [True, True], False
    if shape == "exactly_64_bits":           # 32 + 32 bits
        a = rng.integers(0, 1 << 32, n, dtype=np.int64)
        b = rng.integers(-(1 << 31), 1 << 31, n, dtype=np.int64)
        a[0], a[1], b[2], b[3] = 0, (1 << 32) - 1, -(1 << 31), (1 << 31) - 1
        a[5:5000] = a[5]                     # ... with ties on the first key
        return [a, b], [True, False], True
    # a (values, mask) pair = a key with NULLs: NULLs first whatever the direction (order.rs:33-41), a valid bit in front of the value
    if shape == "one_key_with_nulls":
        return [(rng.integers(-(1 << 20), 1 << 20, n, dtype=np.int64), rng.random(n) < 0.07)], [True], True
    if shape == "one_key_with_nulls_desc_63_bits":   # 63 value bits + the valid bit: exactly one word, the splitter route
        return [(rng.integers(0, 1 << 63, n, dtype=np.int64, endpoint=False), rng.random(n) < 0.3)], [False], True
    if shape == "two_keys_both_with_nulls":
        return [(rng.integers(0, 50, n).astype(np.int32), rng.random(n) < 0.1),
                (rng.integers(-(1 << 33), 1 << 33, n, dtype=np.int64), rng.random(n) < 0.2)], [False, True], True
    if shape == "key_of_nulls_only_then_a_key":
        return [(np.zeros(n, dtype=np.int64), np.ones(n, dtype=bool)), rng.integers(0, 1 << 18, n, dtype=np.int64)], [True, False], True
    if shape == "f64_key_with_nulls":                # not an integer key: general path
        return [(rng.random(n), rng.random(n) < 0.1)], [True], False
    raise ValueError(shape)


@pytest.mark.parametrize("shape", ["two_small_ranges", "int32_then_wide_int64", "three_keys_mixed_directions", "four_keys_with_a_constant",
                                   "ties_on_every_key", "ranges_beyond_64_bits", "exactly_64_bits", "one_key_with_nulls",
                                   "one_key_with_nulls_desc_63_bits", "two_keys_both_with_nulls", "key_of_nulls_only_then_a_key",
                                   "f64_key_with_nulls"])
@pytest.mark.parametrize("extra", ["none", "carry", "carry_and_more"])
def test_order_by_several_integer_keys(hip, oracle, shape, extra):
    """ORDER BY a, b [, c, d] over plain integer columns: one composite key through the single-key routes, key columns decoded
    from the sorted composite (order_fast.hip, order_composite) — against the oracle's lexsort (order.rs:27-66), ties in
    input order; `order_split` launches show the route, `radix_sort` launches beyond the sample's the general path"""
    rng = np.random.default_rng(hash_seed("composite", shape, extra))
    keys, asc, composite = composite_case(rng, shape, N)
    cols = list(keys)
    names = [f"k{i}" for i in range(len(keys))]
    if extra != "none":
        cols.append(np.arange(N, dtype=np.int64))
        names.append("row")
    arrays = [pa.array(c[0], mask=c[1]) if isinstance(c, tuple) else pa.array(c) for c in cols]
    if extra == "carry_and_more":
        arrays += [pa.array(rng.random(N), mask=rng.random(N) < 0.1), pa.array([None if i % 13 == 0 else f"s{i % 89}" for i in range(N)])]
        names += ["f", "s"]
    # (the key columns are not the first columns of the table: k0 last)
    order = list(range(1, len(arrays))) + [0]
    b = pa.RecordBatch.from_arrays([arrays[i] for i in order], names=[names[i] for i in order])
    ob = [OrderBy(InputRef(order.index(i)), asc=asc[i]) for i in range(len(keys))]
    hip.profile(True)
    (got,) = list(OrderExecutor(hip, ob, [b.slice(0, N // 4), b.slice(N // 4)]).execute())
    prof = hip.profile_read()
    hip.profile(False)
    (exp,) = list(OrderExecutor(oracle, ob, [b]).execute())
    for i in range(b.num_columns):
        assert got.column(i).equals(exp.column(i)), b.schema.names[i]
    assert (prof.get("order_split", (0, 0))[1] > 0) == composite, prof


@pytest.mark.parametrize("shape", ["f64_asc", "f64_desc_more_columns", "i64_full_range", "f64_few_valid_rows"])
def test_order_one_key_with_nulls_split(hip, oracle, shape):
    """ORDER BY one key WITH NULLs that the composite key cannot take (a double; an int64 whose values span 64 bits): the NULL
    rows first in input order, the rest through an inner Order on the compacted rows (ops.hip, order_null_split) — against
    the oracle (order.rs:33-41: nulls first whatever the direction); `order_split` launches show the inner fast route"""
    n = 2_600_000
    rng = np.random.default_rng(hash_seed("nullsplit", shape))
    asc = shape != "f64_desc_more_columns"
    if shape == "i64_full_range":
        k = rng.integers(-(1 << 63), (1 << 63) - 1, n, dtype=np.int64, endpoint=True)
        k[7], k[8] = -(1 << 63), (1 << 63) - 1
    else:
        k = rng.normal(0.0, 100.0, n)
    mask = rng.random(n) < (0.9 if shape == "f64_few_valid_rows" else 0.15)   # True = NULL
    arrays = [pa.array(np.arange(n, dtype=np.int64)), pa.array(k, mask=mask)]
    names = ["row", "k"]
    if shape == "f64_desc_more_columns":
        arrays += [pa.array(rng.random(n), mask=rng.random(n) < 0.1), pa.array([None if i % 17 == 0 else f"s{i % 71}" for i in range(n)])]
        names += ["f", "s"]
    b = pa.RecordBatch.from_arrays(arrays, names=names)
    hip.profile(True)
    (got,) = list(OrderExecutor(hip, [OrderBy(InputRef(1), asc=asc)], [b.slice(0, n // 2), b.slice(n // 2)]).execute())
    prof = hip.profile_read()
    hip.profile(False)
    (exp,) = list(OrderExecutor(oracle, [OrderBy(InputRef(1), asc=asc)], [b]).execute())
    for i in range(b.num_columns):
        assert got.column(i).equals(exp.column(i)), names[i]
    split = shape != "f64_few_valid_rows"        # (260 000 valid rows: not worth the split, general path)
    assert (prof.get("order_split", (0, 0))[1] > 0) == split, prof


def test_order_heavy_values_with_all_bits_in_hbm_stay_on_the_narrow_route(hip, oracle):
    """1001 distinct keys over 10 bits, ~1300 rows each: every key bit goes through the HBM passes (no in-LDS finish, no limit on
    a group), so the heavy-value probe — which does see values with a visible share here — must not divert the call"""
    rng = np.random.default_rng(1001)
    k = keys_of(rng, "i64_10bit_many_ties")
    k[: N // 4] = 77                           # a quarter of the rows carry one value
    b = pa.RecordBatch.from_arrays([pa.array(k), pa.array(np.arange(N, dtype=np.int64))], names=["k", "row"])
    hip.profile(True)
    (got,) = list(OrderExecutor(hip, [OrderBy(InputRef(0), asc=True)], [b]).execute())
    prof = hip.profile_read()
    hip.profile(False)
    (exp,) = list(OrderExecutor(oracle, [OrderBy(InputRef(0), asc=True)], [b]).execute())
    assert got.equals(exp)
    assert prof.get("order_knots", (0, 0))[1] == 0 and prof.get("order_split", (0, 0))[1] == 2, prof


# (SQLRS_ORDER_FUZZ_EXTRA=N: N more seeds for a soak outside the suite)
@pytest.mark.parametrize("seed", range(24 + int(os.environ.get("SQLRS_ORDER_FUZZ_EXTRA", "0"))))
def test_fuzz_order_routes(hip, oracle, seed, monkeypatch):
    """random key type / width / distribution / heavy values / NULLs / key count / directions / column mix at sizes where the
    fast routes run (narrow, splitters, composite, NULL split, their fallbacks and the general path) — against the oracle"""
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.integers(1 << 20, 3 << 20))
    if rng.random() < 0.5:
        monkeypatch.setenv("SQLRS_ORDER_SAMPLE", "1")       # key range from a sample, as for >= 2^24 rows

    def key_column():
        kind = rng.choice(["i64", "i32", "f64"])
        bits = int(rng.integers(1, 64 if kind != "i32" else 32))
        if kind == "f64":
            v = rng.choice([rng.normal(0, 10.0 ** rng.integers(-3, 9), n), rng.random(n), np.exp(rng.normal(0, 12, n)) * rng.choice([-1.0, 1.0], n)])
        else:
            lo = int(rng.integers(-(1 << 62), 1 << 61)) if kind == "i64" else int(rng.integers(-(1 << 30), 1 << 29))
            hi = lo + (1 << bits) - 1
            cap = (1 << 63) - 1 if kind == "i64" else (1 << 31) - 1
            v = rng.integers(lo, min(hi, cap), n, dtype=np.int64, endpoint=True)
            v = v.astype(np.int32) if kind == "i32" else v
        shape = rng.choice(["random", "sorted", "reversed", "nearly_sorted"], p=[0.7, 0.1, 0.1, 0.1])
        if shape == "sorted":
            v = np.sort(v)
        elif shape == "reversed":
            v = np.sort(v)[::-1].copy()
        elif shape == "nearly_sorted":
            v = np.sort(v)
            i = rng.integers(0, n - 1, 5)
            v[i], v[i + 1] = v[i + 1].copy(), v[i].copy()
        for _ in range(int(rng.integers(0, 4))):             # heavy values
            v[rng.random(n) < rng.choice([0.001, 0.02, 0.3])] = v[int(rng.integers(0, n))]
        mask = (rng.random(n) < rng.choice([0.01, 0.2])) if rng.random() < 0.3 else None
        return pa.array(v, mask=mask)

    nk = int(rng.choice([1, 1, 1, 2, 3]))
    arrays = [key_column() for _ in range(nk)]
    names = [f"k{i}" for i in range(nk)]
    extra = int(rng.integers(0, 4))
    if extra >= 1:
        arrays.append(pa.array(np.arange(n, dtype=np.int64)))
        names.append("row")
    if extra >= 2:
        arrays.append(pa.array(rng.random(n), mask=(rng.random(n) < 0.1) if rng.random() < 0.5 else None))
        names.append("f")
    if extra >= 3:
        arrays.append(pa.array(rng.integers(0, 1000, n).astype(np.int32)))
        names.append("i")
    order = list(rng.permutation(len(arrays)))               # the key columns anywhere in the table
    b = pa.RecordBatch.from_arrays([arrays[i] for i in order], names=[names[i] for i in order])
    ob = [OrderBy(InputRef(order.index(i)), asc=bool(rng.random() < 0.5)) for i in range(nk)]
    cut = int(rng.integers(1, n - 1))
    (got,) = list(OrderExecutor(hip, ob, [b.slice(0, cut), b.slice(cut)]).execute())
    (exp,) = list(OrderExecutor(oracle, ob, [b]).execute())
    for i in range(b.num_columns):
        assert got.column(i).equals(exp.column(i)), (seed, b.schema.names[i])


@pytest.mark.parametrize("case", ["same_column_twice", "exactly_2p20_rows", "int32_desc_min_max", "second_key_decides_everything"])
def test_order_composite_edge_cases(hip, oracle, case):
    rng = np.random.default_rng(len(case))
    n = 1 << 20 if case == "exactly_2p20_rows" else N
    a = rng.integers(-1000, 1000, n, dtype=np.int64)
    b32 = rng.integers(-(1 << 31), (1 << 31) - 1, n, dtype=np.int64, endpoint=True).astype(np.int32)
    b32[3], b32[4] = -(1 << 31), (1 << 31) - 1
    row = np.arange(n, dtype=np.int64)
    if case == "same_column_twice":
        arrays, ob = [pa.array(a), pa.array(row)], [OrderBy(InputRef(0), asc=True), OrderBy(InputRef(0), asc=False)]
    elif case == "int32_desc_min_max":
        arrays, ob = [pa.array(b32), pa.array(a), pa.array(row)], [OrderBy(InputRef(0), asc=False), OrderBy(InputRef(1), asc=True)]
    elif case == "second_key_decides_everything":   # the first key is constant
        arrays, ob = [pa.array(np.full(n, 5, dtype=np.int64)), pa.array(a), pa.array(row)], [OrderBy(InputRef(0), asc=False), OrderBy(InputRef(1), asc=False)]
    else:
        arrays, ob = [pa.array(a), pa.array(b32), pa.array(row)], [OrderBy(InputRef(0), asc=True), OrderBy(InputRef(1), asc=True)]
    b = pa.RecordBatch.from_arrays(arrays, names=[f"c{i}" for i in range(len(arrays))])
    (got,) = list(OrderExecutor(hip, ob, [b]).execute())
    (exp,) = list(OrderExecutor(oracle, ob, [b]).execute())
    assert got.equals(exp)
