"""Direct-address HashJoin (hash_join.rs:146-253) over the BIT-PACKED table and the one-fetch build (join.hip, round 5).

The probe kernels read `bits` = ceil(log2(build rows + 1)) bits per possible key with one unaligned 4-byte load; the build
sizes the table for the largest range that takes the route and learns range, uniqueness and the NULL row in ONE fetch.
Everything here is compared with the oracle pair by pair (same rows, same order): build sizes either side of a bit-width
step, probe batches that are / are not multiples of the 512-row chunk of the all-hit kernel, a key column that is not
16-byte aligned, keys outside the range, a NULL build key, duplicates, ranges too wide for the route, both A/B hooks."""
import numpy as np
import pyarrow as pa
import pytest

from sqlrs_amd.executor import HashJoinExecutor
from sqlrs_amd.expr import InputRef, JoinCondition

pytestmark = pytest.mark.gpu


def schema_of(lb, rb):
    return pa.schema([pa.field(f"l.{f.name}", f.type) for f in lb.schema] + [pa.field(f"r.{f.name}", f.type) for f in rb.schema])


def run(be, lbs, rbs, jt="inner"):
    cond = JoinCondition([(InputRef(0), InputRef(0))])
    return list(HashJoinExecutor(be, lbs, rbs, jt, cond, schema_of(lbs[0], rbs[0]), lbs[0].num_columns).execute())


def same(got, exp):
    assert len(got) == len(exp)
    for g, e in zip(got, exp):
        assert g.num_rows == e.num_rows, (g.num_rows, e.num_rows)
        for i in range(g.num_columns):
            assert g.column(i).equals(e.column(i)), (i, g.schema.names[i])


def dim(rng, nb, kmin=0, stride=1, null_row=None):
    keys = kmin + stride * rng.permutation(nb).astype(np.int64)
    mask = None
    if null_row is not None:
        mask = np.zeros(nb, dtype=bool)
        mask[null_row] = True
    return pa.RecordBatch.from_arrays([pa.array(keys, mask=mask), pa.array(np.arange(nb, dtype=np.int64) * 7 + 1)], names=["k", "p"])


def fact(rng, n, lo, hi, null_frac=0.0):
    keys = rng.integers(lo, hi, n, dtype=np.int64)
    mask = (rng.random(n) < null_frac) if null_frac else None
    return pa.RecordBatch.from_arrays([pa.array(keys, mask=mask), pa.array(rng.random(n))], names=["k", "v"])


@pytest.mark.parametrize("nb", [200, 255, 256, 65_535, 65_536, 1_000_000])
def test_bit_width_steps_all_hit(hip, oracle, nb):
    """every probe row has a partner: the optimistic kernel over the packed table; 70 001 rows = 136 chunks + a tail"""
    rng = np.random.default_rng(nb)
    lb = dim(rng, nb, kmin=-17)
    for n in (70_001, 65_536 + 512):
        rb = fact(rng, n, -17, -17 + nb)
        hip.profile(True)
        got = run(hip, [lb], [rb])
        prof = hip.profile_read()
        hip.profile(False)
        same(got, run(oracle, [lb], [rb]))
        assert prof.get("join_build_dense", (0, 0))[1] == 1, prof
        assert got[0].num_rows == n


@pytest.mark.parametrize("jt", ["inner", "left", "right", "full"])
def test_misses_nulls_and_a_null_build_row(hip, oracle, jt):
    """keys either side of the range, NULL probe keys, one NULL build key (NULL = NULL matches, hash_utils.rs:91-104)"""
    rng = np.random.default_rng(11)
    nb = 300_000
    lb = dim(rng, nb, kmin=1000, null_row=1234)
    rb = fact(rng, 200_003, 900, 1000 + nb + 100, null_frac=0.01)
    same(run(hip, [lb], [rb], jt), run(oracle, [lb], [rb], jt))


def test_rare_miss_redoes_the_batch(hip, oracle):
    """one row without partner behind the sample: the attempt fails, the compacting kernel (same packed table) redoes it"""
    rng = np.random.default_rng(5)
    nb = 100_000
    lb = dim(rng, nb)
    keys = rng.integers(0, nb, 300_000, dtype=np.int64)
    keys[123_457] = nb + 5
    rb = pa.RecordBatch.from_arrays([pa.array(keys), pa.array(rng.random(len(keys)))], names=["k", "v"])
    rb2 = fact(rng, 100_000, 0, nb)  # (the join remembers the miss: later batches go straight to the compacting kernel)
    same(run(hip, [lb], [rb, rb2]), run(oracle, [lb], [rb, rb2]))


def test_key_column_not_16_byte_aligned(hip, oracle):
    """a probe batch sliced by one row: the key values start 8 bytes into their buffer (8-byte loads instead of 16-byte ones)"""
    rng = np.random.default_rng(6)
    nb = 50_000
    lb = dim(rng, nb)
    rb = fact(rng, 131_073 + 1, 0, nb).slice(1)
    same(run(hip, [lb], [rb]), run(oracle, [lb], [rb]))


def test_sparse_within_the_route_and_beyond(hip, oracle):
    """keys 3 apart (range = 3 x rows: still the table), 5 apart (beyond 4 slots per key: the general routes)"""
    rng = np.random.default_rng(7)
    nb = 120_000
    for stride, dense in ((3, True), (5, False)):
        lb = dim(rng, nb, kmin=-5000, stride=stride)
        rb = fact(rng, 150_000, -5000, -5000 + stride * nb)
        hip.profile(True)
        got = run(hip, [lb], [rb])
        prof = hip.profile_read()
        hip.profile(False)
        same(got, run(oracle, [lb], [rb]))
        # (the first probe runs its optimistic kernel against the build's device-side verdict either way — join_probe_dense —
        #  and the general hash table is built only when that verdict is "not a dense unique key set")
        assert prof.get("join_probe_dense", (0, 0))[1] >= 1, (stride, prof)
        assert (prof.get("join_build", (0, 0))[1] == 0) == dense, (stride, prof)


def test_duplicate_build_keys_leave_the_route(hip, oracle):
    rng = np.random.default_rng(8)
    nb = 90_000
    keys = rng.permutation(nb).astype(np.int64)
    keys[77] = keys[78]
    lb = pa.RecordBatch.from_arrays([pa.array(keys), pa.array(np.arange(nb, dtype=np.int64))], names=["k", "p"])
    rb = fact(rng, 100_000, 0, nb)
    same(run(hip, [lb], [rb]), run(oracle, [lb], [rb]))


def test_two_build_batches_and_all_null_keys(hip, oracle):
    rng = np.random.default_rng(9)
    lb = dim(rng, 80_000, kmin=5)
    rb = fact(rng, 70_000, 5, 80_005)
    same(run(hip, [lb.slice(0, 30_000), lb.slice(30_000)], [rb]), run(oracle, [lb.slice(0, 30_000), lb.slice(30_000)], [rb]))
    ln = pa.RecordBatch.from_arrays([pa.array([None, None, None], type=pa.int64()), pa.array([1, 2, 3], type=pa.int64())], names=["k", "p"])
    same(run(hip, [ln], [rb.slice(0, 100)]), run(oracle, [ln], [rb.slice(0, 100)]))


@pytest.mark.parametrize("var", ["SQLRS_DENSE_PACKED", "SQLRS_DENSE_BUILD_ONE_FETCH", "SQLRS_DENSE_BUILD_DEFER"])
def test_ab_hooks_give_the_same_pairs(hip, oracle, var, monkeypatch):
    """0 = the 4-byte table / the two-fetch build of round 4 / the build's verdict fetched by build_finish instead of by the
    first probe: same result"""
    rng = np.random.default_rng(10)
    lb = dim(rng, 400_000, kmin=-3, null_row=5)
    rb = fact(rng, 300_000, -3, 399_997)
    exp = run(oracle, [lb], [rb])
    for val in ("0", "1"):
        monkeypatch.setenv(var, val)
        same(run(hip, [lb], [rb]), exp)


@pytest.mark.parametrize("shape", ["all_hit", "half_hit", "duplicates", "too_sparse", "small_first_batch", "null_probe_keys"])
def test_first_probe_against_the_device_side_verdict(hip, oracle, shape):
    """build_finish leaves the direct-address build's verdict on the device; the first probe batch runs the optimistic
    all-hit kernel against it and fetches both answers together (join.hip, dense_resolve).  Every outcome of that pair —
    dense + all hit, dense + misses, duplicates, range too wide — and the first batches that cannot take the kernel (small,
    NULL keys: the verdict is fetched first), each followed by a second batch; pairs in hash_join.rs:217-253 order."""
    rng = np.random.default_rng(len(shape))
    nb = 150_000
    keys = rng.permutation(nb).astype(np.int64) * (7 if shape == "too_sparse" else 1)
    if shape == "duplicates":
        keys[5] = keys[99_999]
    lb = pa.RecordBatch.from_arrays([pa.array(keys), pa.array(np.arange(nb, dtype=np.int64))], names=["k", "p"])
    hi = int(keys.max()) + 1
    rb1 = fact(rng, 70_000 if shape != "small_first_batch" else 3_000, 0, hi * 2 if shape == "half_hit" else hi,
               null_frac=0.02 if shape == "null_probe_keys" else 0.0)
    rb2 = fact(rng, 80_000, 0, hi)
    for jt in ("inner", "left"):
        same(run(hip, [lb], [rb1, rb2], jt), run(oracle, [lb], [rb1, rb2], jt))
