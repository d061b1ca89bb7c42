"""Parity at the benchmark's own inputs and at BASELINE.json's full sizes.

* `test_bench_pipeline_matches_oracle_per_group`: the columns bench.py generates (same generator,
  same seeds), through bench.py's own `Pipeline` class (device resident, fused route), against the
  CPU oracle PER GROUP and in the oracle's group order — at a size the oracle finishes in seconds.
* the `full_size` tests run C2 / C3 / C4 / C5 at the sizes BASELINE.json states and check
  size-independent properties against plain torch ops on the same columns (torch is only the
  checker here; nothing of it is on the product path): filter = order-preserving selection,
  join pairs probe-major with equal keys on both sides, group-by counts = bincount, sums =
  index_add_ (1e-9 relative), groups in first-seen order.
"""
import ctypes as C
import os
import sys

import numpy as np
import pyarrow as pa
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(hip):
    import torch

    import bench
    from sqlrs_amd import abi, datagen
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    return type("Env", (), dict(torch=torch, bench=bench, abi=abi, datagen=datagen, dev=dev, be=hip))


def gen_c5(env, n_fact, n_dim, fact_mod=None):
    t, d = env.torch, env.datagen
    fk = d.fill_chunks(t.empty(n_fact, dtype=t.int64, device=env.dev), lambda i: d.key_t(0xF1, i, fact_mod or n_dim))
    fv = d.fill_chunks(t.empty(n_fact, dtype=t.float64, device=env.dev), lambda i: d.val_t(0xF2, i))
    dk = d.fill_chunks(t.empty(n_dim, dtype=t.int64, device=env.dev), lambda i: d.dim_key_t(i, n_dim))
    t.cuda.synchronize()
    return fk, fv, dk


def view(env, col, n, dtype):
    return env.bench._tensor_view(env.torch, col.values, n, dtype, env.dev)


def first_seen_rows(env, keys_of_rows, n_keys):
    """first_row[k] = smallest row index carrying key k (rows given in input order)"""
    t = env.torch
    first = t.full((n_keys,), 1 << 62, dtype=t.int64, device=env.dev)
    for lo in range(0, keys_of_rows.numel(), 1 << 27):
        k = keys_of_rows[lo:lo + (1 << 27)]
        first.scatter_reduce_(0, k, t.arange(lo, lo + k.numel(), dtype=t.int64, device=env.dev), reduce="amin")
    return first


# ------------------------------------------------------------------------------------------
def test_bench_pipeline_matches_oracle_per_group(env, oracle):
    t, abi, d = env.torch, env.abi, env.datagen
    n_fact, n_dim = 20_000_000, 1_000_000
    fk, fv, dk = gen_c5(env, n_fact, n_dim)
    # the device generator and the numpy generator (what the oracle is fed) agree bit for bit
    idx = np.arange(n_fact, dtype=np.int64)
    fk_np, fv_np = d.key_np(0xF1, idx, n_dim), d.val_np(0xF2, idx)
    dk_np = d.dim_key_np(np.arange(n_dim, dtype=np.int64), n_dim)
    assert np.array_equal(fk.cpu().numpy(), fk_np) and np.array_equal(fv.cpu().numpy(), fv_np)
    assert np.array_equal(dk.cpu().numpy(), dk_np)

    pipe = env.bench.Pipeline(env.be, abi, 0.5, fused=True)
    out = pipe.step(env.bench.device_batch(abi, [dk], [abi.INT64]),
                    env.bench.device_batch(abi, [fk, fv], [abi.INT64, abi.FLOAT64]))
    env.be.synchronize()
    assert pipe.fused_batches == 1
    g = out.num_rows
    gk = view(env, out.column(0), g, t.int64).cpu().numpy()
    gc = view(env, out.column(1), g, t.int64).cpu().numpy()
    gs = view(env, out.column(2), g, t.float64).cpu().numpy()
    out.release()

    from sqlrs_amd.executor import FilterExecutor, HashAggExecutor, HashJoinExecutor
    from sqlrs_amd.expr import AggFunc, Constant, InputRef, JoinCondition
    fact = pa.RecordBatch.from_arrays([pa.array(fk_np), pa.array(fv_np)], names=["key", "val"])
    dim = pa.RecordBatch.from_arrays([pa.array(dk_np)], names=["key"])
    schema = pa.schema([("d.key", pa.int64()), ("f.key", pa.int64()), ("f.val", pa.float64())])
    filt = FilterExecutor(oracle, InputRef(1) > Constant(0.5, abi.FLOAT64), [fact])
    join = HashJoinExecutor(oracle, [dim], filt.execute(), "inner", JoinCondition([(InputRef(0), InputRef(0))]), schema, 1)
    agg = HashAggExecutor(oracle, [AggFunc("count", InputRef(2), abi.INT64), AggFunc("sum", InputRef(2), abi.FLOAT64)],
                          [InputRef(0)], join.execute())
    (exp,) = list(agg.execute())
    ek = exp.column(0).to_numpy(zero_copy_only=False)
    ec = exp.column(1).to_numpy(zero_copy_only=False)
    es = exp.column(2).to_numpy(zero_copy_only=False)
    assert g == len(ek)
    assert np.array_equal(gk, ek)  # keys bit-exact AND in the reference's first-seen order
    assert np.array_equal(gc, ec)  # counts bit-exact
    assert np.all(np.abs(gs - es) <= 1e-9 * np.maximum(np.abs(es), 1e-300))  # SUM(double): 1e-9 relative


# ------------------------------------------------------------------------------ full sizes --
def test_full_size_c2_filter_with_null_predicate_rows(env):
    """SURVEY 8d's C2 variant with 10 % NULLs in the predicate column, VALUE-checked at 1e8 rows (review r05, weak #1: the bench
    leg only counts): rows whose v1 is NULL are dropped (filter.rs:16-24, arrow's filter on a NULL mask), the kept values come
    out in input order = torch's masked select over (valid & v1 > k)"""
    t, abi, d = env.torch, env.abi, env.datagen
    from sqlrs_amd.expr import Constant, InputRef
    n, k = 100_000_000, 1 << 30
    v1 = d.fill_chunks(t.empty(n, dtype=t.int64, device=env.dev), lambda i: d._lsr(d.splitmix64_t(0xC2, i), 33))
    nwords = (n + 63) // 64
    bits = t.zeros(nwords, dtype=t.int64, device=env.dev)
    for bit in range(64):  # bit set = valid with probability 0.9 (bench.py's generator)
        bits |= (d.val_t(0xC3 + bit, t.arange(nwords, dtype=t.int64, device=env.dev)) < 0.9).to(t.int64) << bit
    t.cuda.synchronize()
    col = abi.device_column(abi.INT64, n, v1.data_ptr(), validity_ptr=bits.data_ptr(), null_count=-1)
    b = abi.RawBatch([col], n, keepalive=[v1, bits])
    e = (InputRef(0) > Constant(k, abi.INT64)).pack()
    f = C.c_void_p()
    env.be.check(env.be.fn("filter_create")(env.be.ctx, C.byref(e.abi), C.byref(f)))
    o = C.POINTER(abi.Batch)()
    env.be.check(env.be.fn("filter_push")(f, b.ptr, abi.MEM_DEVICE, C.byref(o)))
    env.be.fn("filter_destroy")(f)
    out = env.be.wrap(o)
    env.be.synchronize()
    rows = t.arange(n, dtype=t.int64, device=env.dev)
    valid = ((bits[rows >> 6] >> (rows & 63)) & 1).bool()
    del rows
    exp = v1[valid & (v1 > k)]
    assert out.num_rows == exp.numel() and 0.44 * n < exp.numel() < 0.46 * n
    assert out.column(0).null_count == 0  # (every kept row had a valid predicate value)
    assert t.equal(view(env, out.column(0), out.num_rows, t.int64), exp)
    out.release()


@pytest.mark.parametrize("sel_k", [("0.5", 1 << 30), ("0.01", int((1 << 31) * 0.99)), ("0.99", int((1 << 31) * 0.01))])
def test_full_size_c2_filter(env, sel_k):
    """C2: SELECT v1 FROM t WHERE v1 > k over 1e8 int64 rows = torch's order-preserving masked select"""
    t, abi, d = env.torch, env.abi, env.datagen
    from sqlrs_amd.expr import Constant, InputRef
    n = 100_000_000
    _, k = sel_k
    v1 = d.fill_chunks(t.empty(n, dtype=t.int64, device=env.dev), lambda i: d._lsr(d.splitmix64_t(0xC2, i), 33))
    t.cuda.synchronize()
    e = (InputRef(0) > Constant(k, abi.INT64)).pack()
    f = C.c_void_p()
    env.be.check(env.be.fn("filter_create")(env.be.ctx, C.byref(e.abi), C.byref(f)))
    o = C.POINTER(abi.Batch)()
    env.be.check(env.be.fn("filter_push")(f, env.bench.device_batch(abi, [v1], [abi.INT64]).ptr, abi.MEM_DEVICE, C.byref(o)))
    env.be.fn("filter_destroy")(f)
    out = env.be.wrap(o)
    env.be.synchronize()
    exp = v1[v1 > k]
    assert out.num_rows == exp.numel()
    assert t.equal(view(env, out.column(0), out.num_rows, t.int64), exp)
    out.release()


@pytest.mark.parametrize("variant", ["all_hit", "half_hit", "sparse_keys"])
def test_full_size_c3_join_pairs(env, variant):
    """C3: 1e8 fact JOIN 1e6 dim (Inner), index-pair form: one pair per probe row that has a partner
    (unique build keys), probe-major order = right_idx strictly increasing, equal keys on both sides"""
    t, abi, d = env.torch, env.abi, env.datagen
    from sqlrs_amd.expr import InputRef
    nP, nB = 100_000_000, 1_000_000
    fk, _, dk = gen_c5(env, nP, nB, fact_mod=2 * nB if variant == "half_hit" else nB)
    if variant == "sparse_keys":
        A_s = 0x9E3779B97F4A7C15 - (1 << 64)
        fk.mul_(A_s).add_(12345)
        dk = dk * A_s + 12345
        t.cuda.synchronize()
    be = env.be
    lk, _k1 = abi.pack_exprs([InputRef(0)])
    rk, _k2 = abi.pack_exprs([InputRef(0)])
    rd = (C.c_int32 * 1)(abi.INT64)
    j = C.c_void_p()
    be.check(be.fn("hash_join_create")(be.ctx, abi.JOIN_INNER, 1, lk, rk, None, 1, rd, C.byref(j)))
    be.check(be.fn("hash_join_build_push")(j, env.bench.device_batch(abi, [dk], [abi.INT64]).ptr))
    be.check(be.fn("hash_join_build_finish")(j))
    o = C.POINTER(abi.Batch)()
    be.check(be.fn("hash_join_probe_indices")(j, env.bench.device_batch(abi, [fk], [abi.INT64]).ptr, abi.MEM_DEVICE, C.byref(o)))
    out = be.wrap(o)
    be.fn("hash_join_destroy")(j)
    be.synchronize()
    m = out.num_rows
    left = view(env, out.column(0), m, t.int64)
    right = view(env, out.column(1), m, t.int32).to(t.int64) & 0xffffffff
    in_dim = t.zeros(1, dtype=t.bool, device=env.dev)
    if variant == "half_hit":
        expected_m = int((fk < nB).sum().item())
    else:
        expected_m = nP
    assert m == expected_m
    assert bool((right[1:] > right[:-1]).all().item())          # probe-major, one pair per probe row
    assert bool((dk[left] == fk[right]).all().item())            # the pair joins equal keys
    if variant == "half_hit":                                    # exactly the probe rows with a partner
        assert t.equal(right, t.nonzero(fk < nB).flatten())
    del in_dim
    out.release()


@pytest.mark.parametrize("keys", ["sparse", "dense_range"])
@pytest.mark.parametrize("jt", ["inner", "full"])
def test_full_size_c3_join_duplicate_build_keys(env, jt, keys):
    """Round 6: 1e8 probe rows against 1e6 build rows whose sparse keys repeat ~4 times (Inner: every probe row matches ->
    ~4e8 pairs) and, for Full, a probe side half of whose keys have no partner — the LDS bucket-table route with the DISTINCT keys
    of the general table (join_lds.hip).  Checked on the device: pair count = sum of the build multiplicities of the probe keys
    (+ one pair per unmatched probe row), probe-row major, build insertion order inside a probe row, equal keys on both sides,
    NULL left index exactly on the unmatched rows (hash_join.rs:172-177, 225-248).  `dense_range`: the same multiplicities over keys
    that fill 0 .. 2.5e5 (bench.py's C3_join_dup_build_keys_x4) — runs by key and one direct-address lookup per probe row
    (join.hip: build_dense_dup), neither table."""
    t, abi, d = env.torch, env.abi, env.datagen
    from sqlrs_amd.expr import InputRef
    nP, nB, D = 100_000_000, 1_000_000, 250_000
    A_s = 0x9E3779B97F4A7C15 - (1 << 64)
    g = t.Generator(device=env.dev).manual_seed(44)
    bslot = t.randint(0, D, (nB,), dtype=t.int64, device=env.dev, generator=g)
    pslot = d.fill_chunks(t.empty(nP, dtype=t.int64, device=env.dev), lambda i: d.key_t(0xF1, i, D if jt == "inner" else 2 * D))
    dk, fk = (bslot * A_s + 12345, pslot * A_s + 12345) if keys == "sparse" else (bslot + 7, pslot + 7)
    mult = t.bincount(bslot, minlength=2 * D)
    per_row = mult[pslot]
    expect = int(per_row.sum().item()) + (int((per_row == 0).sum().item()) if jt == "full" else 0)
    del per_row
    t.cuda.synchronize()
    be = env.be
    lk, _k1 = abi.pack_exprs([InputRef(0)])
    rk, _k2 = abi.pack_exprs([InputRef(0)])
    rd = (C.c_int32 * 1)(abi.INT64)
    j = C.c_void_p()
    be.check(be.fn("hash_join_create")(be.ctx, abi.JOIN_INNER if jt == "inner" else abi.JOIN_FULL, 1, lk, rk, None, 1, rd, C.byref(j)))
    be.check(be.fn("hash_join_build_push")(j, env.bench.device_batch(abi, [dk], [abi.INT64]).ptr))
    be.check(be.fn("hash_join_build_finish")(j))
    be.profile(True)
    o = C.POINTER(abi.Batch)()
    be.check(be.fn("hash_join_probe_indices")(j, env.bench.device_batch(abi, [fk], [abi.INT64]).ptr, abi.MEM_DEVICE, C.byref(o)))
    prof = be.profile_read()
    be.profile(False)
    out = be.wrap(o)
    be.fn("hash_join_destroy")(j)
    be.synchronize()
    route = "join_match_unpermute" if keys == "sparse" else "join_probe_count_dense_dup"
    assert prof.get(route, (0, 0))[1] > 0 and prof.get("join_probe_count", (0, 0))[1] == 0, prof
    m = out.num_rows
    assert m == expect
    left = view(env, out.column(0), m, t.int64)
    right = view(env, out.column(1), m, t.int32).to(t.int64) & 0xffffffff
    assert bool((right[1:] >= right[:-1]).all().item())                              # probe-row major
    if jt == "inner":
        assert bool(((right[1:] != right[:-1]) | (left[1:] > left[:-1])).all().item())   # build insertion order inside a probe row
        assert bool((dk[left] == fk[right]).all().item())                                # the pair joins equal keys
    else:
        lv = out.column(0)
        assert lv.null_count == int((mult[pslot] == 0).sum().item())
        vb = env.bench._tensor_view(t, lv.validity, (m + 63) // 64, t.int64, env.dev)  # (device bitmaps are whole 64-bit words)
        valid = ((vb[:, None] >> t.arange(64, device=env.dev, dtype=t.int64)[None, :]) & 1).flatten()[:m].to(t.bool)
        assert bool((valid == (mult[pslot[right]] > 0)).all().item())                    # NULL exactly where the probe key has no partner
        lm, rm = left[valid], right[valid]
        assert bool((dk[lm] == fk[rm]).all().item())
        assert bool(((rm[1:] != rm[:-1]) | (lm[1:] > lm[:-1])).all().item())
        del valid, lm, rm, vb
    out.release()


def _run_hash_agg(env, key, val, where=False):
    abi, be = env.abi, env.be
    from sqlrs_amd.expr import AggFunc, Constant, InputRef
    gb, _k = abi.pack_exprs([InputRef(0)])
    keep = []
    aggs = (abi.AggFunc * 2)(AggFunc("count", InputRef(1), abi.INT64).abi_struct(keep),
                             AggFunc("sum", InputRef(1), abi.FLOAT64).abi_struct(keep))
    a = C.c_void_p()
    be.check(be.fn("hash_agg_create")(be.ctx, 1, gb, 2, aggs, C.byref(a)))
    if where:  # WHERE val > 0.5 handed to the aggregate: evaluated inside the first partition pass
        pred = (InputRef(1) > Constant(0.5, abi.FLOAT64)).pack()
        be.check(be.fn("hash_agg_set_filter")(a, C.byref(pred.abi)))
    be.check(be.fn("hash_agg_push")(a, env.bench.device_batch(abi, [key, val], [abi.INT64, abi.FLOAT64]).ptr))
    o = C.POINTER(abi.Batch)()
    be.check(be.fn("hash_agg_finish")(a, abi.MEM_DEVICE, C.byref(o)))
    be.fn("hash_agg_destroy")(a)
    be.synchronize()
    return be.wrap(o)


@pytest.mark.parametrize("keys", ["uniform", "sparse", "sorted", "sorted_sparse", "where_sparse"])
def test_full_size_c4_group_by(env, keys):
    """C4: 2e8 rows, 1e6 int64 groups, COUNT + SUM(f64): counts = bincount (bit-exact), sums =
    index_add_ (1e-9 relative), every key once, groups in first-seen order (hash_agg.rs:98,132).
    `sorted*`: the rows ordered by key — a sample of the ROWS is then no sample of the GROUPS (an eighth of the chunks
    holds an eighth of the groups): the statistics pass has to notice (key_stats_kernel's two register sets), or the
    bucket tables are sized for 1.5e5 groups and the batch takes the overflow path (250 ms instead of ~3; seen round 5)"""
    t, d = env.torch, env.datagen
    n, G = 200_000_000, 1_000_000
    key = d.fill_chunks(t.empty(n, dtype=t.int64, device=env.dev), lambda i: d.key_t(0xA1, i, G))
    if keys.startswith("sorted"):
        key = t.sort(key).values
    val = d.fill_chunks(t.empty(n, dtype=t.float64, device=env.dev), lambda i: d.val_t(0xF2, i))
    t.cuda.synchronize()
    where = keys.startswith("where")  # (round 6: general keys + a fused WHERE: the unpacked first level counts level 2's digits itself)
    if where:
        keep = val > 0.5
        kk, vk = key[keep], val[keep]
        exp_cnt = t.bincount(kk, minlength=G)
        exp_sum = t.zeros(G, dtype=t.float64, device=env.dev).index_add_(0, kk, vk)
        first = t.full((G,), 1 << 62, dtype=t.int64, device=env.dev)
        first.scatter_reduce_(0, kk, t.nonzero(keep).flatten(), reduce="amin")
        del keep, kk, vk
    else:
        exp_cnt = t.bincount(key, minlength=G)
        exp_sum = t.zeros(G, dtype=t.float64, device=env.dev).index_add_(0, key, val)
        first = first_seen_rows(env, key, G)
    A_s, A_inv = 0x9E3779B97F4A7C15 - (1 << 64), pow(0x9E3779B97F4A7C15, -1, 1 << 64)
    A_inv_s = A_inv - (1 << 64) if A_inv >= (1 << 63) else A_inv
    if keys.endswith("sparse"):  # hashed buckets instead of the key-range partition
        key.mul_(A_s).add_(777)
        t.cuda.synchronize()
    env.be.profile(True)
    out = _run_hash_agg(env, key, val, where=where)
    prof = env.be.profile_read()
    env.be.profile(False)
    if keys in ("sparse", "where_sparse"):
        assert "rp_hist" not in prof and ("rp_chunk_scatter" in prof or "rp_chunk_scatter_filter" in prof), sorted(prof)  # level 2's counts came from level 1
    if keys.startswith("sorted"):
        assert "agg_resolve" not in prof and "agg_update" not in prof, sorted(prof)  # no row took the overflow path
    g = out.num_rows
    gk = view(env, out.column(0), g, t.int64)
    if keys.endswith("sparse"):
        gk = (gk - 777) * A_inv_s
    gc, gs = view(env, out.column(1), g, t.int64), view(env, out.column(2), g, t.float64)
    assert g == int((exp_cnt > 0).sum().item())
    assert bool(((gk >= 0) & (gk < G)).all().item())
    assert int(t.bincount(gk, minlength=G).max().item()) == 1                      # every group once
    assert t.equal(gc, exp_cnt[gk])                                                # COUNT bit-exact
    e = exp_sum[gk]
    assert bool(((gs - e).abs() <= 1e-9 * e.abs().clamp_min(1e-300)).all().item())  # SUM within 1e-9 relative
    fr = first[gk]
    assert bool((fr[1:] > fr[:-1]).all().item())                                   # first-seen order
    out.release()


def test_full_size_c5_pipeline(env):
    """C5 on one GPU at BASELINE.json's size (1e9 fact x 1e7 dim, 1e7 groups), bench.py's Pipeline,
    every group checked (not just totals) + first-seen order of the groups"""
    t, abi = env.torch, env.abi
    n_fact, n_dim = 1_000_000_000, 10_000_000
    fk, fv, dk = gen_c5(env, n_fact, n_dim)
    exp_cnt, exp_sum, has_dim, kept = env.bench.expected_groups(t, None, fk, fv, dk, 0.5, n_dim)
    pipe = env.bench.Pipeline(env.be, abi, 0.5, fused=True)
    out = pipe.step(env.bench.device_batch(abi, [dk], [abi.INT64]),
                    env.bench.device_batch(abi, [fk, fv], [abi.INT64, abi.FLOAT64]))
    env.be.synchronize()
    ok, groups, rows, msg = env.bench.check_groups(t, None, env.dev, out, exp_cnt, exp_sum, has_dim)
    assert ok, msg
    assert rows == kept
    # first-seen order over the JOIN OUTPUT = over the kept fact rows in input order
    first = t.full((n_dim,), 1 << 62, dtype=t.int64, device=env.dev)
    for lo in range(0, n_fact, 1 << 27):
        k, v = fk[lo:lo + (1 << 27)], fv[lo:lo + (1 << 27)]
        m = v > 0.5
        first.scatter_reduce_(0, k[m], t.arange(lo, lo + k.numel(), dtype=t.int64, device=env.dev)[m], reduce="amin")
    gk = view(env, out.column(0), out.num_rows, t.int64)
    fr = first[gk]
    assert bool((fr[1:] > fr[:-1]).all().item())
    out.release()
    env.be.fn("ctx_pool_trim")(env.be.ctx)


@pytest.mark.parametrize("shape", ["sorted_keys", "sorted_keys_mixed_magnitudes", "sorted_keys_per_run_adds_off", "sorted_keys_nearly",
                                   "sorted_keys_one_lane_rank_off", "hot_digit", "hot_digit_one_lane_rank_eager", "skewed",
                                   "every_row_passes", "few_rows_pass", "keys_beyond_the_dim"])
def test_slim_records_at_scale(env, shape, monkeypatch):
    """The slim-record route (radix_part.hip; 768-thread workgroups over 6144-row tiles) at 2^28 rows, on inputs that bend its
    bookkeeping: sorted keys (a tile is ONE digit: chunks fill and spill every tile, every other digit's chunk is closed
    early over and over), one hot digit (most rows in 1/23 of the key range: split bucket work items, long chunk lists),
    skewed keys (hot slots, split buckets), a predicate that keeps everything / almost nothing (full staging area / runs of a row or two),
    and probe keys far beyond the build range (dropped by level 1).  `*_one_lane_rank_*` (SQLRS_RP_CONC, read per call): level 1's
    gated one-lane rank never / already behind a tile with an eighth of its row slots in one digit (mixed waves: the first lane's
    bucket by one lane, the other lanes by themselves).  Every group against torch (COUNT bit-exact,
    SUM 1e-9), groups in first-seen order, and the route really is the slim one."""
    t, abi, d = env.torch, env.abi, env.datagen
    n_fact, n_dim = (1 << 28) + 12_345, 3_000_000
    fk, fv, dk = gen_c5(env, n_fact, n_dim)
    thr = 0.5
    if shape.startswith("sorted_keys"):
        # (ordered rows reach a bucket as a few long runs: the bucket pass adds a wave's rows per run of equal keys, a
        #  SEGMENTED scan of the values — neighbouring keys of 1e+12 and 1e-6 magnitudes must not leak into each other's
        #  rounding; SQLRS_AGG_SEG=0, read per call: one atomic per row as for random rows)
        fk = t.sort(fk).values
        if shape == "sorted_keys_mixed_magnitudes":
            fv = t.where(fk % 2 == 0, fv * 1e12 + 1.0, fv * 1e-6 + 0.6e-6)   # (> thr = 0.5e-6 ... see below)
        if shape == "sorted_keys_per_run_adds_off":
            monkeypatch.setenv("SQLRS_AGG_SEG", "0")
        if shape == "sorted_keys_one_lane_rank_off":
            monkeypatch.setenv("SQLRS_RP_CONC", "9")
        if shape == "sorted_keys_nearly":
            # 1 % of the rows swapped with a row up to 300 positions on: a neighbouring key's first row now lies INSIDE this
            # key's first rows, and the partition levels rank a tile's rows of one digit in LDS-atomic order (two waves
            # interleave) — the first row of a run of equal keys is the minimum over the run, not its head lane's
            g = t.Generator(device=env.dev).manual_seed(5)
            p_ = t.randint(0, n_fact - 301, (n_fact // 100,), device=env.dev, generator=g)
            q_ = p_ + t.randint(1, 300, (n_fact // 100,), device=env.dev, generator=g)
            a_, b_ = fk[p_].clone(), fk[q_].clone()
            fk[p_] = b_
            fk[q_] = a_
            del p_, q_, a_, b_
    elif shape.startswith("hot_digit"):
        if shape == "hot_digit_one_lane_rank_eager":
            monkeypatch.setenv("SQLRS_RP_CONC", "1")
        hot = (fv * 7919.0).frac() < 0.9  # (a second stream of pseudo-random bits: independent of the predicate on fv > 0.5)
        fk = t.where(hot, 1_000_000 + fk % 100_000, fk)
    elif shape == "skewed":
        # key = floor(u^2 * n_dim): density ~ 1 / sqrt(key), the hottest keys carry ~1e5 rows each, the buckets of the low
        # key range several times the average (split work items).  (Zipf keys are C4's leg: the expectation below — torch
        # index_add_ of doubles — crawls when 8 % of the rows hit one address)
        u = d.fill_chunks(t.empty(n_fact, dtype=t.float64, device=env.dev), lambda i: d.val_t(0xA7, i))
        fk = (u * u * n_dim).to(t.int64).clamp_(max=n_dim - 1)
        del u
    elif shape == "every_row_passes":
        thr = -1.0
    if shape == "sorted_keys_mixed_magnitudes":
        thr = 0.8e-6  # keeps every row of the even keys and about four in five of the odd keys' rows
    elif shape == "few_rows_pass":
        thr = 0.9995
    elif shape == "keys_beyond_the_dim":
        fk = fk * 3  # two thirds of the probe rows have no build partner, up to 3x the build key range
    t.cuda.synchronize()
    n_keys = int(fk.max().item()) + 1
    exp_cnt, exp_sum, has_dim, kept = env.bench.expected_groups(t, None, fk, fv, dk, thr, max(n_keys, n_dim))
    pipe = env.bench.Pipeline(env.be, abi, thr, fused=True)
    env.be.profile(True)
    out = pipe.step(env.bench.device_batch(abi, [dk], [abi.INT64]), env.bench.device_batch(abi, [fk, fv], [abi.INT64, abi.FLOAT64]))
    env.be.synchronize()
    prof = env.be.profile_read()
    env.be.profile(False)
    assert pipe.fused_batches == 1 and pipe.filter_fused_batches == 1
    assert prof.get("rp_slim_runs", (0, 0))[1] > 0, prof  # the slim route ran
    ok, groups, rows, msg = env.bench.check_groups(t, None, env.dev, out, exp_cnt, exp_sum, has_dim)
    assert ok, msg
    first = t.full((max(n_keys, n_dim),), 1 << 62, dtype=t.int64, device=env.dev)
    for lo in range(0, n_fact, 1 << 27):
        k, v = fk[lo:lo + (1 << 27)], fv[lo:lo + (1 << 27)]
        m = v > thr
        first.scatter_reduce_(0, k[m], t.arange(lo, lo + k.numel(), dtype=t.int64, device=env.dev)[m], reduce="amin")
    gk = view(env, out.column(0), out.num_rows, t.int64)
    fr = first[gk]
    assert bool((fr[1:] > fr[:-1]).all().item())  # first-seen order (hash_agg.rs:87-99)
    out.release()
    env.be.fn("ctx_pool_trim")(env.be.ctx)


@pytest.mark.parametrize("push", ["order_push", "order_push_retained"])
def test_full_size_order(env, push):
    """bench's Order shape: ORDER BY v1 (int64, 31 significant bits) carrying one f64 column over 1e8 device rows =
    torch's stable sort of the key + the gathered column (ties keep input order, order.rs:27-66); the retained push
    (caller keeps the batch alive, nothing copied) and the copying push give the same batch"""
    t, abi, d = env.torch, env.abi, env.datagen
    from sqlrs_amd.expr import InputRef
    n = 100_000_000
    v1 = d.fill_chunks(t.empty(n, dtype=t.int64, device=env.dev), lambda i: d._lsr(d.splitmix64_t(0xC2, i), 33))
    val = d.fill_chunks(t.empty(n, dtype=t.float64, device=env.dev), lambda i: d.val_t(0xF2, i))
    t.cuda.synchronize()
    bo = env.bench.device_batch(abi, [v1, val], [abi.INT64, abi.FLOAT64])
    pk = InputRef(0).pack()
    obs = (abi.OrderBy * 1)(abi.OrderBy(pk.abi, 1, 0))
    h = C.c_void_p()
    env.be.check(env.be.fn("order_create")(env.be.ctx, 1, obs, C.byref(h)))
    env.be.check(env.be.fn(push)(h, bo.ptr))
    o = C.POINTER(abi.Batch)()
    env.be.check(env.be.fn("order_finish")(h, abi.MEM_DEVICE, C.byref(o)))
    env.be.fn("order_destroy")(h)
    out = env.be.wrap(o)
    env.be.synchronize()
    assert out.num_rows == n
    gk, gv = view(env, out.column(0), n, t.int64), view(env, out.column(1), n, t.float64)
    ek, perm = t.sort(v1, stable=True)
    assert t.equal(gk, ek)
    del ek
    assert t.equal(gv, val[perm])  # bit-exact, ties in input order
    out.release()


@pytest.mark.parametrize("shape", ["f64_normal_2p28", "i64_63bit_desc", "two_keys_one_with_nulls"])
def test_full_size_order_wide_and_composite_keys(env, shape):
    """the splitter route (keys of more than 32 varying bits) and the composite route (several integer keys, NULLs first) at
    and beyond bench size, against torch's stable sorts (ties keep input order, order.rs:27-66); 2^28 + 12345 rows: every
    position of the 32-bit offset arithmetic past 2^28, ragged last tile"""
    t, abi = env.torch, env.abi
    from sqlrs_amd.expr import InputRef
    g = t.Generator(device=env.dev).manual_seed(len(shape))
    n = (1 << 28) + 12345 if shape == "f64_normal_2p28" else 100_000_000
    row = t.arange(n, dtype=t.int64, device=env.dev)
    keep = []
    if shape == "f64_normal_2p28":
        k = t.randn(n, dtype=t.float64, device=env.dev, generator=g)
        cols, types, asc = [k, row], [abi.FLOAT64, abi.INT64], [1]
        perm = t.sort(k, stable=True).indices
    elif shape == "i64_63bit_desc":
        k = t.randint(-(1 << 62), 1 << 62, (n,), dtype=t.int64, device=env.dev, generator=g)
        k[::7] = k[3]  # ties: a seventh of the rows share one key
        cols, types, asc = [k, row], [abi.INT64, abi.INT64], [0]
        perm = t.sort(k, stable=True, descending=True).indices
    else:
        a = t.randint(0, 300, (n,), dtype=t.int64, device=env.dev, generator=g)
        b = t.randint(-(1 << 34), 1 << 34, (n,), dtype=t.int64, device=env.dev, generator=g)
        a_null = t.rand(n, device=env.dev, generator=g) < 0.05
        cols, types, asc = [a, b, row], [abi.INT64, abi.INT64, abi.INT64], [0, 1]
        p1 = t.sort(b, stable=True).indices                       # last key first
        # a DESC with NULLs first: NULL rows get a key above every value
        a_key = t.where(a_null, t.full_like(a, 1000), a)[p1]
        perm = p1[t.sort(a_key, stable=True, descending=True).indices]
        del p1, a_key
    rb = env.bench.device_batch(abi, cols, types)
    if shape == "two_keys_one_with_nulls":  # validity bitmap of `a` (bit set = valid), LSB first
        bits = (~a_null).view(t.uint8)
        pad = (-n) % 64
        if pad:
            bits = t.cat([bits, t.zeros(pad, dtype=t.uint8, device=env.dev)])
        w = (bits.view(-1, 8).to(t.int32) << t.arange(8, device=env.dev, dtype=t.int32)).sum(1).to(t.uint8)
        col0 = abi.device_column(abi.INT64, n, a.data_ptr(), validity_ptr=w.data_ptr(), null_count=int(a_null.sum().item()))
        rb = abi.RawBatch([col0] + [abi.device_column(abi.INT64, n, c.data_ptr()) for c in cols[1:]], n, keepalive=cols + [w])
        keep.append(w)
    packs = [InputRef(i).pack() for i in range(len(asc))]
    obs = (abi.OrderBy * len(asc))(*[abi.OrderBy(p.abi, a_, 0) for p, a_ in zip(packs, asc)])
    h = C.c_void_p()
    env.be.check(env.be.fn("order_create")(env.be.ctx, len(asc), obs, C.byref(h)))
    env.be.check(env.be.fn("order_push_retained")(h, rb.ptr))
    o = C.POINTER(abi.Batch)()
    env.be.check(env.be.fn("order_finish")(h, abi.MEM_DEVICE, C.byref(o)))
    env.be.fn("order_destroy")(h)
    out = env.be.wrap(o)
    env.be.synchronize()
    assert out.num_rows == n
    got_row = view(env, out.column(len(cols) - 1), n, t.int64)
    assert t.equal(got_row, perm)                      # the permutation itself: ties in input order, NULLs first
    if shape == "two_keys_one_with_nulls":
        nn = int(a_null.sum().item())
        assert t.equal(view(env, out.column(1), n, t.int64), b[perm])
        assert t.equal(view(env, out.column(0), n, t.int64)[nn:], a[perm][nn:]) and bool(a_null[perm][:nn].all().item())
        assert out.column(0).null_count == nn
    else:
        assert t.equal(view(env, out.column(0), n, cols[0].dtype), cols[0][perm])
    out.release()
    env.be.fn("ctx_pool_trim")(env.be.ctx)
