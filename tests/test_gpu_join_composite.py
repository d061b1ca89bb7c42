"""HashJoin on SEVERAL integer key columns (ON a.x = b.x AND a.y = b.y).

DEFAULT: the reference's rule — the key columns are hashed into 64 bits and rows are matched by hash alone
(hash_join.rs:161-232).  Its fold collides readily (a 1000 x 1000 grid of build keys: 3e6 probe rows find 3.71e6 "partners"
where 2.96e6 exist), and the library reproduces exactly that: same pairs, same order as the oracle.

OPT-IN (SQLRS_JOIN_COMPOSITE=1, read per build): exact equality on one composite key, sum_c (v_c - min_c) * stride_c over the
build side's ranges (join.hip, composite_build_keys) — dense ranges take the direct-address table, the rest the LDS route.
Checked against pandas' merge (the SQL answer), as row multisets: NULL keys never match."""
import numpy as np
import pyarrow as pa
import pytest

from sqlrs_amd.executor import HashJoinExecutor
from sqlrs_amd.expr import InputRef, JoinCondition

pytestmark = pytest.mark.gpu


def join_schema(lb, rb):
    return pa.schema([pa.field(f"l.{f.name}", f.type) for f in lb.schema] + [pa.field(f"r.{f.name}", f.type) for f in rb.schema])


def tables_equal(got, exp):
    g = pa.Table.from_batches(got) if got else None
    e = pa.Table.from_batches(exp) if exp else None
    if g is None or e is None:
        assert (g is None or g.num_rows == 0) and (e is None or e.num_rows == 0)
        return
    assert g.num_rows == e.num_rows, (g.num_rows, e.num_rows)
    for i in range(g.num_columns):
        assert g.column(i).combine_chunks().equals(e.column(i).combine_chunks()), g.schema.names[i]


def build_and_probe(rng, shape, nb, npb):
    """-> (build batch, probe batch, number of key columns, composite expected)"""
    def arr(v, mask=None):
        return pa.array(v, mask=mask)
    payload_b, payload_p = np.arange(nb, dtype=np.int64), rng.random(npb)
    if shape == "dense_grid":            # 1000 x 1000 values, every pair once: the direct-address table
        side = int(np.sqrt(nb))
        nb = side * side
        perm = rng.permutation(nb)
        bx, by = (perm // side).astype(np.int64), (perm % side).astype(np.int64) - 500
        px, py = rng.integers(-3, side + 3, npb, dtype=np.int64), rng.integers(-503, side - 497, npb, dtype=np.int64)
        payload_b = np.arange(nb, dtype=np.int64)
        return ([arr(bx), arr(by), arr(payload_b)], [arr(px), arr(py), arr(payload_p)], 2, True)
    if shape == "int32_int64_sparse":    # ranges 2^20 x 2^30: far from dense, the LDS route / the table of exact keys
        bx, by = rng.integers(0, 1 << 20, nb).astype(np.int32), rng.integers(-(1 << 29), 1 << 29, nb, dtype=np.int64)
        pick = rng.integers(0, nb, npb)
        px, py = bx[pick].copy(), by[pick].copy()
        miss = rng.random(npb) < 0.3
        py[miss] += 1                    # a neighbouring value in ONE column: no partner (almost surely)
        return ([arr(bx), arr(by), arr(payload_b)], [arr(px), arr(py), arr(payload_p)], 2, True)
    if shape == "duplicates_three_keys":  # 12 x 40 x 25 values, ~8 build rows per key
        bx, by, bz = rng.integers(0, 12, nb).astype(np.int32), rng.integers(100, 140, nb, dtype=np.int64), rng.integers(-5, 20, nb).astype(np.int32)
        px, py, pz = rng.integers(0, 13, npb).astype(np.int32), rng.integers(99, 141, npb, dtype=np.int64), rng.integers(-6, 20, npb).astype(np.int32)
        return ([arr(bx), arr(by), arr(bz), arr(payload_b)], [arr(px), arr(py), arr(pz), arr(payload_p)], 3, True)
    if shape == "probe_nulls":           # NULL probe keys never match (the build side has none)
        bx, by = rng.integers(0, 300, nb, dtype=np.int64), rng.integers(0, 300, nb, dtype=np.int64)
        px, py = rng.integers(0, 300, npb, dtype=np.int64), rng.integers(0, 300, npb, dtype=np.int64)
        return ([arr(bx), arr(by), arr(payload_b)], [arr(px, rng.random(npb) < 0.1), arr(py, rng.random(npb) < 0.1), arr(payload_p)], 2, True)
    if shape == "build_nulls":           # a NULL build key: the reference's hash rule decides (NULL leaves the hash unchanged)
        bx, by = rng.integers(0, 300, nb, dtype=np.int64), rng.integers(0, 300, nb, dtype=np.int64)
        px, py = rng.integers(0, 300, npb, dtype=np.int64), rng.integers(0, 300, npb, dtype=np.int64)
        return ([arr(bx, rng.random(nb) < 0.05), arr(by), arr(payload_b)], [arr(px, rng.random(npb) < 0.05), arr(py), arr(payload_p)], 2, False)
    if shape == "ranges_beyond_62_bits":
        bx, by = rng.integers(-(1 << 40), 1 << 40, nb, dtype=np.int64), rng.integers(-(1 << 40), 1 << 40, nb, dtype=np.int64)
        pick = rng.integers(0, nb, npb)
        return ([arr(bx), arr(by), arr(payload_b)], [arr(bx[pick]), arr(by[pick]), arr(payload_p)], 2, False)
    if shape == "key_types_differ":      # int32 on one side, int64 on the other: the reference's hashes differ by type
        bx, by = rng.integers(0, 100, nb).astype(np.int32), rng.integers(0, 100, nb, dtype=np.int64)
        px, py = rng.integers(0, 100, npb, dtype=np.int64), rng.integers(0, 100, npb, dtype=np.int64)
        return ([arr(bx), arr(by), arr(payload_b)], [arr(px), arr(py), arr(payload_p)], 2, True)
    raise ValueError(shape)


def run_join(be, lb, rb, nk, jt, ncols_left, split=True):
    cond = JoinCondition([(InputRef(i), InputRef(i)) for i in range(nk)])
    lbs = [lb.slice(0, lb.num_rows // 3), lb.slice(lb.num_rows // 3)] if split else [lb]   # two build batches: ranges over both
    rbs = [rb.slice(0, rb.num_rows // 2), rb.slice(rb.num_rows // 2)] if split else [rb]
    return list(HashJoinExecutor(be, lbs, rbs, jt, cond, join_schema(lb, rb), ncols_left).execute())


def batches_of(rng, shape):
    nb, npb = (1_000_000, 3_000_000) if shape in ("dense_grid", "int32_int64_sparse") else (40_000, 300_000)
    lcols, rcols, nk, composite = build_and_probe(rng, shape, nb, npb)
    lb = pa.RecordBatch.from_arrays(lcols, names=[f"c{i}" for i in range(len(lcols))])
    rb = pa.RecordBatch.from_arrays(rcols, names=[f"c{i}" for i in range(len(rcols))])
    return lb, rb, nk, composite


SHAPES = ["dense_grid", "int32_int64_sparse", "duplicates_three_keys", "probe_nulls", "build_nulls", "ranges_beyond_62_bits", "key_types_differ"]


@pytest.mark.parametrize("shape", ["dense_grid", "duplicates_three_keys", "probe_nulls", "build_nulls"])
@pytest.mark.parametrize("jt", ["inner", "full"])
def test_default_is_the_reference_rule(hip, oracle, shape, jt):
    """match by hash, collisions included: the oracle's pairs in the oracle's order"""
    rng = np.random.default_rng(len(shape) + len(jt))
    lb, rb, nk, _ = batches_of(rng, shape)
    got, exp = run_join(hip, lb, rb, nk, jt, lb.num_columns), run_join(oracle, lb, rb, nk, jt, lb.num_columns)
    tables_equal(got, exp)
    if shape == "dense_grid" and jt == "inner":   # (the reference's false matches are there)
        assert sum(b.num_rows for b in got) > rb.num_rows


def sql_answer(lb, rb, nk, jt):
    import pandas as pd
    l, r = lb.to_pandas(types_mapper=pd.ArrowDtype), rb.to_pandas(types_mapper=pd.ArrowDtype)
    l.columns = [f"l.{c}" for c in l.columns]
    r.columns = [f"r.{c}" for c in r.columns]
    lk, rk = [f"l.c{i}" for i in range(nk)], [f"r.c{i}" for i in range(nk)]
    how = {"inner": "inner", "left": "left"}[jt]
    if jt == "left":   # (sqlrs names: the build side is "left"; LEFT keeps every build row)
        pass
    # NULL keys never match in SQL; pandas would match NA with NA: take them out of the matching part
    lm, rm = l.dropna(subset=lk), r.dropna(subset=rk)
    if any(str(lm[a].dtype) != str(rm[b].dtype) for a, b in zip(lk, rk)):
        m = lm.iloc[:0].merge(rm.iloc[:0], left_on=lk, right_on=rk, how="inner")   # other integer type: nothing matches
    else:
        m = lm.merge(rm, left_on=lk, right_on=rk, how="inner")
    if how == "left":
        matched = set(m["l.c%d" % (lb.num_columns - 1)].tolist())   # the payload column of the build side is a row id
        rest = l[~l["l.c%d" % (lb.num_columns - 1)].isin(matched)]
        m = pd.concat([m, rest.reindex(columns=m.columns)], ignore_index=True)
    return m


def as_sorted_rows(table_or_df, names):
    import pandas as pd
    df = table_or_df if isinstance(table_or_df, pd.DataFrame) else table_or_df.to_pandas(types_mapper=pd.ArrowDtype)
    df = df[names]
    return df.sort_values(names, na_position="first", kind="stable").reset_index(drop=True)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("jt", ["inner", "left"])
def test_opt_in_exact_composite_key(hip, shape, jt, monkeypatch):
    if jt == "left" and shape not in ("dense_grid", "duplicates_three_keys", "probe_nulls"):
        pytest.skip("the outer variant is crossed with three shapes only")
    monkeypatch.setenv("SQLRS_JOIN_COMPOSITE", "1")
    rng = np.random.default_rng(len(shape) * 7 + len(jt))
    lb, rb, nk, composite = batches_of(rng, shape)
    hip.profile(True)
    got = run_join(hip, lb, rb, nk, jt, lb.num_columns)
    prof = hip.profile_read()
    hip.profile(False)
    names = [f.name for f in join_schema(lb, rb)]
    if not composite:   # build NULLs / ranges beyond 62 bits: the default rule runs — only the route is checked here
        assert prof.get("join_build_dense", (0, 0))[1] == 0
        return
    exp = sql_answer(lb, rb, nk, jt)
    g = pa.Table.from_batches(got) if got else None
    if g is None:
        assert len(exp) == 0
        return
    assert g.num_rows == len(exp), (g.num_rows, len(exp))
    a, b = as_sorted_rows(g, names), as_sorted_rows(exp, names)
    for c in names:
        assert a[c].isna().equals(b[c].isna()), c
        assert (a[c].fillna(0).to_numpy() == b[c].fillna(0).to_numpy()).all(), c
    if shape == "dense_grid":
        assert prof.get("join_build_dense", (0, 0))[1] >= 1, prof
