"""The all-core "fair" CPU baseline (oracle/cpu_fair.cpp) against the faithful oracle and the
numpy column generators — it is a timed baseline, so it must compute the same query."""
import numpy as np
import pyarrow as pa
import pytest

import cpu_fair
from sqlrs_amd import abi, datagen
from sqlrs_amd.executor import FilterExecutor, HashAggExecutor, HashJoinExecutor
from sqlrs_amd.expr import AggFunc, Constant, InputRef, JoinCondition


def test_generators_match_numpy_datagen():
    n_fact, n_dim = 100_003, 7_919
    fk, fv, dk = cpu_fair.gen_c5(n_fact, n_dim, fact_start=12345)
    idx = np.arange(12345, 12345 + n_fact, dtype=np.int64)
    assert np.array_equal(fk, datagen.key_np(0xF1, idx, n_dim))
    assert np.array_equal(fv, datagen.val_np(0xF2, idx))
    assert np.array_equal(dk, datagen.dim_key_np(np.arange(n_dim, dtype=np.int64), n_dim))


@pytest.mark.parametrize("n_fact,n_dim,threads", [(0, 10, 1), (50_000, 1000, 1), (300_000, 40_000, 0), (200_000, 3, 3)])
def test_fair_c5_matches_oracle(oracle, n_fact, n_dim, threads):
    fk, fv, dk = cpu_fair.gen_c5(n_fact, n_dim)
    rng = np.random.default_rng(n_fact + n_dim)
    if n_dim > 100:  # keys without partner, duplicate build keys
        fk = fk.copy()
        fk[rng.integers(0, max(n_fact, 1), n_fact // 10)] = n_dim + 17
        dk = dk.copy()
        dk[:5] = dk[5:10]
    keys, counts, sums, _ = cpu_fair.run_c5(fk, fv, dk, 0.5, threads)
    fact = pa.RecordBatch.from_arrays([pa.array(fk), pa.array(fv)], names=["key", "val"])
    dim = pa.RecordBatch.from_arrays([pa.array(dk)], names=["key"])
    schema = pa.schema([("d.key", pa.int64()), ("f.key", pa.int64()), ("f.val", pa.float64())])
    filt = FilterExecutor(oracle, InputRef(1) > Constant(0.5, abi.FLOAT64), [fact])
    join = HashJoinExecutor(oracle, [dim], filt.execute(), "inner", JoinCondition([(InputRef(0), InputRef(0))]), schema, 1)
    agg = HashAggExecutor(oracle, [AggFunc("count", InputRef(2), abi.INT64), AggFunc("sum", InputRef(2), abi.FLOAT64)],
                          [InputRef(0)], join.execute())
    (exp,) = list(agg.execute())
    ek = np.array(exp.column(0).to_pylist(), dtype=np.int64)
    ec = np.array(exp.column(1).to_pylist(), dtype=np.int64)
    es = np.array(exp.column(2).to_pylist(), dtype=np.float64)
    o1, o2 = np.argsort(keys, kind="stable"), np.argsort(ek, kind="stable")
    assert np.array_equal(keys[o1], ek[o2])
    assert np.array_equal(counts[o1], ec[o2])
    assert np.all(np.abs(sums[o1] - es[o2]) <= 1e-9 * np.maximum(np.abs(es[o2]), 1e-300))
