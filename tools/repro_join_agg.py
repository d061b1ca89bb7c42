import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np, pyarrow as pa
import sqlrs_amd
from sqlrs_amd import abi
from sqlrs_amd.executor import HashAggExecutor, HashJoinExecutor, HashJoinAggExecutor
from sqlrs_amd.expr import AggFunc, InputRef, JoinCondition
from oracle_backend import load_oracle
from test_gpu_parity import join_schema, rows_of
hip, oracle = sqlrs_amd.hip(0), load_oracle()
rng = np.random.default_rng(6012)
nb, npb, card = 3000, 2_300_000, 30000
_ = rng.choice([50, 3000, 40_000]); _ = rng.choice([70_000, 400_000, 2_300_000]); _ = rng.choice([1, 2, 10]); _ = rng.random()
lkeys = rng.permutation(card)[:nb].astype(np.int64)
lb = pa.RecordBatch.from_arrays([pa.array(lkeys), pa.array(rng.integers(0, 9, nb, dtype=np.int64))], names=["k", "x"])
_ = rng.choice([0.0, 0.04])
rb = pa.RecordBatch.from_arrays([pa.array(rng.integers(0, card, npb, dtype=np.int64)), pa.array(rng.random(npb))], names=["k", "v"])
cond = JoinCondition([(InputRef(0), InputRef(0))]); sch = join_schema(lb, rb)
aggs = [AggFunc("count", InputRef(3), abi.INT64), AggFunc("sum", InputRef(3), abi.FLOAT64)]
gb = [InputRef(0)]
cuts = [0, 83688, 867601, 2160843, npb]
for label, rbs in (("4 batches", [rb.slice(a, b - a) for a, b in zip(cuts[:-1], cuts[1:])]), ("1 batch", [rb])):
    ex = HashJoinAggExecutor(hip, [lb], rbs, cond, sch, 2, aggs, gb)
    got = rows_of(ex.execute())
    exp = rows_of(HashAggExecutor(oracle, aggs, gb, HashJoinExecutor(oracle, [lb], rbs, "inner", cond, sch, 2).execute()).execute())
    bad = [(g, e) for g, e in zip(got, exp) if g[:2] != e[:2]]
    print(label, "fused", ex.fused_batches, "groups", len(got), len(exp), "bad", len(bad), bad[:3], "total count", sum(g[1] for g in got), sum(e[1] for e in exp))
