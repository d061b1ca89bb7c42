import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np, pyarrow as pa
import sqlrs_amd
from sqlrs_amd import abi
from sqlrs_amd.executor import HashAggExecutor, HashJoinExecutor, HashJoinAggExecutor
from sqlrs_amd.expr import AggFunc, InputRef, JoinCondition
from oracle_backend import load_oracle
from test_gpu_parity import batch, join_schema, rows_of
hip, oracle = sqlrs_amd.hip(0), load_oracle()
for (nb, np_, nulls) in ((3000, 100_000, 0.05), (3000, 100_000, 0.0), (300, 5000, 0.05), (3000, 1_000, 0.05)):
    rng = np.random.default_rng(5)
    lb = batch(rng, nb, [("i64", 0.02, 0, 500), ("i64", 0.0, 0, 9)])
    rb = batch(rng, np_, [("i64", 0.02, 0, 700), ("f64", nulls, 0, 1)])
    cond = JoinCondition([(InputRef(0), InputRef(0))])
    sch = join_schema(lb, rb)
    aggs = [AggFunc("count", InputRef(3), abi.INT64), AggFunc("sum", InputRef(1), abi.INT64), AggFunc("sum", InputRef(3), abi.FLOAT64)]
    got = rows_of(HashJoinAggExecutor(hip, [lb], [rb], cond, sch, 2, aggs, [InputRef(0)]).execute())
    j = HashJoinExecutor(oracle, [lb], [rb], "inner", cond, sch, 2)
    exp = rows_of(HashAggExecutor(oracle, aggs, [InputRef(0)], j.execute()).execute())
    jh = HashJoinExecutor(hip, [lb], [rb], "inner", cond, sch, 2)
    got2 = rows_of(HashAggExecutor(hip, aggs, [InputRef(0)], jh.execute()).execute())
    def diff(g, e):
        if g[:3] != e[:3]: return True
        if g[3] is None or e[3] is None: return g[3] != e[3]
        return abs(g[3] - e[3]) > 1e-9 * abs(e[3])
    bad = [(g, e) for g, e in zip(got, exp) if diff(g, e)]
    bad2 = [(g, e) for g, e in zip(got2, exp) if diff(g, e)]
    print(nb, np_, nulls, "join_agg bad", len(bad), "of", len(exp), "| separate operators bad", len(bad2), bad[:1])
print("---- device-resident probe batch")
rng = np.random.default_rng(5)
lb = batch(rng, 300, [("i64", 0.02, 0, 500), ("i64", 0.0, 0, 9)])
rb = batch(rng, 5000, [("i64", 0.02, 0, 700), ("f64", 0.05, 0, 1)])
cond = JoinCondition([(InputRef(0), InputRef(0))])
sch = join_schema(lb, rb)
aggs = [AggFunc("count", InputRef(3), abi.INT64), AggFunc("sum", InputRef(1), abi.INT64), AggFunc("sum", InputRef(3), abi.FLOAT64)]
j = HashJoinExecutor(oracle, [lb], [rb], "inner", cond, sch, 2)
exp = rows_of(HashAggExecutor(oracle, aggs, [InputRef(0)], j.execute()).execute())
rbd = hip.to_device(rb)
got = rows_of(HashJoinAggExecutor(hip, [lb], [rbd], cond, sch, 2, aggs, [InputRef(0)]).execute())
print("join_agg device probe: same", got[:3] == exp[:3], got[:2], exp[:2])
jh = HashJoinExecutor(hip, [lb], [rbd], "inner", cond, sch, 2)
jo = list(jh.execute())
je = list(HashJoinExecutor(oracle, [lb], [rb], "inner", cond, sch, 2).execute())
print("join only, device probe: rows", sum(b.num_rows for b in jo), sum(b.num_rows for b in je), rows_of(jo)[:2], rows_of(je)[:2])
print("---- HashAgg over device-resident join output")
import inspect
sig = inspect.signature(HashJoinExecutor.__init__)
kw = {"out_mem": abi.MEM_DEVICE} if "out_mem" in sig.parameters else {}
jd = list(HashJoinExecutor(hip, [lb], [rb], "inner", cond, sch, 2, **kw).execute())
print("join out type", type(jd[0]).__name__, kw)
got3 = rows_of(HashAggExecutor(hip, aggs, [InputRef(0)], jd).execute())
print("agg over device join output: same", got3[:3] == exp[:3], got3[:2])
