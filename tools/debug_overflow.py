import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np, pyarrow as pa
import sqlrs_amd
from sqlrs_amd import abi
from sqlrs_amd.executor import HashAggExecutor
from sqlrs_amd.expr import AggFunc, InputRef
from oracle_backend import load_oracle
hip, oracle = sqlrs_amd.hip(0), load_oracle()
rng = np.random.default_rng(21)
n = 2_400_000
keys = rng.integers(0, 20_000, n, dtype=np.int64)
keys[rng.random(n) < 0.6] = 7
b = pa.RecordBatch.from_arrays([pa.array(keys), pa.array(rng.random(n)), pa.array(rng.integers(-9, 9, n, dtype=np.int64))], names=["k", "v", "w"])
aggs = [AggFunc("count", InputRef(1), abi.INT64), AggFunc("sum", InputRef(1), abi.FLOAT64), AggFunc("min", InputRef(2), abi.INT64), AggFunc("sum", InputRef(2), abi.INT64)]
got = list(HashAggExecutor(hip, aggs, [InputRef(0)], [b]).execute())[0]
exp = list(HashAggExecutor(oracle, aggs, [InputRef(0)], [b]).execute())[0]
gk, ek = got.column(0).to_pylist(), exp.column(0).to_pylist()
print("groups", len(gk), len(ek), "same order", gk == ek, "same set", set(gk) == set(ek), "dups in got", len(gk) - len(set(gk)))
gd = {k: tuple(got.column(i)[j].as_py() for i in range(1, 5)) for j, k in enumerate(gk)}
ed = {k: tuple(exp.column(i)[j].as_py() for i in range(1, 5)) for j, k in enumerate(ek)}
bad = [(k, gd.get(k), ed[k]) for k in ed if gd.get(k) is None or gd[k][0] != ed[k][0] or gd[k][2:] != ed[k][2:] or abs(gd[k][1] - ed[k][1]) > 1e-9 * abs(ed[k][1])]
print("bad", len(bad), bad[:5])
first_diff = next((i for i, (x, y) in enumerate(zip(gk, ek)) if x != y), None)
print("first order diff at", first_diff, gk[first_diff - 1:first_diff + 3] if first_diff is not None else None, ek[first_diff - 1:first_diff + 3] if first_diff is not None else None)
