import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np, pyarrow as pa
import sqlrs_amd
from sqlrs_amd import abi
from sqlrs_amd.executor import HashAggExecutor
from sqlrs_amd.expr import AggFunc, InputRef
from oracle_backend import load_oracle
hip, oracle = sqlrs_amd.hip(0), load_oracle()
rng = np.random.default_rng(1)
for n in (1000, 100_000):
    k = pa.array(rng.integers(0, 50, n, dtype=np.int64))
    v = pa.array(rng.random(n), mask=rng.random(n) < 0.05)
    b = pa.RecordBatch.from_arrays([k, v], names=["k", "v"])
    aggs = [AggFunc("count", InputRef(1), abi.INT64), AggFunc("sum", InputRef(1), abi.FLOAT64)]
    exp = list(HashAggExecutor(oracle, aggs, [InputRef(0)], [b]).execute())[0]
    for label, inp in (("host", [b]), ("device", [hip.to_device(b)]), ("device x2", [hip.to_device(b.slice(0, n // 2)), hip.to_device(b.slice(n // 2))])):
        got = list(HashAggExecutor(hip, aggs, [InputRef(0)], inp).execute())[0]
        ok = got.column(1).to_pylist() == exp.column(1).to_pylist() and np.allclose(got.column(2).to_numpy(zero_copy_only=False), exp.column(2).to_numpy(zero_copy_only=False), rtol=1e-9)
        print(n, label, "OK" if ok else "MISMATCH", got.column(2).to_pylist()[:3], exp.column(2).to_pylist()[:3])
