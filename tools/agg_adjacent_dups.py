import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import numpy as np, pyarrow as pa
from opshapes_common import be, D, abi, dev, drain, timed
from sqlrs_amd.expr import AggFunc, InputRef
from sqlrs_amd.executor import HashAggExecutor
rng = np.random.default_rng(1)
aggs = [AggFunc("count", InputRef(1), abi.INT64), AggFunc("sum", InputRef(1), abi.FLOAT64)]
for label, n, G, rep in (("random 4e7 rows 5e5 groups", 40_000_000, 500_000, 1), ("adjacent pairs 4e7 rows 5e5 groups", 40_000_000, 500_000, 2),
                         ("adjacent x4 4e7 rows 5e5 groups", 40_000_000, 500_000, 4), ("random 4e7 rows 1e6 groups", 40_000_000, 1_000_000, 1),
                         ("random 2e7 rows 5e5 groups", 20_000_000, 500_000, 1)):
    k = np.repeat(rng.integers(0, G, n // rep, dtype=np.int64), rep)
    v = rng.random(n)
    b = dev([k, v])
    timed(label, lambda: HashAggExecutor(be, aggs, [InputRef(0)], [b], out_mem=D).execute(), n, top_n=6)
    b.release()
