"""Inner join, unique build keys that cover 1 / D of their range (D = 2, 8, 16), 1e6 build rows, N probe rows (all hit):
direct-address table (SQLRS_DENSE_JOIN_SLOTS_PLAIN >= D) against the general routes.  python tools/join_density.py"""
import os as _os; _os.environ.setdefault("SQLRS_HOOKS", "1")  # the SQLRS_* tuning hooks are consulted only in a process that opts in (common.hpp: hook)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, pyarrow as pa
from opshapes_common import be, D, abi, dev, drain, timed
from sqlrs_amd.expr import InputRef, JoinCondition
from sqlrs_amd.executor import HashJoinExecutor
n = int(float(os.environ.get("N", 1e8))); nb = 1_000_000
rng = np.random.default_rng(3)
for dens in (2, 8, 16):
    dimk = rng.permutation(nb).astype(np.int64) * dens
    pk = rng.integers(0, nb, n, dtype=np.int64) * dens
    lb = dev([dimk, dimk * 3 + 1]); pr = dev([pk, rng.random(n)])
    sch = pa.schema([pa.field("l.0", pa.int64()), pa.field("l.1", pa.int64()), pa.field("r.0", pa.int64()), pa.field("r.1", pa.float64())])
    for slots in ("4", "16"):
        os.environ["SQLRS_DENSE_JOIN_SLOTS_PLAIN"] = slots
        for io in (True, False):
            timed(f"1/{dens} of the range, slots {slots}, {'pairs' if io else 'joined batch'}",
                  lambda: HashJoinExecutor(be, [lb], [pr], "inner", JoinCondition([(InputRef(0), InputRef(0))]), sch, 2, out_mem=D).execute(indices_only=io), n)
    lb.release(); pr.release()
