"""Fourth pathology sweep: ORDER BY shapes (long / many distinct Utf8 keys, many payload columns, many input batches)."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, pyarrow as pa
import sqlrs_amd
from sqlrs_amd import abi
from sqlrs_amd.expr import InputRef, OrderBy
from sqlrs_amd.executor import OrderExecutor
be = sqlrs_amd.new_ctx(0)
D = abi.MEM_DEVICE
n = int(float(os.environ.get("N", 2e6)))
rng = np.random.default_rng(3)
def dev(arrays):
    return be.to_device(pa.RecordBatch.from_arrays([pa.array(a) if isinstance(a, np.ndarray) else a for a in arrays], names=[f"c{i}" for i in range(len(arrays))]))
def drain(it):
    rows = 0
    for b in it:
        rows += b.num_rows
        b.release()
    return rows
def timed(label, make, rows_in):
    drain(make()); drain(make()); drain(make()); be.synchronize()
    t = time.perf_counter(); out = drain(make()); be.synchronize(); ms = (time.perf_counter() - t) * 1e3
    be.profile(True); drain(make()); pr = be.profile_read(); be.profile(False)
    top = sorted(pr.items(), key=lambda kv: -kv[1][0])[:4]
    print(f"{label:56s} in {rows_in:.1e} out {out:9d} {ms:9.2f} ms {rows_in/ms/1e6:7.3f} Grows/s   " + ", ".join(f"{a} {v[0]:.2f}" for a, v in top), flush=True)
words = np.array(["".join(chr(97 + int(c)) for c in rng.integers(0, 26, int(rng.integers(3, 24)))) for _ in range(200_000)])
s1 = pa.array(words[rng.integers(0, len(words), n)].tolist())
s2 = pa.array([f"customer#{x:09d}" for x in rng.integers(0, 10**9, n)])
ob = dev([s1, s2, rng.integers(0, 1 << 30, n, dtype=np.int64), rng.random(n), rng.random(n), rng.integers(0, 9, n, dtype=np.int64), rng.random(n)])
timed("order by utf8, 2e5 distinct words of 3-23 bytes", lambda: OrderExecutor(be, [OrderBy(InputRef(0), True)], [ob], out_mem=D).execute(), n)
timed("order by utf8 'customer#%09d' (18 bytes, all distinct)", lambda: OrderExecutor(be, [OrderBy(InputRef(1), True)], [ob], out_mem=D).execute(), n)
timed("order by int64, 6 payload columns (2 utf8)", lambda: OrderExecutor(be, [OrderBy(InputRef(2), True)], [ob], out_mem=D).execute(), n)
ob.release()
parts = [dev([rng.integers(0, 1 << 30, n // 64, dtype=np.int64), rng.random(n // 64)]) for _ in range(64)]
timed("order by int64, input pushed as 64 batches", lambda: OrderExecutor(be, [OrderBy(InputRef(0), True)], parts, out_mem=D).execute(), n // 64 * 64)
small = dev([rng.integers(0, 100, 1000, dtype=np.int64), rng.random(1000)])
timed("order by int64, 1000 rows", lambda: OrderExecutor(be, [OrderBy(InputRef(0), False)], [small], out_mem=D).execute(), 1000)
