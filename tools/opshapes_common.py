"""Shared helpers of the operator-shape sweep (tools/bench_opshapes.py)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, pyarrow as pa
import sqlrs_amd
from sqlrs_amd import abi

be = sqlrs_amd.new_ctx(0)
D = abi.MEM_DEVICE


def dev(arrays, names=None):
    names = names or [f"c{i}" for i in range(len(arrays))]
    return be.to_device(pa.RecordBatch.from_arrays([pa.array(a) if isinstance(a, np.ndarray) else a for a in arrays], names=names))


def drain(it):
    rows = 0
    for b in it:
        rows += b.num_rows
        b.release()
    return rows


def timed(label, make, rows_in, top_n=4):
    """one timing (after three warm-up runs: code objects, pool growth) + the top kernel classes of one more run"""
    drain(make()); drain(make()); drain(make()); be.synchronize()
    t = time.perf_counter(); out = drain(make()); be.synchronize(); ms = (time.perf_counter() - t) * 1e3
    be.profile(True); drain(make()); pr = be.profile_read(); be.profile(False)
    top = sorted(pr.items(), key=lambda kv: -kv[1][0])[:top_n]
    print(f"{label:60s} in {rows_in:.1e} out {out:10d} {ms:9.2f} ms {rows_in/ms/1e6:7.3f} Grows/s   " +
          ", ".join(f"{a} {v[0]:.2f}" for a, v in top), flush=True)
