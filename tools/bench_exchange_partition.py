"""sqlrs_hash_partition on device-resident columns: 1e7 partial-aggregate rows (key, count, sum)
into 8 partitions, and 6.2e7 filtered fact rows (key, val).  Run on the GPU box."""
import ctypes as C
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch

import sqlrs_amd
from sqlrs_amd import abi, datagen
from sqlrs_amd.expr import InputRef

dev = torch.device("cuda", 0)
be = sqlrs_amd.hip(0)
for n, ncol in ((10_000_000, 3), (62_000_000, 2)):
    cols = [datagen.fill_chunks(torch.empty(n, dtype=torch.int64, device=dev), lambda i: datagen.key_t(0xA1 + c, i, 1 << 40)) for c in range(ncol)]
    torch.cuda.synchronize()
    b = abi.RawBatch([abi.device_column(abi.INT64, n, t.data_ptr()) for t in cols], n, keepalive=cols)
    for it in range(4):
        be.synchronize()
        t0 = time.perf_counter()
        parts, offs = be.hash_partition(b, InputRef(0), 8, abi.MEM_DEVICE)
        be.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        parts.release()
    be.profile(True)
    parts, offs = be.hash_partition(b, InputRef(0), 8, abi.MEM_DEVICE)
    be.synchronize()
    pr = be.profile_read()
    be.profile(False)
    parts.release()
    print(f"rows {n:>10,} x {ncol} cols -> 8 partitions: {dt:.3f} ms  |", ", ".join(f"{k} {v[0]:.2f}ms" for k, v in sorted(pr.items(), key=lambda kv: -kv[1][0])[:6]))
