"""HashAgg fed with several device-resident batches (the reference pushes one batch per child
poll): 2e8 rows / 1e6 groups as 1, 4 and 16 batches.  Run on the GPU box."""
import ctypes as C
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch

import sqlrs_amd
from sqlrs_amd import abi, datagen
from sqlrs_amd.expr import AggFunc, InputRef

dev = torch.device("cuda", 0)
be = sqlrs_amd.hip(0)
n, G = 200_000_000, int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
key = datagen.fill_chunks(torch.empty(n, dtype=torch.int64, device=dev), lambda i: datagen.key_t(0xA1, i, G))
val = datagen.fill_chunks(torch.empty(n, dtype=torch.float64, device=dev), lambda i: datagen.val_t(0xF2, i))
torch.cuda.synchronize()
gb, _k = abi.pack_exprs([InputRef(0)])
keep = []
aggs = (abi.AggFunc * 2)(AggFunc("count", InputRef(1), abi.INT64).abi_struct(keep), AggFunc("sum", InputRef(1), abi.FLOAT64).abi_struct(keep))


def run(nb):
    step = n // nb
    batches = []
    for b in range(nb):
        k, v = key[b * step:(b + 1) * step], val[b * step:(b + 1) * step]
        cols = [abi.device_column(abi.INT64, k.numel(), k.data_ptr()), abi.device_column(abi.FLOAT64, v.numel(), v.data_ptr())]
        batches.append(abi.RawBatch(cols, k.numel(), keepalive=[k, v]))

    def once():
        a = C.c_void_p()
        be.check(be.fn("hash_agg_create")(be.ctx, 1, gb, 2, aggs, C.byref(a)))
        for b in batches:
            be.check(be.fn("hash_agg_push")(a, b.ptr))
        o = C.POINTER(abi.Batch)()
        be.check(be.fn("hash_agg_finish")(a, abi.MEM_DEVICE, C.byref(o)))
        g = o.contents.num_rows
        be.fn("batch_release")(o)
        be.fn("hash_agg_destroy")(a)
        return g
    once()
    t = C.c_void_p()
    be.check(be.fn("timer_create")(be.ctx, C.byref(t)))
    best = 1e30
    for _ in range(3):
        be.check(be.fn("timer_start")(t))
        g = once()
        be.check(be.fn("timer_stop")(t))
        ms = C.c_double()
        be.check(be.fn("timer_elapsed_ms")(t, C.byref(ms)))
        best = min(best, ms.value)
    be.profile(True)
    once()
    pr = be.profile_read()
    be.profile(False)
    top = ", ".join(f"{k} {v[0]:.2f}ms x{v[1]}" for k, v in sorted(pr.items(), key=lambda kv: -kv[1][0])[:8])
    print(f"batches {nb:3d}: {best:8.3f} ms  groups {g}  | {top}")


for nb in (1, 4, 16):
    run(nb)
