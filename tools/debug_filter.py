import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
import sqlrs_amd
from sqlrs_amd import abi, datagen
from sqlrs_amd.expr import InputRef, Constant
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import device_batch
dev = torch.device("cuda", 0)
be = sqlrs_amd.new_ctx(0)
for n in (1000, 100_000, 10_000_000, 100_000_000):
    v1 = datagen.fill_chunks(torch.empty(n, dtype=torch.int64, device=dev), lambda i: datagen._lsr(datagen.splitmix64_t(0xC2, i), 33))
    ref_np = (datagen.splitmix64_np(0xC2, np.arange(min(n, 1000), dtype=np.int64)) >> np.uint64(33)).astype(np.int64)
    print("n", n, "torch==numpy first 1000:", bool((v1[:1000].cpu().numpy() == ref_np).all()), "min", v1.min().item(), "max", v1.max().item())
    for k in (1 << 30, int((1 << 31) * 0.99), int((1 << 31) * 0.01), 5, (1 << 30) + 1, (1 << 30) - 1, 1 << 29):
        e = (InputRef(0) > Constant(k, abi.INT64)).pack()
        b = device_batch(abi, [v1], [abi.INT64])
        f = C.c_void_p(); be.check(be.fn("filter_create")(be.ctx, C.byref(e.abi), C.byref(f)))
        o = C.POINTER(abi.Batch)(); be.check(be.fn("filter_push")(f, b.ptr, abi.MEM_DEVICE, C.byref(o)))
        kept = o.contents.num_rows
        exp = int((v1 > k).sum().item())
        print(f"   k={k}: kept {kept} expected {exp} {'OK' if kept == exp else 'MISMATCH'}")
        be.fn("batch_release")(o); be.fn("filter_destroy")(f)
