"""C4 with Zipf(1.1) keys (and uniform) in one process with an A/B of a per-call hook (VAR / VALUES)."""
import os as _os; _os.environ.setdefault("SQLRS_HOOKS", "1")  # the SQLRS_* tuning hooks are consulted only in a process that opts in (common.hpp: hook)
import os, sys, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
import sqlrs_amd
from sqlrs_amd import abi, datagen
from sqlrs_amd.expr import AggFunc, InputRef
from bench import device_batch
dev = torch.device("cuda", 0); be = sqlrs_amd.new_ctx(0)
n, G = 200_000_000, 1_000_000
val = datagen.fill_chunks(torch.empty(n, dtype=torch.float64, device=dev), lambda i: datagen.val_t(0xF2, i))
w = np.arange(1, G + 1, dtype=np.float64) ** -1.1
cdf = torch.from_numpy(np.cumsum(w) / w.sum()).to(dev)
perm_a = datagen._coprime_multiplier(G)
kz = datagen.fill_chunks(torch.empty(n, dtype=torch.int64, device=dev), lambda i: (torch.searchsorted(cdf, datagen.val_t(0xA7, i)).clamp_(max=G - 1) * perm_a + 7) % G)
ku = datagen.fill_chunks(torch.empty(n, dtype=torch.int64, device=dev), lambda i: datagen.key_t(0xA1, i, G))
torch.cuda.synchronize()
gb, _k3 = abi.pack_exprs([InputRef(0)]); keep = []
aggs = (abi.AggFunc * 2)(AggFunc("count", InputRef(1), abi.INT64).abi_struct(keep), AggFunc("sum", InputRef(1), abi.FLOAT64).abi_struct(keep))
def run(b):
    a = C.c_void_p()
    be.check(be.fn("hash_agg_create")(be.ctx, 1, gb, 2, aggs, C.byref(a)))
    be.check(be.fn("hash_agg_push")(a, b.ptr))
    o = C.POINTER(abi.Batch)()
    be.check(be.fn("hash_agg_finish")(a, abi.MEM_DEVICE, C.byref(o)))
    be.fn("batch_release")(o); be.fn("hash_agg_destroy")(a)
VAR = os.environ.get("VAR", "SQLRS_DENSE_CHUNK_DIV")
for val_ in os.environ.get("VALUES", "2").split(","):
    os.environ[VAR] = val_
    for name, k in (("uniform", ku), ("zipf1.1", kz)):
        b = device_batch(abi, [k, val], [abi.INT64, abi.FLOAT64])
        run(b); run(b)
        be.profile(True); run(b); run(b); pr = be.profile_read(); be.profile(False)
        tot = sum(v[0] for v in pr.values()) / 2
        print(f"{VAR}={val_:>3} {name:8s}: kernels {tot:6.3f} ms | " + " ".join(f"{kk} {v[0]/2:.3f}" for kk, v in sorted(pr.items(), key=lambda kv: -kv[1][0])[:5]), flush=True)
