"""HashAgg under key skew (Zipf): correctness vs torch + time (SURVEY.md §8d asks for the variant)."""
import sys, os, time, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, numpy as np
import sqlrs_amd
from sqlrs_amd import abi
from sqlrs_amd.expr import AggFunc, InputRef
from bench import device_batch
dev = torch.device("cuda", 0); be = sqlrs_amd.new_ctx(0)
n, G = 100_000_000, 1_000_000
for name, gen in (("uniform", lambda: torch.randint(0, G, (n,), device=dev, dtype=torch.int64)),
                  ("zipf1.1", lambda: torch.from_numpy((np.random.default_rng(1).zipf(1.1, n) % G).astype(np.int64)).to(dev)),
                  ("one-hot 50%", lambda: torch.where(torch.rand(n, device=dev) < 0.5, torch.zeros(n, device=dev, dtype=torch.int64), torch.randint(0, G, (n,), device=dev, dtype=torch.int64)))):
    key = gen(); val = torch.rand(n, device=dev, dtype=torch.float64); torch.cuda.synchronize()
    b = device_batch(abi, [key, val], [abi.INT64, abi.FLOAT64])
    gb, _k = abi.pack_exprs([InputRef(0)]); keep = []
    aggs = (abi.AggFunc * 2)(AggFunc("count", InputRef(1), abi.INT64).abi_struct(keep), AggFunc("sum", InputRef(1), abi.FLOAT64).abi_struct(keep))
    def run():
        a = C.c_void_p(); be.check(be.fn("hash_agg_create")(be.ctx, 1, gb, 2, aggs, C.byref(a)))
        be.check(be.fn("hash_agg_push")(a, b.ptr)); o = C.POINTER(abi.Batch)()
        be.check(be.fn("hash_agg_finish")(a, abi.MEM_DEVICE, C.byref(o))); be.fn("hash_agg_destroy")(a); return be.wrap(o)
    out = run(); be.synchronize()
    t = time.perf_counter(); out2 = run(); be.synchronize(); dt = time.perf_counter() - t
    g = out.num_rows
    from bench import _tensor_view
    k = _tensor_view(torch, out.column(0).values, g, torch.int64, dev); c = _tensor_view(torch, out.column(1).values, g, torch.int64, dev)
    s = _tensor_view(torch, out.column(2).values, g, torch.float64, dev)
    uk, cnt = torch.unique(key, return_counts=True)
    ok = (g == uk.numel()) and bool((torch.sort(k).values == uk).all()) and int(c.sum()) == n and abs(float(s.sum()) - float(val.sum())) < 1e-6 * n
    order = torch.argsort(k); ok = ok and bool((c[order] == cnt).all())
    be.profile(True); run().release(); pr = be.profile_read(); be.profile(False)
    top = sorted(pr.items(), key=lambda kv: -kv[1][0])[:4]
    print(f"{name:12s} groups {g:8d} max group {int(cnt.max()):9d} ok={ok} time {dt*1e3:7.2f} ms  top: " + ", ".join(f"{a} {v[0]:.2f}ms" for a, v in top))
