"""C4 (2e8 rows / 1e6 groups) in one process, the ctx pool trimmed between rounds (fresh hipMalloc blocks = fresh physical
placement of every intermediate buffer): does the claimed level's time follow the placement (it runs 1.43 ms on one box and
1.96 on another with the same binary)?  REGEN=1 also re-allocates the INPUT columns every round.
    python tools/c4_placement.py      (ROUNDS=8)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, sqlrs_amd
from sqlrs_amd import abi, datagen
from sqlrs_amd.expr import AggFunc, InputRef
dev = torch.device("cuda", 0)
be = sqlrs_amd.new_ctx(0)
n, G = int(float(os.environ.get("N", 2e8))), int(float(os.environ.get("G", 1e6)))
def gen():
    key = datagen.fill_chunks(torch.empty(n, dtype=torch.int64, device=dev), lambda i: datagen.key_t(0xA1, i, G))
    val = datagen.fill_chunks(torch.empty(n, dtype=torch.float64, device=dev), lambda i: datagen.val_t(0xF2, i))
    torch.cuda.synchronize()
    return key, val
key, val = gen()
gb, _k = abi.pack_exprs([InputRef(0)])
keep = []
aggs = (abi.AggFunc * 2)(AggFunc("count", InputRef(1), abi.INT64).abi_struct(keep), AggFunc("sum", InputRef(1), abi.FLOAT64).abi_struct(keep))
def run(b):
    a = C.c_void_p()
    be.check(be.fn("hash_agg_create")(be.ctx, 1, gb, 2, aggs, C.byref(a)))
    be.check(be.fn("hash_agg_push")(a, b.ptr))
    o = C.POINTER(abi.Batch)()
    be.check(be.fn("hash_agg_finish")(a, abi.MEM_DEVICE, C.byref(o)))
    be.fn("batch_release")(o); be.fn("hash_agg_destroy")(a)
t = C.c_void_p(); be.check(be.fn("timer_create")(be.ctx, C.byref(t)))
hold = []
for rnd in range(int(os.environ.get("ROUNDS", 8))):
    be.synchronize(); be.fn("ctx_pool_trim")(be.ctx)
    if os.environ.get("REGEN") == "1" and rnd:
        del key, val; torch.cuda.empty_cache()
        if os.environ.get("HOLD") == "1":
            hold.append(torch.empty((rnd * 53 + 17) << 20, dtype=torch.uint8, device=dev))
        key, val = gen()
    b = bench.device_batch(abi, [key, val], [abi.INT64, abi.FLOAT64])
    run(b); run(b)
    best = 1e9
    for _ in range(3):
        be.check(be.fn("timer_start")(t)); run(b); be.check(be.fn("timer_stop")(t))
        ms = C.c_double(); be.check(be.fn("timer_elapsed_ms")(t, C.byref(ms))); best = min(best, ms.value)
    be.profile(True); run(b); pr = be.profile_read(); be.profile(False)
    cls = ", ".join(f"{k} {x[0]:.3f}" for k, x in sorted(pr.items(), key=lambda kv: -kv[1][0])[:3])
    print(f"round {rnd}: C4 {best:.3f} ms | {cls} | key@{key.data_ptr():#x} val@{val.data_ptr():#x}", flush=True)
