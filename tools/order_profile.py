"""Per-kernel-class device time of ORDER BY v1 (int64, 31 bits) carrying one f64 column, N rows (bench's Order shape)."""
import os as _os; _os.environ.setdefault("SQLRS_HOOKS", "1")  # the SQLRS_* tuning hooks are consulted only in a process that opts in (common.hpp: hook)
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, sqlrs_amd
from sqlrs_amd import abi, datagen
from sqlrs_amd.expr import InputRef
dev = torch.device("cuda", 0)
be = sqlrs_amd.new_ctx(0)
n = int(float(os.environ.get("N", 1e8)))
v1 = datagen.fill_chunks(torch.empty(n, dtype=torch.int64, device=dev), lambda i: datagen._lsr(datagen.splitmix64_t(0xC2, i), 33))
val = datagen.fill_chunks(torch.empty(n, dtype=torch.float64, device=dev), lambda i: datagen.val_t(0xF2, i))
torch.cuda.synchronize()
bo = bench.device_batch(abi, [v1, val], [abi.INT64, abi.FLOAT64])
pk = InputRef(0).pack()
obs = (abi.OrderBy * 1)(abi.OrderBy(pk.abi, 1, 0))
def run_order():
    h = C.c_void_p()
    be.check(be.fn("order_create")(be.ctx, 1, obs, C.byref(h)))
    be.check(be.fn("order_push_retained")(h, bo.ptr))  # `bo` outlives the sort (order.rs:19-26 keeps Arcs)
    o = C.POINTER(abi.Batch)()
    be.check(be.fn("order_finish")(h, abi.MEM_DEVICE, C.byref(o)))
    be.fn("batch_release")(o)
    be.fn("order_destroy")(h)
VAR = os.environ.get("VAR", "SQLRS_ORDER_VARIANT")
for st in os.environ.get("VALUES", "0").split(","):
    os.environ[VAR] = st
    run_order(); be.synchronize()
    prof = os.environ.get("PROFILE", "1") != "0"
    be.profile(prof)
    t = time.perf_counter()
    for _ in range(5):
        run_order()
    be.synchronize()
    ms = (time.perf_counter() - t) / 5 * 1e3
    pr = be.profile_read() if prof else {}; be.profile(False)
    print(f"{VAR}={st}: {ms:.2f} ms | " + " ".join(f"{k} {v[0]/5:.3f}" for k, v in sorted(pr.items(), key=lambda kv: -kv[1][0])[:8]), flush=True)
