"""C2 alone (1e8 int64 rows, v1 > k; filter_create .. push .. release .. destroy) REPS times in one process: event-timed best
and what tools/timeline_ops.sh slices (CMD="python tools/c2_filter.py" DELIM=filter_cmp_const).  SEL = selectivity."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, sqlrs_amd
from sqlrs_amd import abi, datagen
from sqlrs_amd.expr import Constant, InputRef
dev = torch.device("cuda", 0)
be = abi.Backend(os.environ["LIB"], "sqlrs_", 0) if os.environ.get("LIB") else sqlrs_amd.new_ctx(0)
n = int(float(os.environ.get("N", 1e8)))
sel = float(os.environ.get("SEL", 0.5))
v1 = datagen.fill_chunks(torch.empty(n, dtype=torch.int64, device=dev), lambda i: datagen._lsr(datagen.splitmix64_t(0xC2, i), 33))
torch.cuda.synchronize()
e = (InputRef(0) > Constant(int((1 << 31) * (1 - sel)), abi.INT64)).pack()
b = bench.device_batch(abi, [v1], [abi.INT64])
kept = [0]
def run():
    f = C.c_void_p()
    be.check(be.fn("filter_create")(be.ctx, C.byref(e.abi), C.byref(f)))
    o = C.POINTER(abi.Batch)()
    be.check(be.fn("filter_push")(f, b.ptr, abi.MEM_DEVICE, C.byref(o)))
    kept[0] = o.contents.num_rows
    be.fn("batch_release")(o)
    be.fn("filter_destroy")(f)
run(); run(); be.synchronize()
t = C.c_void_p(); be.check(be.fn("timer_create")(be.ctx, C.byref(t)))
best = 1e9
for _ in range(int(os.environ.get("REPS", 7))):
    be.check(be.fn("timer_start")(t)); run(); be.check(be.fn("timer_stop")(t))
    ms = C.c_double(); be.check(be.fn("timer_elapsed_ms")(t, C.byref(ms))); best = min(best, ms.value)
be.fn("timer_destroy")(t)
print(f"C2 {os.path.basename(os.environ.get('LIB', 'default'))} sel {sel}: {best:.3f} ms  kept {kept[0]}  {(8 * n + 8 * kept[0]) / best / 1e6 / 8000:.4f} of 8 TB/s", flush=True)
