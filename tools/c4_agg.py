"""C4 alone (2e8 rows, 1e6 int64 groups, COUNT + SUM(double); hash_agg_create .. push .. finish .. destroy) in one process:
event-timed best of REPS per variant of an env hook that the library reads per call, e.g.
  VAR=SQLRS_RP_SLIM VALS=1,0 python tools/c4_agg.py        (slim / 16-byte rows out of the claimed level)
and what tools/timeline_ops.sh slices:  CMD="python tools/c4_agg.py" DELIM=key_stats_kernel bash tools/timeline_ops.sh
SHAPE=uniform|zipf|sorted, SPARSE=1 = the same groups under keys k * A + B, N / G = rows / groups, WHERE=1 adds `val > 0.5` handed to the aggregate."""
import os as _os; _os.environ.setdefault("SQLRS_HOOKS", "1")  # the SQLRS_* tuning hooks are consulted only in a process that opts in (common.hpp: hook)
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, sqlrs_amd
from sqlrs_amd import abi, datagen
from sqlrs_amd.expr import AggFunc, Constant, InputRef
dev = torch.device("cuda", 0)
be = abi.Backend(os.environ["LIB"], "sqlrs_", 0) if os.environ.get("LIB") else sqlrs_amd.new_ctx(0)
n, G = int(float(os.environ.get("N", 2e8))), int(float(os.environ.get("G", 1e6)))
shape = os.environ.get("SHAPE", "uniform")
key = datagen.fill_chunks(torch.empty(n, dtype=torch.int64, device=dev), lambda i: datagen.key_t(0xA1, i, G))
if shape == "sorted":
    key = torch.sort(key).values
elif shape == "zipf":
    u = torch.rand(n, device=dev, dtype=torch.float64)
    key = ((G ** u - 1).to(torch.int64).clamp_(0, G - 1) * 7919 + 13) % G  # log-uniform ranks: Zipf(1)-like weights
if os.environ.get("SPARSE") == "1":  # keys k -> k * A + B: a bijection of int64, the groups no longer fill a range
    key = key * (0x9E3779B97F4A7C15 - (1 << 64)) + 12345
val = datagen.fill_chunks(torch.empty(n, dtype=torch.float64, device=dev), lambda i: datagen.val_t(0xF2, i))
torch.cuda.synchronize()
b = bench.device_batch(abi, [key, val], [abi.INT64, abi.FLOAT64])
gb, _k = abi.pack_exprs([InputRef(0)])
keep = []
aggs = (abi.AggFunc * 2)(AggFunc("count", InputRef(1), abi.INT64).abi_struct(keep), AggFunc("sum", InputRef(1), abi.FLOAT64).abi_struct(keep))
pred = (InputRef(1) > Constant(0.5, abi.FLOAT64)).pack()
groups = [0]
def run():
    a = C.c_void_p()
    be.check(be.fn("hash_agg_create")(be.ctx, 1, gb, 2, aggs, C.byref(a)))
    if os.environ.get("WHERE") == "1":
        be.check(be.fn("hash_agg_set_filter")(a, C.byref(pred.abi)))
    be.check(be.fn("hash_agg_push")(a, b.ptr))
    o = C.POINTER(abi.Batch)()
    be.check(be.fn("hash_agg_finish")(a, abi.MEM_DEVICE, C.byref(o)))
    groups[0] = o.contents.num_rows
    be.fn("batch_release")(o)
    be.fn("hash_agg_destroy")(a)
var, vals = os.environ.get("VAR"), os.environ.get("VALS", "").split(",")
t = C.c_void_p(); be.check(be.fn("timer_create")(be.ctx, C.byref(t)))
for rnd in range(int(os.environ.get("ROUNDS", 2))):
    for v in (vals if var else [""]):
        if var:
            os.environ[var] = v
        run(); run()
        best = 1e9
        for _ in range(int(os.environ.get("REPS", 5))):
            be.check(be.fn("timer_start")(t)); run(); be.check(be.fn("timer_stop")(t))
            ms = C.c_double(); be.check(be.fn("timer_elapsed_ms")(t, C.byref(ms))); best = min(best, ms.value)
        be.profile(True); run(); pr = be.profile_read(); be.profile(False)
        cls = ", ".join(f"{k} {x[0]:.3f}" for k, x in sorted(pr.items(), key=lambda kv: -kv[1][0]) if x[0] > 0.01)
        print(f"C4 {os.path.basename(os.environ.get('LIB', 'default'))} {shape} {var}={v}: {best:.3f} ms  {groups[0]} groups  frac {(16 * n + 24 * groups[0]) / best / 1e6 / 8000:.4f} | {cls}", flush=True)
be.fn("timer_destroy")(t)
