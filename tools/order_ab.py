"""ORDER BY on 1e8 rows of a wide key (f64 uniform / 63-bit int64), one carried column: time per call under the values of one
environment hook that order_fast.hip reads per call.  VAR=SQLRS_ORDER_SAMPLES VALUES=16,32,64 python tools/order_ab.py"""
import os as _os; _os.environ.setdefault("SQLRS_HOOKS", "1")  # the SQLRS_* tuning hooks are consulted only in a process that opts in (common.hpp: hook)
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sqlrs_amd  # noqa: E402
from sqlrs_amd import abi  # noqa: E402
from sqlrs_amd.expr import InputRef  # noqa: E402

n = int(float(os.environ.get("ROWS", 1e8)))
var = os.environ.get("VAR", "SQLRS_ORDER_SAMPLES")
values = os.environ.get("VALUES", "16,32").split(",")
reps = int(os.environ.get("REPS", 5))
be = sqlrs_amd.hip(0)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
keys = {"f64_unit": torch.rand(n, dtype=torch.float64, device=dev, generator=g),
        "i64_63bit": torch.randint(-(1 << 62), 1 << 62, (n,), dtype=torch.int64, device=dev, generator=g),
        "f64_normal": torch.randn(n, dtype=torch.float64, device=dev, generator=g)}
keys["i64_31bit"] = torch.randint(0, 1 << 31, (n,), dtype=torch.int64, device=dev, generator=g)   # the bench's Order shape
z = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
z[torch.rand(n, device=dev, generator=g) < 0.3] = 0.0           # a heavy value: 30 % zeros
keys["f64_zeros30"] = z
keys["i64_50_values"] = torch.randint(-(1 << 62), 1 << 62, (50,), dtype=torch.int64, device=dev, generator=g)[
    torch.randint(0, 50, (n,), device=dev, generator=g)]
if os.environ.get("SHAPES"):
    keys = {k: v for k, v in keys.items() if k in os.environ["SHAPES"].split(",")}
carry = torch.arange(n, dtype=torch.int64, device=dev)
D = abi.MEM_DEVICE
pk = InputRef(0).pack()
obs = (abi.OrderBy * 1)(abi.OrderBy(pk.abi, 1, 0))


valid_bits = None
if "f64_nulls10" in os.environ.get("SHAPES", ""):   # (only on request: a key column with 10 % NULLs)
    keys["f64_nulls10"] = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
    v8 = (torch.rand(n + (-n) % 64, device=dev, generator=g) >= 0.1).view(-1, 8).to(torch.int32)
    valid_bits = (v8 << torch.arange(8, device=dev, dtype=torch.int32)).sum(1).to(torch.uint8)


def batch_of(k, name=""):
    kt = abi.FLOAT64 if k.dtype == torch.float64 else abi.INT64
    if name == "f64_nulls10":
        c0 = abi.device_column(kt, n, k.data_ptr(), validity_ptr=valid_bits.data_ptr(), null_count=-1)
    else:
        c0 = abi.device_column(kt, n, k.data_ptr())
    return abi.RawBatch([c0, abi.device_column(abi.INT64, n, carry.data_ptr())], n, keepalive=[k, carry])


def run(b):
    h = C.c_void_p()
    be.check(be.fn("order_create")(be.ctx, 1, obs, C.byref(h)))
    be.check(be.fn("order_push_retained")(h, b.ptr))
    o = C.POINTER(abi.Batch)()
    be.check(be.fn("order_finish")(h, D, C.byref(o)))
    be.fn("batch_release")(o)
    be.fn("order_destroy")(h)


for name, k in keys.items():
    b = batch_of(k, name)
    for rnd in range(2):
        for v in values:
            os.environ[var] = v
            run(b)
            best = 1e9
            t = C.c_void_p()
            be.check(be.fn("timer_create")(be.ctx, C.byref(t)))  # (events on the library's own stream)
            for _ in range(reps):
                be.check(be.fn("timer_start")(t))
                run(b)
                be.check(be.fn("timer_stop")(t))
                ms = C.c_double()
                be.check(be.fn("timer_elapsed_ms")(t, C.byref(ms)))
                best = min(best, ms.value)
            be.fn("timer_destroy")(t)
            be.profile(True)
            run(b)
            pr = be.profile_read()
            be.profile(False)
            top = ", ".join(f"{kk} {vv[0]:.2f}" for kk, vv in sorted(pr.items(), key=lambda kv: -kv[1][0])[:5])
            print(f"{name:11s} {var}={v:>4s}: {best:6.3f} ms | {top}", flush=True)
    os.environ.pop(var, None)
