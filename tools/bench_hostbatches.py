"""Drop-in realism: the operators fed 1024-row HOST batches (the reference's CSV batch size,
storage/csv.rs:105) through the C ABI, PCIe inclusive.  Batches are marshalled once (the binding's
cost is not the library's); the timed region is the push loop + finish + the result on the host.
N = fact rows (default 2e7), B = rows per batch (default 1024)."""
import ctypes as C
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import pyarrow as pa

import sqlrs_amd
from sqlrs_amd import abi, datagen
from sqlrs_amd.expr import AggFunc, Constant, InputRef, JoinCondition, OrderBy

be = sqlrs_amd.new_ctx(0)
n, B = int(float(os.environ.get("N", 2e7))), int(os.environ.get("B", 1024))
nd = 1_000_000
idx = np.arange(n, dtype=np.int64)
fact = pa.RecordBatch.from_arrays([pa.array(datagen.key_np(0xF1, idx, nd)), pa.array(datagen.val_np(0xF2, idx))], names=["key", "val"])
dim = pa.RecordBatch.from_arrays([pa.array(datagen.dim_key_np(np.arange(nd, dtype=np.int64), nd))], names=["key"])
fb = [abi.HostBatch(fact.slice(lo, B)) for lo in range(0, n, B)]
db = [abi.HostBatch(dim.slice(lo, B)) for lo in range(0, nd, B)]
print(f"{n:,} fact rows as {len(fb):,} host batches of {B} rows, {nd:,} dim rows as {len(db):,} batches", flush=True)
H = abi.MEM_HOST


def report(label, rows, fn):
    fn()  # warm-up: code objects, memory pool
    t = time.perf_counter()
    out_rows = fn()
    dt = time.perf_counter() - t
    print(f"{label:66s} {dt * 1e3:9.1f} ms  {rows / dt / 1e6:9.1f} Mrows/s  ({out_rows:,} result rows)", flush=True)


def hash_agg():
    gb, _k = abi.pack_exprs([InputRef(0)])
    keep = []
    aggs = (abi.AggFunc * 2)(AggFunc("count", InputRef(1), abi.INT64).abi_struct(keep), AggFunc("sum", InputRef(1), abi.FLOAT64).abi_struct(keep))
    a = C.c_void_p()
    be.check(be.fn("hash_agg_create")(be.ctx, 1, gb, 2, aggs, C.byref(a)))
    push = be.fn("hash_agg_push")
    for b in fb:
        be.check(push(a, b.ptr))
    o = C.POINTER(abi.Batch)()
    be.check(be.fn("hash_agg_finish")(a, H, C.byref(o)))
    r = o.contents.num_rows
    be.fn("batch_release")(o)
    be.fn("hash_agg_destroy")(a)
    return r


def join_agg(with_filter):
    lk, _1 = abi.pack_exprs([InputRef(0)])
    rk, _2 = abi.pack_exprs([InputRef(0)])
    gb, _3 = abi.pack_exprs([InputRef(0)])
    keep = []
    aggs = (abi.AggFunc * 2)(AggFunc("count", InputRef(2), abi.INT64).abi_struct(keep), AggFunc("sum", InputRef(2), abi.FLOAT64).abi_struct(keep))
    rd = (C.c_int32 * 2)(abi.INT64, abi.FLOAT64)
    ja = C.c_void_p()
    be.check(be.fn("join_agg_create")(be.ctx, 1, lk, rk, 1, 2, rd, 1, gb, 2, aggs, C.byref(ja)))
    if with_filter:
        pf = (InputRef(1) > Constant(0.5, abi.FLOAT64)).pack()
        be.check(be.fn("join_agg_set_probe_filter")(ja, C.byref(pf.abi)))
    for b in db:
        be.check(be.fn("join_agg_build_push")(ja, b.ptr))
    be.check(be.fn("join_agg_build_finish")(ja))
    push = be.fn("join_agg_probe_push")
    for b in fb:
        be.check(push(ja, b.ptr))
    o = C.POINTER(abi.Batch)()
    be.check(be.fn("join_agg_finish")(ja, H, C.byref(o)))
    r = o.contents.num_rows
    be.fn("batch_release")(o)
    be.fn("join_agg_destroy")(ja)
    return r


def order():
    pk = InputRef(0).pack()
    obs = (abi.OrderBy * 1)(abi.OrderBy(pk.abi, 1, 0))
    h = C.c_void_p()
    be.check(be.fn("order_create")(be.ctx, 1, obs, C.byref(h)))
    push = be.fn("order_push")
    for b in fb:
        be.check(push(h, b.ptr))
    o = C.POINTER(abi.Batch)()
    be.check(be.fn("order_finish")(h, H, C.byref(o)))
    r = o.contents.num_rows
    be.fn("batch_release")(o)
    be.fn("order_destroy")(h)
    return r


def filter_stream(limit_batches):
    e = (InputRef(1) > Constant(0.5, abi.FLOAT64)).pack()
    f = C.c_void_p()
    be.check(be.fn("filter_create")(be.ctx, C.byref(e.abi), C.byref(f)))
    push, rel = be.fn("filter_push"), be.fn("batch_release")
    r = 0
    for b in fb[:limit_batches]:
        o = C.POINTER(abi.Batch)()
        be.check(push(f, b.ptr, H, C.byref(o)))
        r += o.contents.num_rows
        rel(o)
    be.fn("filter_destroy")(f)
    return r


report("HashAgg  GROUP BY key COUNT,SUM  (blocking: pushes are staged on the host)", n, hash_agg)
report("HashJoinAgg dim x fact, group by key", n, lambda: join_agg(False))
report("HashJoinAgg with the probe-side Filter val > 0.5", n, lambda: join_agg(True))
report("Order BY key (result downloaded)", n, order)
nb = min(len(fb), 2000)
report(f"Filter val > 0.5, one output batch per input batch ({nb} batches)", nb * B, lambda: filter_stream(nb))
