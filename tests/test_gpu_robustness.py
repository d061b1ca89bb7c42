"""Robustness of the C ABI around the kernels: look-back launches stay on their first (fast) launch on
an idle GPU and fall back cheaply when they cannot, batches may outlive their ctx, device buffers are
handed over stream-ordered."""
import ctypes as C
import time

import numpy as np
import pyarrow as pa
import pytest

import sqlrs_amd
from sqlrs_amd import abi
from sqlrs_amd.executor import FilterExecutor, HashJoinExecutor
from sqlrs_amd.expr import Constant, InputRef, JoinCondition

pytestmark = pytest.mark.gpu


def reruns(be):
    return be.profile_read().get("lookback_ticket_reruns", (0, 0))[1]


def test_lookback_kernels_stay_on_their_first_launch():
    """every single-pass (look-back) kernel shape on a large batch: no ticketed rerun is counted, i.e. no
    launch sat out the spin limit (a mis-sized persistent grid did exactly that once: 1.4 s per call)"""
    be = sqlrs_amd.new_ctx(0)
    try:
        rng = np.random.default_rng(0)
        n = 6_000_000
        cols = {"i64": pa.array(rng.integers(-100, 100, n)), "i32": pa.array(rng.integers(-100, 100, n).astype(np.int32)),
                "f64": pa.array(rng.random(n)), "i64n": pa.array(rng.integers(-100, 100, n), mask=rng.random(n) < 0.1)}
        consts = {"i64": Constant(3, abi.INT64), "i32": Constant(3, abi.INT32), "f64": Constant(0.5, abi.FLOAT64), "i64n": Constant(3, abi.INT64)}
        t = time.perf_counter()
        for name, arr in cols.items():
            b = be.to_device(pa.RecordBatch.from_arrays([arr, cols["f64"], cols["i32"]], names=["a", "b", "c"]))
            for _ in range(3):
                (o,) = list(FilterExecutor(be, InputRef(0) > consts[name], [b], out_mem=abi.MEM_DEVICE).execute())
                o.release()
        # dense (direct-address) and hashed unique-key probes, with and without NULL probe keys
        for sparse in (False, True):
            bk = rng.permutation(1_000_000).astype(np.int64) * (7919 if sparse else 1)
            lb = pa.RecordBatch.from_arrays([pa.array(bk)], names=["k"])
            for nulls in (0.0, 0.1):
                pk = rng.integers(0, 1_000_000, n, dtype=np.int64) * (7919 if sparse else 1)
                rb = be.to_device(pa.RecordBatch.from_arrays([pa.array(pk, mask=rng.random(n) < nulls if nulls else None)], names=["k"]))
                sch = pa.schema([("l.k", pa.int64()), ("r.k", pa.int64())])
                for jt in ("inner", "left"):
                    for o in HashJoinExecutor(be, [lb], [rb], jt, JoinCondition([(InputRef(0), InputRef(0))]), sch, 1,
                                              out_mem=abi.MEM_DEVICE).execute(indices_only=(jt == "inner")):
                        o.release()
        be.synchronize()
        assert reruns(be) == 0
        assert time.perf_counter() - t < 60
    finally:
        be.close()


def test_lookback_fallback_is_cheap_and_counted():
    """two ctxs hammering the same GPU from two threads: residency of the persistent grids is no longer
    guaranteed; results stay right, and whatever times out is counted and costs milliseconds"""
    import threading
    rng = np.random.default_rng(1)
    n = 4_000_000
    arr = rng.integers(-100, 100, n)
    exp = int((arr > 3).sum())
    out = {}

    def work(tag):
        be = sqlrs_amd.new_ctx(0)
        try:
            b = be.to_device(pa.RecordBatch.from_arrays([pa.array(arr)], names=["a"]))
            t = time.perf_counter()
            bad = 0
            for _ in range(40):
                (o,) = list(FilterExecutor(be, InputRef(0) > Constant(3, abi.INT64), [b], out_mem=abi.MEM_DEVICE).execute())
                bad += o.num_rows != exp
                o.release()
            be.synchronize()
            out[tag] = (bad, reruns(be), time.perf_counter() - t)
        finally:
            be.close()
    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for bad, n_reruns, secs in out.values():
        assert bad == 0
        assert secs < 20, (n_reruns, secs)  # 40 calls: even if every one timed out this is ~10 ms each


def test_batches_may_outlive_their_ctx():
    be = sqlrs_amd.new_ctx(0)
    b = pa.RecordBatch.from_arrays([pa.array(np.arange(100_000)), pa.array(np.arange(100_000) * 0.5)], names=["a", "b"])
    dev = be.to_device(b)
    (kept,) = list(FilterExecutor(be, InputRef(0) > Constant(10, abi.INT64), [dev], out_mem=abi.MEM_DEVICE).execute())
    host = be.to_host(kept)
    be.close()          # sqlrs_ctx_destroy while three batches are alive
    assert host.to_arrow(["a", "b"]).num_rows == 99_989
    for x in (kept, dev, host):
        x.release()     # device blocks go straight back to the driver: no use of the freed ctx
    be2 = sqlrs_amd.new_ctx(0)  # and the device is still usable
    assert be2.to_host(be2.to_device(b)).to_arrow(["a", "b"]).equals(b)
    be2.close()


def test_stream_ordered_handover_with_a_torch_stream():
    """inputs produced on another stream: sqlrs_ctx_wait_stream instead of a host synchronisation; the
    result is consumed on that stream after sqlrs_ctx_release_to_stream"""
    import torch
    be = sqlrs_amd.new_ctx(0)
    try:
        dev = torch.device("cuda", 0)
        side = torch.cuda.Stream(device=dev)
        n = 50_000_000
        with torch.cuda.stream(side):
            v = torch.arange(n, dtype=torch.int64, device=dev)
            for _ in range(20):  # keep the side stream busy so that an unordered read would see old data
                v = (v * 3 + 1) % 1000
            v = v.contiguous()
        be.check(be.fn("ctx_wait_stream")(be.ctx, C.c_void_p(side.cuda_stream)))
        batch = abi.RawBatch([abi.device_column(abi.INT64, n, v.data_ptr())], n, keepalive=[v])
        (o,) = list(FilterExecutor(be, InputRef(0) > Constant(499, abi.INT64), [batch], out_mem=abi.MEM_DEVICE).execute())
        be.check(be.fn("ctx_release_to_stream")(be.ctx, C.c_void_p(side.cuda_stream)))
        with torch.cuda.stream(side):
            expected = int((v > 499).sum().item())
        assert o.num_rows == expected
        o.release()
    finally:
        be.close()


def _claimed_batch(be, num_rows, ncols=2):
    """a caller-built DEVICE batch that CLAIMS `num_rows` rows over 64-byte buffers: an operator that reads a column
    before it has checked the claim faults instead of returning an error"""
    import torch
    keep = [torch.zeros(8, dtype=torch.int64, device="cuda:0") for _ in range(ncols)]
    cols = (abi.Column * ncols)()
    for i, t in enumerate(keep):
        cols[i].dtype, cols[i].mem, cols[i].length, cols[i].null_count = abi.INT64, abi.MEM_DEVICE, num_rows, 0
        cols[i].values = t.data_ptr()
    b = abi.Batch()
    b.num_rows, b.num_columns, b.columns = num_rows, ncols, cols
    return b, (keep, cols)


@pytest.mark.parametrize("rows", [1 << 31, (1 << 31) + 12345, -1, -(1 << 40)])
def test_every_operator_refuses_a_batch_outside_the_row_limit(hip, rows):
    """Review r05 #7: one guard at the ABI entrance (InBatch) instead of piecemeal checks inside the routes — a batch that
    claims < 0 or >= 2^31 rows is SQLRS_ERR_ARROW from every call that takes a batch, and the operator stays usable."""
    from sqlrs_amd.expr import AggFunc
    be = hip
    b, keep = _claimed_batch(be, rows)
    ok, keep_ok = _claimed_batch(be, 8)
    e0, _k0 = abi.pack_exprs([InputRef(0)])
    pred = (InputRef(0) > Constant(0, abi.INT64)).pack()
    rd = (C.c_int32 * 2)(abi.INT64, abi.INT64)
    kk = []
    aggs = (abi.AggFunc * 1)(AggFunc("count", InputRef(1), abi.INT64).abi_struct(kk))
    out = C.POINTER(abi.Batch)()

    def refused(st):
        assert st == abi.ERR_ARROW, st
        assert b"2^31" in (be.fn("last_error")(be.ctx) or b"")

    h = C.c_void_p()
    be.check(be.fn("filter_create")(be.ctx, C.byref(pred.abi), C.byref(h)))
    refused(be.fn("filter_push")(h, C.byref(b), abi.MEM_DEVICE, C.byref(out)))
    be.check(be.fn("filter_push")(h, C.byref(ok), abi.MEM_DEVICE, C.byref(out)))
    be.fn("batch_release")(out)
    be.fn("filter_destroy")(h)
    be.check(be.fn("project_create")(be.ctx, 1, e0, C.byref(h)))
    refused(be.fn("project_push")(h, C.byref(b), abi.MEM_DEVICE, C.byref(out)))
    be.fn("project_destroy")(h)
    be.check(be.fn("hash_join_create")(be.ctx, abi.JOIN_INNER, 1, e0, e0, None, 2, rd, C.byref(h)))
    refused(be.fn("hash_join_build_push")(h, C.byref(b)))
    be.check(be.fn("hash_join_build_push")(h, C.byref(ok)))
    be.check(be.fn("hash_join_build_finish")(h))
    refused(be.fn("hash_join_probe_push")(h, C.byref(b), abi.MEM_DEVICE, C.byref(out)))
    refused(be.fn("hash_join_probe_indices")(h, C.byref(b), abi.MEM_DEVICE, C.byref(out)))
    be.check(be.fn("hash_join_probe_push")(h, C.byref(ok), abi.MEM_DEVICE, C.byref(out)))
    be.fn("batch_release")(out)
    be.fn("hash_join_destroy")(h)
    be.check(be.fn("hash_agg_create")(be.ctx, 1, e0, 1, aggs, C.byref(h)))
    refused(be.fn("hash_agg_push")(h, C.byref(b)))
    be.check(be.fn("hash_agg_push")(h, C.byref(ok)))
    be.check(be.fn("hash_agg_finish")(h, abi.MEM_DEVICE, C.byref(out)))
    assert out.contents.num_rows == 1
    be.fn("batch_release")(out)
    be.fn("hash_agg_destroy")(h)
    be.check(be.fn("join_agg_create")(be.ctx, 1, e0, e0, 2, 2, rd, 1, e0, 1, aggs, C.byref(h)))
    refused(be.fn("join_agg_build_push")(h, C.byref(b)))
    be.check(be.fn("join_agg_build_push")(h, C.byref(ok)))
    be.check(be.fn("join_agg_build_finish")(h))
    refused(be.fn("join_agg_probe_push")(h, C.byref(b)))
    be.fn("join_agg_destroy")(h)
    keys = (abi.OrderBy * 1)()
    keys[0].expr, keys[0].asc = e0[0], 1
    be.check(be.fn("order_create")(be.ctx, 1, keys, C.byref(h)))
    refused(be.fn("order_push")(h, C.byref(b)))
    be.fn("order_destroy")(h)
    be.check(be.fn("limit_create")(be.ctx, 1, 5, 0, 0, C.byref(h)))
    done = C.c_int(0)
    refused(be.fn("limit_push")(h, C.byref(b), abi.MEM_DEVICE, C.byref(out), C.byref(done)))
    be.fn("limit_destroy")(h)
    be.check(be.fn("simple_agg_create")(be.ctx, 1, aggs, C.byref(h)))
    refused(be.fn("simple_agg_push")(h, C.byref(b)))
    be.fn("simple_agg_destroy")(h)
    refused(be.fn("eval_expr")(be.ctx, C.byref(pred.abi), C.byref(b), abi.MEM_DEVICE, C.byref(out)))
    offs = (C.c_int64 * 3)()
    refused(be.fn("hash_partition")(be.ctx, C.byref(b), e0, 2, abi.MEM_DEVICE, C.byref(out), offs))
    be.synchronize()
