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
