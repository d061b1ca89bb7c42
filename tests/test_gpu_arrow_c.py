"""The Arrow C Data Interface at the boundary (sqlrs_batch_import_arrow / sqlrs_batch_export_arrow, review r05 #7): what
arrow-rs's arrow::ffi (and pyarrow's _export_to_c / _import_from_c) binds without glue.  [ref: src/executor/mod.rs:34 — the
item of every operator stream is an arrow RecordBatch]"""
import ctypes as C

import numpy as np
import pyarrow as pa
import pytest

from golden_runner import Runner, load, render_rows
from sqlrs_amd import abi
from sqlrs_amd.executor import FilterExecutor
from sqlrs_amd.expr import Constant, InputRef

pytestmark = pytest.mark.gpu
FX = load()
GPU_CASES = [c for c in FX["cases"] if c["gpu"]]


@pytest.fixture
def arrow_c(hip):
    abi.ARROW_C_BACKEND = hip
    try:
        yield hip
    finally:
        abi.ARROW_C_BACKEND = None


def sample(n, seed=0):
    rng = np.random.default_rng(seed)
    return pa.RecordBatch.from_arrays(
        [pa.array(rng.integers(-50, 50, n), mask=rng.random(n) < 0.2),
         pa.array(rng.integers(-50, 50, n).astype(np.int32)),
         pa.array(rng.random(n), mask=rng.random(n) < 0.1),
         pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.3),
         pa.array([None if rng.random() < 0.15 else "s" * int(rng.integers(0, 5)) + str(i) for i in range(n)])],
        names=["a", "b", "c", "d", "e"])


@pytest.mark.parametrize("lo,hi", [(0, 1000), (0, 0), (3, 1000), (8, 520), (13, 14), (64, 777)])
def test_import_export_round_trip(arrow_c, lo, hi):
    """every column type, NULLs, and SLICED batches (child offsets that are / are not whole bytes): import moves the exported
    structures (no copy of the value buffers), export moves the batch back out; pyarrow sees the same table"""
    rb = sample(1000).slice(lo, hi - lo)
    b = abi.ArrowCBatch(arrow_c, rb)
    assert b.ptr.contents.num_rows == rb.num_rows and b.ptr.contents.num_columns == 5
    if rb.num_rows and lo % 8 == 0:  # zero copy: the int32 column's values pointer lies inside pyarrow's buffer
        buf = rb.column(1).buffers()[1]
        assert buf.address <= b.ptr.contents.columns[1].values < buf.address + buf.size
    back = arrow_c.wrap(b.p)
    b.p = None  # (moved into the LibBatch wrapper)
    got = back.to_arrow(rb.schema.names)
    assert got.schema.names == rb.schema.names
    for i in range(rb.num_columns):
        assert got.column(i).to_pylist() == rb.column(i).to_pylist(), rb.schema.names[i]


def test_operator_over_imported_batches_and_device_export(arrow_c, oracle):
    """an operator fed imported batches gives the oracle's rows; a DEVICE-resident result is downloaded by the export"""
    rb = sample(5000, 1)
    pred = InputRef(1) > Constant(0, abi.INT32)
    got = list(FilterExecutor(arrow_c, pred, [rb, rb.slice(17, 900)]).execute())
    abi.ARROW_C_BACKEND = None
    exp = list(FilterExecutor(oracle, pred, [rb, rb.slice(17, 900)]).execute())
    assert [g.to_pylist() for g in got] == [e.to_pylist() for e in exp]
    abi.ARROW_C_BACKEND = arrow_c
    dev = arrow_c.to_device(rb)
    out = dev.to_arrow(rb.schema.names)  # export of a device batch
    assert out.to_pylist() == rb.to_pylist()


def test_import_refuses_what_it_cannot_represent(arrow_c):
    """unsupported child types and a plain (non-struct) array are SQLRS_ERR_ARROW, and NOTHING is consumed: the exported
    structures are still live and pyarrow takes them back"""
    rb = pa.RecordBatch.from_arrays([pa.array([1, 2, 3], type=pa.int16())], names=["x"])
    with pytest.raises(abi.ExecutorError) as e:
        abi.ArrowCBatch(arrow_c, rb)
    assert e.value.status == abi.ERR_ARROW and "unsupported column type" in str(e.value)
    arr, sch = abi.ArrowArrayC(), abi.ArrowSchemaC()
    pa.array([1, 2, 3])._export_to_c(C.addressof(arr), C.addressof(sch))
    out = C.POINTER(abi.Batch)()
    assert arrow_c.fn("batch_import_arrow")(arrow_c.ctx, C.byref(arr), C.byref(sch), C.byref(out)) == abi.ERR_ARROW
    assert arr.release and sch.release and not out
    assert pa.Array._import_from_c(C.addressof(arr), C.addressof(sch)).to_pylist() == [1, 2, 3]


@pytest.mark.parametrize("case", GPU_CASES, ids=[c["name"] for c in GPU_CASES])
def test_reference_goldens_through_the_arrow_c_interface(arrow_c, case):
    """the 80 golden tables of the reference's own tests with every input batch imported and every output batch exported
    through the Arrow C Data Interface"""
    assert Runner(arrow_c, FX).rows(case["plan"]) == case["expected"], f"{case['name']} ({case['source']})"
    assert Runner(arrow_c, FX).text(case["plan"]) == render_rows(case["expected"])
