"""The exchange behind the C ABI (exchange.hip: sqlrs_exchange_* over RCCL on the ctx stream, no torch) on ONE GPU:
a communicator of one rank carries real ncclAllGather / grouped ncclSend + ncclRecv traffic on the hardware, the
bookkeeping (counts, receive offsets, output batch) is the N-rank code.  World 2 over gloo drives the same plan function on
CPU (tests/test_distributed_cpu.py::combine_abi); a second real GPU has never been available to this repo."""
import numpy as np
import pyarrow as pa
import pytest

from sqlrs_amd import abi
from sqlrs_amd.executor import FilterExecutor, HashAggExecutor, HashJoinAggExecutor, HashJoinExecutor
from sqlrs_amd.expr import AggFunc, Constant, InputRef, JoinCondition

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def xchg(hip):
    h = hip.exchange_create(hip.exchange_unique_id(), 0, 1)
    yield h
    hip.fn("exchange_destroy")(h)


def test_exchange_world_1_round_trip(hip, xchg):
    """hash partition (1 part) -> all-to-all with itself: the received batch is the partitioned batch, column by column"""
    rng = np.random.default_rng(3)
    n = 1_000_003
    b = pa.RecordBatch.from_arrays([pa.array(rng.integers(0, 50_000, n, dtype=np.int64)), pa.array(rng.random(n)),
                                    pa.array(rng.integers(-5, 5, n).astype(np.int32))], names=["k", "v", "w"])
    parts, offs = hip.hash_partition(hip.to_device(b), InputRef(0), 1, abi.MEM_DEVICE)
    got, recv = hip.exchange_all_to_all(xchg, parts, offs[:1], [offs[1] - offs[0]])
    assert recv == [n] and got.num_rows == n
    g, p = hip.to_host(got).to_arrow(["k", "v", "w"]), hip.to_host(parts).to_arrow(["k", "v", "w"])
    assert g.equals(p)
    assert hip.fn("exchange_bytes_off_rank")(xchg) == 0  # nothing left the rank
    got.release()
    parts.release()


def test_exchange_sub_range_and_empty(hip, xchg):
    """part_start / part_rows select a slice of the batch (the regions of sqlrs_hash_partition_filter have padding
    around them); an empty partition sends and receives nothing"""
    b = pa.RecordBatch.from_arrays([pa.array(np.arange(1000, dtype=np.int64)), pa.array(np.arange(1000) * 0.5)], names=["k", "v"])
    d = hip.to_device(b)
    got, recv = hip.exchange_all_to_all(xchg, d, [100], [250])
    t = hip.to_host(got).to_arrow(["k", "v"])
    assert recv == [250] and t.column(0).to_pylist() == list(range(100, 350)) and t.column(1).to_pylist()[0] == 50.0
    got.release()
    got, recv = hip.exchange_all_to_all(xchg, d, [0], [0])
    assert recv == [0] and got.num_rows == 0
    got.release()
    with pytest.raises(abi.ExecutorError):
        hip.exchange_all_to_all(xchg, d, [900], [200])  # a partition outside the batch
    d.release()


def test_exchange_rejects_what_it_cannot_carry(hip, xchg):
    """a Utf8 column: the rank says so in the count all-gather, the call fails (on every rank alike) and nothing is sent;
    the communicator is usable afterwards"""
    text = pa.RecordBatch.from_arrays([pa.array(["a", "b", "c"])], names=["s"])
    d = hip.to_device(text)
    with pytest.raises(abi.ExecutorError):
        hip.exchange_all_to_all(xchg, d, [0], [3])
    d.release()
    ok = hip.to_device(pa.RecordBatch.from_arrays([pa.array([1, 2, 3], type=pa.int64())], names=["k"]))
    got, recv = hip.exchange_all_to_all(xchg, ok, [0], [3])
    assert recv == [3] and hip.to_host(got).to_arrow(["k"]).column(0).to_pylist() == [1, 2, 3]
    got.release()
    ok.release()


def test_exchange_carries_validity(hip, xchg):
    """NULLs travel (a byte per row beside the values): slices that start inside a bitmap byte, a column without NULLs next to
    nullable ones, int32; against the same slice taken on the host"""
    rng = np.random.default_rng(12)
    n = 100_003
    k = pa.array(rng.integers(0, 1000, n, dtype=np.int64), mask=rng.random(n) < 0.1)
    v = pa.array(rng.random(n), mask=rng.random(n) < 0.3)
    w = pa.array(rng.integers(-9, 9, n).astype(np.int32))
    u = pa.array(rng.integers(-9, 9, n).astype(np.int32), mask=rng.random(n) < 0.5)
    b = pa.RecordBatch.from_arrays([k, v, w, u], names=["k", "v", "w", "u"])
    d = hip.to_device(b)
    for start, rows in ((0, n), (13, 70_001), (n - 5, 5)):
        got, recv = hip.exchange_all_to_all(xchg, d, [start], [rows])
        assert recv == [rows]
        t = hip.to_host(got).to_arrow(["k", "v", "w", "u"])
        assert t.equals(b.slice(start, rows)), (start, rows)
        got.release()
    d.release()


@pytest.mark.parametrize("capacity", [0, 10_000_000])
def test_exchange_chunk_sequence(hip, xchg, capacity):
    """sqlrs_exchange_begin / send_chunk / finish: chunks of different sizes (an empty one, one whose column has NULLs after
    chunks without any, library-owned and pyarrow-built device batches) arrive as ONE batch in chunk order; the receive
    batch grows from a small capacity hint; the payload of chunk k goes out with the count words of chunk k + 1"""
    rng = np.random.default_rng(13)
    hip.exchange_begin(xchg, [abi.INT64, abi.FLOAT64], capacity)
    exp = []
    keep = []
    for ci, n in enumerate([50_000, 0, 300_001, 7, 120_000]):
        mask = (rng.random(n) < 0.2) if ci >= 2 else None
        b = pa.RecordBatch.from_arrays([pa.array(rng.integers(0, 1 << 40, n, dtype=np.int64)), pa.array(rng.random(n), mask=mask)], names=["k", "v"])
        d = hip.to_device(b)
        lo = min(3, n)
        hip.exchange_send_chunk(xchg, d, [lo], [n - lo])
        exp.append(b.slice(lo))
        keep.append(d)
    got = hip.exchange_finish(xchg)
    want = pa.Table.from_batches(exp).combine_chunks()
    t = hip.to_host(got).to_arrow(["k", "v"])
    assert got.num_rows == want.num_rows
    assert pa.Table.from_batches([t]).equals(want)
    got.release()
    for d in keep:
        d.release()
    # a sequence is over when finish returns: the single-shot call works again, a second finish does not
    with pytest.raises(abi.ExecutorError):
        hip.exchange_finish(xchg)
    d = hip.to_device(pa.RecordBatch.from_arrays([pa.array([5, 6], type=pa.int64())], names=["k"]))
    g2, _ = hip.exchange_all_to_all(xchg, d, [0], [2])
    assert g2.num_rows == 2
    g2.release()
    d.release()


def test_exchange_chunk_sequence_error_reaches_the_next_call(hip, xchg):
    """a chunk whose columns do not match the sequence's types raises its flag in ITS count words: the call that sends its
    payload (the next send_chunk, or finish) fails — on every rank in the same call — and the sequence is over"""
    hip.exchange_begin(xchg, [abi.INT64], 1000)
    good = hip.to_device(pa.RecordBatch.from_arrays([pa.array([1, 2, 3], type=pa.int64())], names=["k"]))
    bad = hip.to_device(pa.RecordBatch.from_arrays([pa.array([1.0, 2.0])], names=["v"]))
    hip.exchange_send_chunk(xchg, good, [0], [3])
    hip.exchange_send_chunk(xchg, bad, [0], [2])  # (sends the payload of `good`; its own flag is on the wire)
    with pytest.raises(abi.ExecutorError):
        hip.exchange_finish(xchg)
    good.release()
    bad.release()
    hip.exchange_begin(xchg, [abi.INT64], 1000)  # a new sequence starts clean
    ok = hip.to_device(pa.RecordBatch.from_arrays([pa.array([9], type=pa.int64())], names=["k"]))
    hip.exchange_send_chunk(xchg, ok, [0], [1])
    got = hip.exchange_finish(xchg)
    assert hip.to_host(got).to_arrow(["k"]).column(0).to_pylist() == [9]
    got.release()
    ok.release()


def test_partitioned_join_group_by_through_the_exchange(hip, oracle, xchg):
    """the multi-GPU plan of the headline query with a world of one rank, every step through the C ABI: Filter + hash
    partition of the fact rows in one pass -> exchange; hash partition of the dim -> exchange; HashJoinAgg over what
    arrived — per group against the oracle running Filter -> HashJoin -> HashAgg on the original tables"""
    rng = np.random.default_rng(11)
    nd, nf = 40_000, 2_500_000
    dim = pa.RecordBatch.from_arrays([pa.array(rng.permutation(nd).astype(np.int64))], names=["key"])
    fact = pa.RecordBatch.from_arrays([pa.array(rng.integers(0, int(nd * 1.2), nf, dtype=np.int64)), pa.array(rng.random(nf))],
                                      names=["key", "val"])
    pred = InputRef(1) > Constant(0.5, abi.FLOAT64)
    fparts, starts, rows = hip.hash_partition_filter(hip.to_device(fact), InputRef(0), pred, 1, abi.MEM_DEVICE)
    frecv, _ = hip.exchange_all_to_all(xchg, fparts, starts, rows)
    dparts, doffs = hip.hash_partition(hip.to_device(dim), InputRef(0), 1, abi.MEM_DEVICE)
    drecv, _ = hip.exchange_all_to_all(xchg, dparts, doffs[:1], [doffs[1] - doffs[0]])
    assert frecv.num_rows == int((fact.column(1).to_numpy() > 0.5).sum())
    sch = pa.schema([("d.key", pa.int64()), ("f.key", pa.int64()), ("f.val", pa.float64())])
    cond = JoinCondition([(InputRef(0), InputRef(0))])
    aggs = [AggFunc("count", InputRef(2), abi.INT64), AggFunc("sum", InputRef(2), abi.FLOAT64)]
    got = pa.Table.from_batches(list(HashJoinAggExecutor(hip, [drecv], [frecv], cond, sch, 1, aggs, [InputRef(0)]).execute()))
    filt = FilterExecutor(oracle, pred, [fact])
    join = HashJoinExecutor(oracle, [dim], filt.execute(), "inner", cond, sch, 1)
    exp = pa.Table.from_batches(list(HashAggExecutor(oracle, aggs, [InputRef(0)], join.execute()).execute()))
    # (the order inside a partition after the fused filter + partition is unspecified: compare per group, by key)
    gk, ek = got.column(0).to_numpy(), exp.column(0).to_numpy()
    go, eo = np.argsort(gk), np.argsort(ek)
    assert (gk[go] == ek[eo]).all()
    assert (got.column(1).to_numpy()[go] == exp.column(1).to_numpy()[eo]).all()  # COUNT: bit exact
    gs, es = got.column(2).to_numpy()[go], exp.column(2).to_numpy()[eo]
    assert (np.abs(gs - es) <= 1e-9 * np.abs(es)).all()  # SUM(double): 1e-9 relative
    for x in (fparts, frecv, dparts, drecv):
        x.release()
