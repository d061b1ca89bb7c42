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
    nulls = pa.RecordBatch.from_arrays([pa.array([1, None, 3], type=pa.int64())], names=["k"])
    text = pa.RecordBatch.from_arrays([pa.array(["a", "b", "c"])], names=["s"])
    for b in (nulls, text):
        d = hip.to_device(b)
        with pytest.raises(abi.ExecutorError):
            hip.exchange_all_to_all(xchg, d, [0], [3])
        d.release()


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
