"""The N>1 path on CPU: world_size 2, gloo.  Every rank hash-partitions its slice of fact and
dim with the numpy restatement of the device partition function, exchanges the slices with
the same all_to_all code the GPU bench uses, runs the local filter -> join -> group-by on the
CPU oracle, and rank 0 checks that the concatenated per-rank results equal the single-process
result (group key = join key => disjoint results)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

N_FACT, N_DIM = 20_000, 500


def make_tables():
    from sqlrs_amd import datagen
    fi, di = np.arange(N_FACT, dtype=np.int64), np.arange(N_DIM, dtype=np.int64)
    fact_key = datagen.key_np(0xF1, fi, 2 * N_DIM)  # half of the fact keys have no partner
    fact_val = datagen.val_np(0xF2, fi)
    dim_key = datagen.dim_key_np(di, N_DIM)
    return fact_key, fact_val, dim_key


def local_pipeline(oracle, dim_key, fact_key, fact_val):
    import pyarrow as pa
    from sqlrs_amd import abi
    from sqlrs_amd.executor import FilterExecutor, HashAggExecutor, HashJoinExecutor
    from sqlrs_amd.expr import AggFunc, Constant, InputRef, JoinCondition
    dim = pa.RecordBatch.from_arrays([pa.array(dim_key)], names=["key"])
    fact = pa.RecordBatch.from_arrays([pa.array(fact_key), pa.array(fact_val)], names=["key", "val"])
    schema = pa.schema([("d.key", pa.int64()), ("f.key", pa.int64()), ("f.val", pa.float64())])
    filt = FilterExecutor(oracle, InputRef(1) > Constant(0.5, abi.FLOAT64), [fact])
    join = HashJoinExecutor(oracle, [dim], filt.execute(), "inner", JoinCondition([(InputRef(0), InputRef(0))]), schema, 1)
    agg = HashAggExecutor(oracle, [AggFunc("count", InputRef(2), abi.INT64), AggFunc("sum", InputRef(2), abi.FLOAT64)],
                          [InputRef(0)], join.execute())
    outs = list(agg.execute())
    if not outs:
        return np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0)
    o = outs[0]
    return (np.array(o.column(0).to_pylist(), dtype=np.int64), np.array(o.column(1).to_pylist(), dtype=np.int64),
            np.array(o.column(2).to_pylist(), dtype=np.float64))


def merge_partials(oracle, keys, cnt, sm):
    """final merge of exchanged partial aggregates: SUM(count), SUM(sum) per key"""
    import pyarrow as pa
    from sqlrs_amd import abi
    from sqlrs_amd.executor import HashAggExecutor
    from sqlrs_amd.expr import AggFunc, InputRef
    if len(keys) == 0:
        return keys, cnt, sm
    b = pa.RecordBatch.from_arrays([pa.array(keys), pa.array(cnt), pa.array(sm)], names=["k", "c", "s"])
    (o,) = list(HashAggExecutor(oracle, [AggFunc("sum", InputRef(1), abi.INT64), AggFunc("sum", InputRef(2), abi.FLOAT64)],
                                [InputRef(0)], [b]).execute())
    return (np.array(o.column(0).to_pylist(), dtype=np.int64), np.array(o.column(1).to_pylist(), dtype=np.int64),
            np.array(o.column(2).to_pylist(), dtype=np.float64))


def local_partials(oracle, fact_key, fact_val):
    """a rank's partial aggregation below the exchange: HashAgg[GROUP BY key; COUNT(val), SUM(val)](Filter(fact slice))"""
    import pyarrow as pa
    from sqlrs_amd import abi
    from sqlrs_amd.executor import FilterExecutor, HashAggExecutor
    from sqlrs_amd.expr import AggFunc, Constant, InputRef
    fact = pa.RecordBatch.from_arrays([pa.array(fact_key), pa.array(fact_val)], names=["key", "val"])
    filt = FilterExecutor(oracle, InputRef(1) > Constant(0.5, abi.FLOAT64), [fact])
    outs = list(HashAggExecutor(oracle, [AggFunc("count", InputRef(1), abi.INT64), AggFunc("sum", InputRef(1), abi.FLOAT64)],
                                [InputRef(0)], filt.execute()).execute())
    if not outs:
        return np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0)
    o = outs[0]
    return (np.array(o.column(0).to_pylist(), dtype=np.int64), np.array(o.column(1).to_pylist(), dtype=np.int64),
            np.array(o.column(2).to_pylist(), dtype=np.float64))


def merge_join(oracle, dim_key, keys, cnt, sm):
    """the owning rank's merge: HashAgg[GROUP BY d.key; SUM(count), SUM(sum)](HashJoin(dim partition, partials))"""
    import pyarrow as pa
    from sqlrs_amd import abi
    from sqlrs_amd.executor import HashAggExecutor, HashJoinExecutor
    from sqlrs_amd.expr import AggFunc, InputRef, JoinCondition
    dim = pa.RecordBatch.from_arrays([pa.array(dim_key)], names=["key"])
    part = pa.RecordBatch.from_arrays([pa.array(keys), pa.array(cnt), pa.array(sm)], names=["k", "c", "s"])
    schema = pa.schema([("d.key", pa.int64()), ("p.k", pa.int64()), ("p.c", pa.int64()), ("p.s", pa.float64())])
    join = HashJoinExecutor(oracle, [dim], [part], "inner", JoinCondition([(InputRef(0), InputRef(0))]), schema, 1)
    outs = list(HashAggExecutor(oracle, [AggFunc("sum", InputRef(2), abi.INT64), AggFunc("sum", InputRef(3), abi.FLOAT64)],
                                [InputRef(0)], join.execute()).execute())
    if not outs:
        return np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0)
    o = outs[0]
    return (np.array(o.column(0).to_pylist(), dtype=np.int64), np.array(o.column(1).to_pylist(), dtype=np.int64),
            np.array(o.column(2).to_pylist(), dtype=np.float64))


def worker(rank, world, port, result_path, strategy="partition"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle_backend import load_oracle
    from sqlrs_amd import distributed as D
    oracle = load_oracle()
    fact_key, fact_val, dim_key = make_tables()
    f_lo, f_hi = D.shard_bounds(N_FACT, rank, world)
    d_lo, d_hi = D.shard_bounds(N_DIM, rank, world)

    def exchange(cols, chunks=1):
        """the bench's exchange: ChunkedExchange (split sizes over a side group, asynchronous payload
        collectives into one receive buffer), the local rows cut into `chunks` chunks"""
        n = len(cols[0])
        dts = [torch.from_numpy(c[:0].copy()).dtype for c in cols]
        ex = D.ChunkedExchange(dist, torch, world, dts, torch.device("cpu"), max(n // 3, 1), count_group=count_group)
        for c in range(chunks):
            lo, hi = n * c // chunks, n * (c + 1) // chunks
            parts, offs = D.partition_numpy([x[lo:hi] for x in cols], world)
            ex.send_chunk([torch.from_numpy(np.ascontiguousarray(p)) for p in parts], offs)
        outs = [o.numpy().copy() for o in ex.finish()]
        sent_off_rank.append(ex.bytes_off_rank)
        return outs

    def exchange_filtered(cols, keep, chunks):
        """the fused form (sqlrs_hash_partition_filter): Filter + partition in one step, per-partition
        regions with padding, slices sent straight out of the regions (ChunkedExchange.send_regions)"""
        n = len(cols[0])
        dts = [torch.from_numpy(c[:0].copy()).dtype for c in cols]
        ex = D.ChunkedExchange(dist, torch, world, dts, torch.device("cpu"), max(n // 5, 1), count_group=count_group)
        for c in range(chunks):
            lo, hi = n * c // chunks, n * (c + 1) // chunks
            regs, starts, rows = D.partition_filter_numpy([x[lo:hi] for x in cols], world, keep[lo:hi])
            ex.send_regions([torch.from_numpy(r) for r in regs], starts, rows)
        outs = [o.numpy().copy() for o in ex.finish()]
        sent_off_rank.append(ex.bytes_off_rank)
        return outs

    def exchange_abi_plan(cols):
        """the exchange as the C ABI does it (sqlrs_exchange_all_to_all, exchange.hip) with gloo standing in for RCCL:
        all-gather of the send counts -> sqlrs_exchange_plan (the library's own host arithmetic, called through
        ctypes: no device needed) -> per column one send / recv per peer into the planned slices of ONE output"""
        import ctypes as C
        from sqlrs_amd import build as B
        lib = C.CDLL(B.OUT)
        parts, offs = D.partition_numpy(cols, world)
        send_rows = torch.tensor([offs[p + 1] - offs[p] for p in range(world)], dtype=torch.int64)
        allc = [torch.empty(world, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(allc, send_rows)  # (ncclAllGather of the counts in the library)
        flat = (C.c_int64 * (world * world))(*[int(v) for row in allc for v in row.tolist()])
        rr, rs, tot = (C.c_int64 * world)(), (C.c_int64 * world)(), C.c_int64()
        lib.sqlrs_exchange_plan.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                            C.POINTER(C.c_int64)]
        assert lib.sqlrs_exchange_plan(world, rank, flat, rr, rs, C.byref(tot)) == 0
        assert list(rr) == [int(allc[q][rank]) for q in range(world)] and tot.value == sum(rr)
        assert lib.sqlrs_exchange_plan(world, world, flat, rr, rs, C.byref(tot)) != 0  # rank outside the world
        outs = []
        for col in parts:
            col = np.ascontiguousarray(col)
            out = np.empty(tot.value, dtype=col.dtype)
            sends = [torch.from_numpy(col[offs[p]:offs[p + 1]].copy()) for p in range(world)]
            recvs = [torch.from_numpy(out[rs[q]:rs[q] + rr[q]]) for q in range(world)]
            reqs = []
            for peer in range(world):  # (gloo has no all-to-all-v: isend / irecv pairs, like the grouped ncclSend / ncclRecv)
                if peer == rank:
                    recvs[peer].copy_(sends[peer])
                    continue
                reqs.append(dist.isend(sends[peer], peer))
                reqs.append(dist.irecv(recvs[peer], peer))
            for r in reqs:
                r.wait()
            outs.append(out)
        sent_off_rank.append(sum(int(send_rows[p]) for p in range(world) if p != rank) * sum(c.itemsize for c in parts))
        return outs

    count_group = dist.new_group(backend="gloo")
    sent_off_rank = []

    if strategy == "broadcast":
        # all-gather the dim, aggregate the local fact slice, exchange + merge partial aggregates
        parts = [torch.empty(D.shard_bounds(N_DIM, r, world)[1] - D.shard_bounds(N_DIM, r, world)[0], dtype=torch.int64)
                 for r in range(world)]
        dist.all_gather(parts, torch.from_numpy(np.ascontiguousarray(dim_key[d_lo:d_hi])))
        dk = torch.cat(parts).numpy()
        assert (dk == dim_key).all()
        pk, pc, ps = local_pipeline(oracle, dk, fact_key[f_lo:f_hi], fact_val[f_lo:f_hi])
        rk, rc, rs = exchange([pk, pc, ps])
        assert (D.partition_of(rk, world) == rank).all()
        keys, cnt, sm = merge_partials(oracle, rk, rc, rs)
        fk = fact_key[f_lo:f_hi]
        dk = dim_key[d_lo:d_hi]
    elif strategy in ("combine", "combine_abi"):
        # partitioned join with the partial aggregation below the exchange (bench.py: step_combine)
        xchg = exchange_abi_plan if strategy == "combine_abi" else exchange
        (dk,) = xchg([dim_key[d_lo:d_hi]])
        pk, pc, ps = local_partials(oracle, fact_key[f_lo:f_hi], fact_val[f_lo:f_hi])
        fk, rc, rs = xchg([pk, pc, ps]) if strategy == "combine_abi" else exchange([pk, pc, ps], chunks=2)
        assert (D.partition_of(dk, world) == rank).all() and (D.partition_of(fk, world) == rank).all()
        keys, cnt, sm = merge_join(oracle, dk, fk, rc, rs)
    else:
        (dk,) = exchange([dim_key[d_lo:d_hi]])
        if strategy == "partition_fused":  # Filter below the exchange, fused into the partition step
            fk, fv = exchange_filtered([fact_key[f_lo:f_hi], fact_val[f_lo:f_hi]], fact_val[f_lo:f_hi] > 0.5, chunks=3)
            assert (fv > 0.5).all()
            n_kept = torch.tensor([len(fk)])
            dist.all_reduce(n_kept)
            assert int(n_kept.item()) == int((fact_val > 0.5).sum())
        else:
            fk, fv = exchange([fact_key[f_lo:f_hi], fact_val[f_lo:f_hi]], chunks=3)  # receive buffer grows on the way
        assert sent_off_rank[-1] > 0
        # every key this rank received belongs to this rank's partition
        assert (D.partition_of(dk, world) == rank).all() and (D.partition_of(fk, world) == rank).all()
        keys, cnt, sm = local_pipeline(oracle, dk, fk, fv)
    gathered = [None] * world
    dist.all_gather_object(gathered, (keys, cnt, sm, len(fk), len(dk)))
    if rank == 0:
        all_keys = np.concatenate([g[0] for g in gathered])
        all_cnt = np.concatenate([g[1] for g in gathered])
        all_sum = np.concatenate([g[2] for g in gathered])
        if not strategy.startswith("combine"):  # (combine exchanges partial groups, not rows)
            assert sum(g[3] for g in gathered) == (N_FACT if strategy != "partition_fused" else int((fact_val > 0.5).sum()))
        assert sum(g[4] for g in gathered) == N_DIM
        assert len(np.unique(all_keys)) == len(all_keys), "per-rank results must be disjoint"
        ek, ec, es = local_pipeline(oracle, dim_key, fact_key, fact_val)  # single process
        o1, o2 = np.argsort(all_keys), np.argsort(ek)
        assert (all_keys[o1] == ek[o2]).all()
        assert (all_cnt[o1] == ec[o2]).all()
        assert np.allclose(all_sum[o1], es[o2], rtol=1e-9, atol=0)
        with open(result_path, "w") as f:
            f.write(f"ok {len(ek)}")
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("strategy", ["partition", "partition_fused", "broadcast", "combine", "combine_abi"])
def test_partitioned_join_groupby_world2_gloo(tmp_path, strategy):
    result = tmp_path / "result.txt"
    mp.spawn(worker, args=(2, free_port(), str(result), strategy), nprocs=2, join=True)
    assert result.read_text().startswith("ok")


def test_partition_function_is_balanced_and_total():
    from sqlrs_amd import distributed as D
    keys = np.arange(100_000, dtype=np.int64)
    for parts in (1, 2, 3, 8):
        p = D.partition_of(keys, parts)
        assert p.min() >= 0 and p.max() < parts
        counts = np.bincount(p, minlength=parts)
        assert counts.min() > 0.9 * len(keys) / parts
    cols, offs = D.partition_numpy([keys, keys * 2], 8)
    assert offs[0] == 0 and offs[-1] == len(keys)
    for p in range(8):  # stable inside a partition
        seg = cols[0][offs[p]:offs[p + 1]]
        assert (np.diff(seg) > 0).all() and (D.partition_of(seg, 8) == p).all()
    assert (cols[1] == cols[0] * 2).all()
