#!/usr/bin/env python
"""bench.py — the hot path of BASELINE.json measured on MI355X through the C ABI.

Workload (config C5, the configuration the metric is quoted on; it fits one GPU):

    SELECT d.key, COUNT(f.val), SUM(f.val)
    FROM fact f JOIN dim d ON f.key = d.key          -- dim = build (left), fact = probe (right)
    WHERE f.val > 0.5                                 -- FilterExecutor, selectivity 0.5
    GROUP BY d.key                                    -- HashAggExecutor, 1e7 groups

fact = 1e9 rows (key int64, val float64), dim = 1e7 rows (key int64), synthetic SplitMix64
columns generated in HBM.  One "step" = one pass of Filter -> HashJoin(build+probe) -> HashAgg
over the whole input; inputs are HBM resident when the timed region starts.  `value` is
fact rows / s summed over all ranks (Mrows/s).  The plan shape is HashAgg(HashJoin(dim, Filter(fact))):
the fused HashJoinAgg operator with the Filter handed to it (sqlrs_join_agg_set_probe_filter).

N > 1 (one process per GPU, launched by torch.distributed.run): the TOTAL input is fixed
("scaling": "strong"); every rank generates a contiguous 1/N slice of fact and dim.  Default
strategy (`combine`) = north_star's partitioned hash join with the partial aggregation below the exchange: the dim
is hash-partitioned on the join key (sqlrs_hash_partition) and exchanged with an RCCL all-to-all, every rank
pre-aggregates its fact slice by key (HashAgg with the Filter handed to it), the partial aggregates are
hash-partitioned + exchanged the same way and merged by HashJoinAgg(dim partition, partials) on the owning
rank (group key = join key, so the per-rank results are disjoint).  `--exchange partition` = the row form (Filter
below the exchange, kept fact ROWS partitioned and exchanged in overlapped chunks, local HashJoinAgg);
`--exchange broadcast` = all-gather the dim, aggregate locally, exchange + merge the partial aggregates; the line
times the other two strategies too (`exchange.alternative`, `exchange.alternative2`).

Output: ONE JSON line on rank 0 (contract in the task description) with `roofline` (dominant kernel:
HIP-event time per launch measured live on the ctx stream; and the operator-level pipeline figure),
`cpu_baseline` (all-core CPU port + the single-threaded restatement of the reference, on bounded samples),
and at N = 1 `c5_variants` (sparse keys, three operators, duplicate build keys, GROUP BY a dim attribute, adversarial inputs) and `operators` (C2 / C3 / C4 / Order).
Every result is checked per group before it is timed.
"""
from __future__ import annotations

import os as _os; _os.environ.setdefault("SQLRS_HOOKS", "1")  # the SQLRS_* tuning hooks are consulted only in a process that opts in (common.hpp: hook)
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# --------------------------------------------------------------------------------------
def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--rows", type=float, default=float(os.environ.get("SQLRS_BENCH_ROWS", 1e9)),
                   help="total fact rows over all ranks (C5: 1e9)")
    p.add_argument("--dim-rows", type=float, default=float(os.environ.get("SQLRS_BENCH_DIM", 1e7)),
                   help="total dim rows = number of groups (C5: 1e7)")
    p.add_argument("--threshold", type=float, default=0.5, help="WHERE f.val > threshold")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-sample-rows", type=float, default=0, help="0 = auto (about 10-30 s)")
    p.add_argument("--exchange", choices=["auto", "combine", "partition", "broadcast"], default="auto",
                   help="N>1: 'partition' = hash-partition the kept fact ROWS and the dim on the join key + all-to-all "
                        "(partitioned hash join, every kept row crosses xGMI); 'combine' = the same partitioned join with the "
                        "group-by's partial aggregation pushed below the exchange: the dim is hash-partitioned + exchanged, "
                        "every rank pre-aggregates its fact slice by key (Filter fused), the partial aggregates are "
                        "hash-partitioned + exchanged and merged by the join on the owning rank; 'broadcast' = all-gather the "
                        "dim keys instead of partitioning them, join + aggregate the local fact slice, exchange + merge the "
                        "partial aggregates; 'auto' = combine")
    p.add_argument("--exchange-impl", choices=["abi", "torch"], default="abi",
                   help="N>1: who carries the single-chunk all-to-alls (the dim keys, the partial aggregates of `combine`, the "
                        "partials of `broadcast`): 'abi' = sqlrs_exchange_all_to_all (exchange.hip: RCCL on the ctx stream, "
                        "torch-free; the ncclUniqueId travels over the gloo side group), 'torch' = torch.distributed "
                        "all_to_all_single.  The chunked row exchange of `partition` runs through sqlrs_exchange_begin / send_chunk / finish with 'abi'")
    p.add_argument("--force-exchange", action="store_true",
                   help="N=1 only: run the multi-GPU code path (RCCL process group of ONE rank, fused filter + hash "
                        "partition, all-to-all, stream hand-over, local HashJoinAgg) instead of the single-GPU step; "
                        "a hardware smoke of the exchange path, never the headline configuration")
    p.add_argument("--unfused", action="store_true",
                   help="run HashJoin and HashAgg as two operators (joined batch materialised in HBM)")
    p.add_argument("--separate-filter", action="store_true",
                   help="run the Filter as its own operator in front of the fused HashJoinAgg (round-1 shape)")
    p.add_argument("--operators", action="store_true", help="(default at N=1; kept for old command lines)")
    p.add_argument("--no-operators", action="store_true",
                   help="skip the per-operator configs C2/C3/C4/Order and the C5 variants (sparse keys, three "
                        "operators) that the default N=1 run times after the headline measurement")
    p.add_argument("--cpu-threads", type=int, default=0, help="threads of the all-core CPU baseline (0 = all)")
    p.add_argument("--processes", type=int, default=3,
                   help="N=1: ms_per_step is ALSO measured in this many fresh processes in total (this one + children run "
                        "after its own timed region); the line carries their min / median / max: the partition kernels' "
                        "time follows how the driver backs the big buffers, which differs from process to process")
    p.add_argument("--child", action="store_true", help=argparse.SUPPRESS)  # a --processes child: headline timing only
    return p.parse_args()


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this command line as N ranks under
    torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1 at a free port); rank 0's JSON
    line is this process's stdout.  Returns the launcher's exit code."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    log("[bench] no launcher in the environment: starting", " ".join(cmd))
    return subprocess.call(cmd, env=env)


def device_batch(abi, tensors, dtypes):
    cols = [abi.device_column(dt, t.numel(), t.data_ptr()) for t, dt in zip(tensors, dtypes)]
    return abi.RawBatch(cols, tensors[0].numel(), keepalive=tensors)


class Pipeline:
    """Filter -> HashJoin -> HashAgg driven through the C ABI on device-resident batches."""

    def __init__(self, be, abi, threshold, fused=True, partial=False, fuse_filter=True):
        from sqlrs_amd.expr import AggFunc, Constant, InputRef, JoinCondition
        self.be, self.abi, self.fused, self.fused_batches = be, abi, fused, 0
        self.fuse_filter, self.filter_fused_batches = fuse_filter, 0
        # partial = the groups are exchanged and merged again (multi-GPU broadcast strategy): their
        # order is irrelevant, so the first-seen ordering of the local result is skipped
        self.partial = partial
        self.filter_expr = (InputRef(1) > Constant(threshold, abi.FLOAT64)).pack()
        self.cond = JoinCondition([(InputRef(0), InputRef(0))])
        self.lk, self._k1 = abi.pack_exprs([InputRef(0)])
        self.rk, self._k2 = abi.pack_exprs([InputRef(0)])
        self.right_dtypes = (C.c_int32 * 2)(abi.INT64, abi.FLOAT64)
        self.gb, self._k3 = abi.pack_exprs([InputRef(0)])  # d.key
        keep = []
        self.aggs = (abi.AggFunc * 2)(AggFunc("count", InputRef(2), abi.INT64).abi_struct(keep),
                                      AggFunc("sum", InputRef(2), abi.FLOAT64).abi_struct(keep))
        self._keep = keep

    def filter(self, fact_b):
        """Filter alone; returns the device-resident kept rows (LibBatch)"""
        be, abi = self.be, self.abi
        f = C.c_void_p()
        be.check(be.fn("filter_create")(be.ctx, C.byref(self.filter_expr.abi), C.byref(f)))
        fo = C.POINTER(abi.Batch)()
        be.check(be.fn("filter_push")(f, fact_b.ptr, abi.MEM_DEVICE, C.byref(fo)))
        be.fn("filter_destroy")(f)
        return be.wrap(fo)

    def step(self, dim_b, fact_b):
        """one pass; returns the device-resident result batch (LibBatch)"""
        if self.fused and self.fuse_filter:
            # HashAgg(HashJoin(dim, Filter(fact))) as ONE operator: the Filter is handed to the fused
            # join+aggregate (sqlrs_join_agg_set_probe_filter), which evaluates it in its first partition pass
            return self.join_agg(dim_b, fact_b, probe_filter=True)
        return self.join_agg(dim_b, self.filter(fact_b))

    def join_agg(self, dim_b, filtered, probe_filter=False):
        """HashJoin + HashAgg over already filtered fact rows (`filtered` is released); with
        `probe_filter` the rows are unfiltered and the operator applies the Filter itself"""
        be, abi = self.be, self.abi
        D = abi.MEM_DEVICE
        if self.fused:
            # HashAgg directly over the Inner HashJoin (sqlrs_join_agg_*): same result, the joined
            # batch is not materialised when the library's fused route applies
            ja = C.c_void_p()
            be.check(be.fn("join_agg_create")(be.ctx, 1, self.lk, self.rk, 1, 2, self.right_dtypes, 1, self.gb, 2,
                                              self.aggs, C.byref(ja)))
            if self.partial:
                be.check(be.fn("join_agg_set_group_order")(ja, abi.GROUP_ORDER_ANY))
            if probe_filter:
                be.check(be.fn("join_agg_set_probe_filter")(ja, C.byref(self.filter_expr.abi)))
            be.check(be.fn("join_agg_build_push")(ja, dim_b.ptr))
            be.check(be.fn("join_agg_build_finish")(ja))
            be.check(be.fn("join_agg_probe_push")(ja, filtered.ptr))
            filtered.release()
            ao = C.POINTER(abi.Batch)()
            be.check(be.fn("join_agg_finish")(ja, D, C.byref(ao)))
            self.fused_batches = be.fn("join_agg_fused_batches")(ja)
            self.filter_fused_batches = be.fn("join_agg_filter_fused_batches")(ja)
            be.fn("join_agg_destroy")(ja)
            return be.wrap(ao)
        j = C.c_void_p()
        be.check(be.fn("hash_join_create")(be.ctx, abi.JOIN_INNER, 1, self.lk, self.rk, None, 2,
                                           self.right_dtypes, C.byref(j)))
        be.check(be.fn("hash_join_build_push")(j, dim_b.ptr))
        be.check(be.fn("hash_join_build_finish")(j))
        jo = C.POINTER(abi.Batch)()
        be.check(be.fn("hash_join_probe_push")(j, filtered.ptr, D, C.byref(jo)))
        joined = be.wrap(jo)
        filtered.release()
        be.fn("hash_join_destroy")(j)
        a = C.c_void_p()
        be.check(be.fn("hash_agg_create")(be.ctx, 1, self.gb, 2, self.aggs, C.byref(a)))
        if self.partial:
            be.check(be.fn("hash_agg_set_group_order")(a, abi.GROUP_ORDER_ANY))
        be.check(be.fn("hash_agg_push")(a, joined.ptr))
        joined.release()
        ao = C.POINTER(abi.Batch)()
        be.check(be.fn("hash_agg_finish")(a, D, C.byref(ao)))
        be.fn("hash_agg_destroy")(a)
        return be.wrap(ao)


def expected_groups(torch, dist, fact_key, fact_val, dim_key, threshold, n_keys, key_of=None, chunk=1 << 27):
    """Per-group expectation of the query from the raw columns with plain torch ops (independent of
    the library): exp_cnt[k], exp_sum[k] over fact rows with val > threshold (all ranks when `dist`),
    has_dim[k] = k occurs in dim.  Keys must lie in [0, n_keys) (`key_of` maps a column chunk to
    that range when the bench transformed the keys)."""
    dev = fact_key.device
    exp_cnt = torch.zeros(n_keys, dtype=torch.int64, device=dev)
    exp_sum = torch.zeros(n_keys, dtype=torch.float64, device=dev)
    has_dim = torch.zeros(n_keys, dtype=torch.int64, device=dev)
    kept = 0
    for lo in range(0, fact_key.numel(), chunk):
        k, v = fact_key[lo:lo + chunk], fact_val[lo:lo + chunk]
        m = v > threshold
        km = (key_of(k[m]) if key_of else k[m])
        kept += int(km.numel())
        exp_cnt += torch.bincount(km, minlength=n_keys)
        exp_sum.index_add_(0, km, v[m])
    dk = key_of(dim_key) if key_of else dim_key
    has_dim += torch.bincount(dk, minlength=n_keys)
    if dist is not None:
        for t in (exp_cnt, exp_sum, has_dim):
            dist.all_reduce(t)
    return exp_cnt, exp_sum, has_dim, kept


def check_groups(torch, dist, dev, out, exp_cnt, exp_sum, has_dim, key_of=None):
    """Every group of the (per-rank) result batch `out` [key, COUNT, SUM] against the expectation:
    keys distinct and joined (has_dim, count > 0), COUNT bit-exact, SUM within 1e-9 relative, and
    the groups of all ranks together are exactly the keys that have kept rows and a build partner."""
    g = out.num_rows
    keys = _tensor_view(torch, out.column(0).values, g, torch.int64, dev)
    cnt = _tensor_view(torch, out.column(1).values, g, torch.int64, dev)
    sm = _tensor_view(torch, out.column(2).values, g, torch.float64, dev)
    k = key_of(keys) if key_of else keys
    in_range = bool(((k >= 0) & (k < exp_cnt.numel())).all().item()) if g else True
    bad_cnt = bad_sum = dup = -1
    if in_range:
        seen = torch.bincount(k, minlength=exp_cnt.numel()) if g else torch.zeros_like(exp_cnt)
        dup = int((seen > 1).sum().item())
        mult = has_dim[k]  # duplicate build keys multiply the joined rows (hash_join.rs:225-234)
        bad_cnt = int((cnt != exp_cnt[k] * mult).sum().item()) + int((mult == 0).sum().item())
        e = exp_sum[k] * mult.to(torch.float64)
        bad_sum = int(((sm - e).abs() > 1e-9 * e.abs().clamp_min(1e-300)).sum().item())
    totals = torch.tensor([g, int(cnt.sum().item()) if g else 0, max(bad_cnt, 0), max(bad_sum, 0), max(dup, 0),
                           0 if in_range else 1], dtype=torch.int64, device=dev)
    if dist is not None:
        dist.all_reduce(totals)
    expected_groups_n = int(((exp_cnt > 0) & (has_dim > 0)).sum().item())
    expected_rows = int((exp_cnt * has_dim)[has_dim > 0].sum().item())  # duplicate build keys multiply rows
    groups, rows, bc, bs, dp, oor = [int(x) for x in totals.tolist()]
    ok = oor == 0 and bc == 0 and bs == 0 and dp == 0 and groups == expected_groups_n and rows == expected_rows
    msg = (f"groups {groups:,} (expected {expected_groups_n:,}), joined rows {rows:,} (expected {expected_rows:,}), "
           f"count mismatches {bc}, sum mismatches {bs}, duplicate keys {dp}, keys out of range {oor}")
    return ok, groups, rows, msg


# algorithmic HBM bytes per launch of the kernels that can dominate
def algorithmic_bytes(kernel, w):
    nP, nB, s, M, G = w["fact_rows"], w["dim_rows"], w["selectivity"], w["matches"], w["groups"]
    rec = 16 if os.environ.get("SQLRS_RP_SLIM") == "0" else 12
    table = {
        "filter_cmp_const": 8 * nP + 8 * s * nP,            # read predicate column, write kept values
        "compact": 8 * nP + 8 * s * nP,                     # second column through the same selection
        "join_probe_count": 8 * s * nP,                     # probe keys read
        "join_probe_fill": 8 * s * nP + 12 * M,             # probe keys read, (u64,u32) pairs written
        "gather": 12 * M + 8 * M,                           # index read + gathered column written (per column)
        "agg_resolve": 8 * M,                               # group keys read
        "agg_update": 12 * M,                               # group id + value read per row
        "join_build": 8 * nB,
        "join_probe_unique": 8 * s * nP + 12 * M,           # one pass: keys read, pairs written
        # (rec = bytes of a partitioned row: 12 in the slim form — value + 32-bit word, radix_part.hip — 16 with SQLRS_RP_SLIM=0)
        "rp_scatter": 2 * rec * M,                          # partitioned rows read + written
        "rp_chunk_scatter_filter": 16 * nP + rec * M,       # key + val of every row read, kept rows written once
        "rp_chunk_scatter": 16 * M + rec * M,
        "rp_hist": 8 * M,
        "lds_agg": rec * M + 28 * G,                        # partitioned rows read, groups written
        "normalize_keys": 16 * M,
    }
    return table.get(kernel)


def survey_bytes(kernel, w):
    """SURVEY.md 8(d) bytes of one launch: the operator-level per-row figure (16 B per probe row of the fused
    pipeline: key + value read; temporaries, hash tables and partition traffic EXCLUDED) x the rows the launch
    processes.  `roofline.frac` is computed from this; the kernel's own traffic model (algorithmic_bytes, which
    counts the intermediate it writes) is reported beside it as kernel_own_*."""
    nP, M = w["fact_rows"], w["matches"]
    table = {
        "rp_chunk_scatter_filter": 16 * nP,   # every probe row enters here (Filter fused in)
        "rp_chunk_scatter": 16 * M,           # (Filter ran before: the kept rows)
        "rp_scatter": 16 * M,                 # the kept rows, second partition level
        "lds_agg": 16 * M,                    # the kept rows, bucket pass
        "filter_cmp_const": 8 * nP + 8 * w["selectivity"] * nP,
    }
    return table.get(kernel)


def main():
    args = parse()
    # (read when the HSA runtime starts, i.e. before the first HIP call of this process and of the ranks it launches:
    #  the host driver only supports dmabuf IPC, RCCL's peer buffers fail with the legacy mode)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    import sqlrs_amd
    from sqlrs_amd import abi, datagen

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, same command line)
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench: --gpus {args.gpus} but WORLD_SIZE={world} (launch with --nproc-per-node {args.gpus}, "
                         f"or run `python bench.py --gpus {args.gpus}` without a launcher)")
    # multi = the exchange path runs: N > 1, or N = 1 with --force-exchange (RCCL group of one rank)
    multi = world > 1 or args.force_exchange
    # test hook: all ranks on GPU 0 with a gloo process group (exchange staged through the host) so
    # that the multi-rank logic can be exercised on a one-GPU box; never used for reported numbers
    single_dev = os.environ.get("SQLRS_BENCH_SINGLE_DEVICE") == "1"
    if single_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    json_fd = None
    if multi:
        # RCCL prints its version banner on STDOUT when the first communicator comes up: rank 0's stdout must hold the
        # JSON line and nothing else, so file descriptor 1 points at stderr from here on and the line goes to a copy
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:  # --force-exchange without a launcher: a rendezvous of one
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if single_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    rank_info = None
    if multi:
        # self-check: the process group really spans `world` ranks on `world` distinct GPUs over RCCL
        one = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(one)
        props = torch.cuda.get_device_properties(dev)
        ident = f"{os.uname().nodename}:{local_rank}:{getattr(props, 'uuid', '')}"
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        backend = dist.get_backend()
        if int(one.item()) != world or (not single_dev and (backend != "nccl" or len(set(idents)) != world)):
            raise SystemExit(f"bench: process group check failed: all_reduce(1) = {int(one.item())}, world {world}, "
                             f"backend {backend}, devices {idents}")
        rank_info = {"world": world, "backend": "nccl (RCCL)" if backend == "nccl" else backend,
                     "distinct_devices": len(set(idents)), "all_reduce_of_ones": int(one.item())}
    be = sqlrs_amd.new_ctx(local_rank)  # raises if the HIP library / GPU is missing: no fallback
    n_fact_total, n_dim_total = int(args.rows), int(args.dim_rows)
    # contiguous slice of the global tables owned by this rank
    f_lo, f_hi = n_fact_total * rank // world, n_fact_total * (rank + 1) // world
    d_lo, d_hi = n_dim_total * rank // world, n_dim_total * (rank + 1) // world
    t0 = time.time()
    fact_key = datagen.fill_chunks(torch.empty(f_hi - f_lo, dtype=torch.int64, device=dev),
                                   lambda i: datagen.key_t(0xF1, i, n_dim_total), f_lo)
    fact_val = datagen.fill_chunks(torch.empty(f_hi - f_lo, dtype=torch.float64, device=dev),
                                   lambda i: datagen.val_t(0xF2, i), f_lo)
    dim_key = datagen.fill_chunks(torch.empty(d_hi - d_lo, dtype=torch.int64, device=dev),
                                  lambda i: datagen.dim_key_t(i, n_dim_total), d_lo)
    torch.cuda.synchronize()
    if rank == 0:
        log(f"[bench] generated {f_hi - f_lo:,} fact rows + {d_hi - d_lo:,} dim rows per rank in {time.time() - t0:.1f}s")
    # expected result per group, computed by torch on the same columns (independent of the library):
    # exp_cnt[k] / exp_sum[k] over the kept fact rows of ALL ranks, has_dim[k] = key k has a build partner
    exp_cnt, exp_sum, has_dim, expected_kept = expected_groups(torch, dist if multi else None, fact_key, fact_val,
                                                               dim_key, args.threshold, n_dim_total)

    pipe = Pipeline(be, abi, args.threshold, fused=not args.unfused, fuse_filter=not args.separate_filter)

    def D_shard(total, r, w):
        return total * (r + 1) // w - total * r // w
    from sqlrs_amd import distributed as D
    from sqlrs_amd.expr import InputRef

    strategy = args.exchange
    if strategy == "auto":
        # north_star: build AND probe side hash-partitioned on the join key, RCCL all-to-all — with the probe side
        # pre-aggregated by key below the exchange (1e7 partial groups per rank instead of 5e8 / N kept rows: at
        # N = 2 the row form moves 2 GB per rank over ONE xGMI link, several times the local work)
        strategy = "combine"
    pipe.partial = multi and strategy == "broadcast"
    dim_sizes = [D_shard(n_dim_total, r, world) for r in range(world)]
    merge_gb, _mk = abi.pack_exprs([InputRef(0)])
    _mkeep = []
    from sqlrs_amd.expr import AggFunc as _AggFunc
    merge_aggs = (abi.AggFunc * 2)(_AggFunc("sum", InputRef(1), abi.INT64).abi_struct(_mkeep),
                                   _AggFunc("sum", InputRef(2), abi.FLOAT64).abi_struct(_mkeep))
    # split sizes of the all-to-alls travel over a CPU group: the host never waits for a payload collective
    count_group = dist.new_group(backend="gloo") if multi else None
    data_group = count_group if single_dev else None  # None = the default (RCCL) group
    wire_out = (lambda t: t.cpu()) if single_dev else None
    wire_in = (lambda t: t.to(dev)) if single_dev else None
    n_chunks = max(1, int(os.environ.get("SQLRS_BENCH_EXCHANGE_CHUNKS", "4")))
    xstat = {"on": False, "exchange_ms": 0.0, "bytes_off_rank": 0, "steps": 0}
    T_DT = {abi.INT64: torch.int64, abi.FLOAT64: torch.float64}

    def torch_stream():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    # the exchange behind the C ABI (exchange.hip): one communicator per rank over an id made by rank 0
    abi_x = {"h": None, "keep": [], "bytes0": 0}
    if multi and not single_dev and args.exchange_impl == "abi":
        ids = [be.exchange_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0, group=count_group)
        abi_x["h"] = be.exchange_create(ids[0], rank, world)

    def exchange_abi(cols, dtypes):
        """one chunk through sqlrs_hash_partition + sqlrs_exchange_all_to_all, all on the ctx stream (no torch stream
        hand-over: the operators that consume the result run on the same stream)"""
        b = device_batch(abi, cols, dtypes)
        be.check(be.fn("ctx_wait_stream")(be.ctx, torch_stream()))  # the columns may come from torch's stream
        parts, offs = be.hash_partition(b, InputRef(0), world, abi.MEM_DEVICE)
        got, _ = be.exchange_all_to_all(abi_x["h"], parts, offs[:-1], [offs[p + 1] - offs[p] for p in range(world)])
        abi_x["keep"] += [parts, got]  # (alive until the next step starts: the step's operators read the views below)
        sent = int(be.fn("exchange_bytes_off_rank")(abi_x["h"]))
        xstat["bytes_off_rank"] += sent - abi_x["bytes0"]
        abi_x["bytes0"] = sent
        return [_tensor_view(torch, got.column(ci).values, got.num_rows, T_DT[d], dev) for ci, d in enumerate(dtypes)]

    def exchange(chunks, dtypes, capacity):
        """`chunks`: iterable of lists of device columns (column 0 = the join key).  Every chunk is
        hash-partitioned on the key by the library; its all-to-all (one per column, asynchronous) runs
        while the next chunk is produced and partitioned.  Returns the received columns (one tensor each)."""
        if abi_x["h"] is not None and isinstance(chunks, list) and len(chunks) == 1:
            return exchange_abi(chunks[0], dtypes)
        ex = D.ChunkedExchange(dist, torch, world, [T_DT[d] for d in dtypes], dev, capacity, data_group=data_group,
                               count_group=count_group, wire_out=wire_out, wire_in=wire_in)
        keep = []
        for cols in chunks:
            b = device_batch(abi, cols, dtypes)
            parts, offs = be.hash_partition(b, InputRef(0), world, abi.MEM_DEVICE)
            # the collectives start behind torch's current stream: order it behind the partition kernels
            be.check(be.fn("ctx_release_to_stream")(be.ctx, torch_stream()))
            views = [_tensor_view(torch, parts.column(ci).values, parts.column(ci).length, T_DT[d], dev)
                     for ci, d in enumerate(dtypes)]
            ex.send_chunk(views, offs)
            keep.append((parts, b, cols))
        outs = ex.finish()  # torch's stream now waits for every collective ...
        be.check(be.fn("ctx_wait_stream")(be.ctx, torch_stream()))  # ... and the ctx stream for torch's
        for parts, _, _ in keep:
            parts.release()  # (pool blocks go to LATER ctx-stream work, i.e. behind the collectives that read them)
        xstat["bytes_off_rank"] += ex.bytes_off_rank
        return outs

    fused_exchange = os.environ.get("SQLRS_BENCH_EXCHANGE_FUSED", "1") != "0"

    def exchange_filtered_fact():
        """Filter + hash partition of the local fact slice in ONE pass per chunk (sqlrs_hash_partition_filter: every
        row read once, every kept row written once into its partition's region), the slices of chunk k travelling
        (list all-to-all straight out of the regions) while chunk k + 1 is filtered and partitioned."""
        from sqlrs_amd.expr import Constant
        dtypes = [abi.INT64, abi.FLOAT64]
        pred = InputRef(1) > Constant(args.threshold, abi.FLOAT64)
        nloc = fact_key.numel()
        if abi_x["h"] is not None:
            # the chunk sequence behind the C ABI (exchange.hip: sqlrs_exchange_begin / send_chunk / finish): everything on the
            # ctx stream, no torch on the data path — send_chunk(k) puts chunk k's count words on the wire and sends chunk
            # k - 1's payload, so the host never waits for the device inside the loop
            be.check(be.fn("ctx_wait_stream")(be.ctx, torch_stream()))  # the fact columns come from torch's stream
            be.exchange_begin(abi_x["h"], dtypes, int(nloc * 0.75) + 1024)
            keep = []
            for c in range(n_chunks):
                lo, hi = nloc * c // n_chunks, nloc * (c + 1) // n_chunks
                if hi == lo:
                    continue
                b = device_batch(abi, [fact_key[lo:hi], fact_val[lo:hi]], dtypes)
                parts, starts, rows = be.hash_partition_filter(b, InputRef(0), pred, world, abi.MEM_DEVICE)
                be.exchange_send_chunk(abi_x["h"], parts, starts, rows)
                keep.append((parts, b))
            got = be.exchange_finish(abi_x["h"])
            for parts, _ in keep:
                parts.release()  # (the exchange kept its own reference until the chunk's payload was queued)
            abi_x["keep"] += [got]
            sent = int(be.fn("exchange_bytes_off_rank")(abi_x["h"]))
            xstat["bytes_off_rank"] += sent - abi_x["bytes0"]
            abi_x["bytes0"] = sent
            return [_tensor_view(torch, got.column(ci).values, got.num_rows, T_DT[d], dev) for ci, d in enumerate(dtypes)]
        ex = D.ChunkedExchange(dist, torch, world, [T_DT[d] for d in dtypes], dev, int(fact_key.numel() * 0.75) + 1024,
                               data_group=data_group, count_group=count_group, wire_out=wire_out, wire_in=wire_in)
        keep = []
        for c in range(n_chunks):
            lo, hi = nloc * c // n_chunks, nloc * (c + 1) // n_chunks
            if hi == lo:
                continue
            b = device_batch(abi, [fact_key[lo:hi], fact_val[lo:hi]], dtypes)
            parts, starts, rows = be.hash_partition_filter(b, InputRef(0), pred, world, abi.MEM_DEVICE)
            be.check(be.fn("ctx_release_to_stream")(be.ctx, torch_stream()))
            views = [_tensor_view(torch, parts.column(ci).values, parts.column(ci).length, T_DT[d], dev)
                     for ci, d in enumerate(dtypes)]
            ex.send_regions(views, starts, rows)
            keep.append((parts, b))
        outs = ex.finish()
        be.check(be.fn("ctx_wait_stream")(be.ctx, torch_stream()))
        for parts, _ in keep:
            parts.release()
        xstat["bytes_off_rank"] += ex.bytes_off_rank
        return outs

    def filtered_chunks():
        """Filter below the exchange (only kept fact rows cross xGMI), chunk by chunk"""
        nloc = fact_key.numel()
        for c in range(n_chunks):
            lo, hi = nloc * c // n_chunks, nloc * (c + 1) // n_chunks
            if hi == lo:
                continue
            kept = pipe.filter(device_batch(abi, [fact_key[lo:hi], fact_val[lo:hi]], [abi.INT64, abi.FLOAT64]))
            kk = _tensor_view(torch, kept.column(0).values, kept.num_rows, torch.int64, dev)
            kv = _tensor_view(torch, kept.column(1).values, kept.num_rows, torch.float64, dev)
            yield [kk, kv]
            kept.release()  # (its partitioned copy is complete: sqlrs_hash_partition returned the offsets)

    def gather_dim():
        """all-gather of the dim keys (every rank ends up with the whole build side)"""
        if single_dev:
            parts = [torch.empty(n, dtype=torch.int64) for n in dim_sizes]
            dist.all_gather(parts, dim_key.cpu(), group=count_group)
            return torch.cat(parts).to(dev)
        parts = [torch.empty(n, dtype=torch.int64, device=dev) for n in dim_sizes]
        dist.all_gather(parts, dim_key)
        out = torch.cat(parts)
        be.check(be.fn("ctx_wait_stream")(be.ctx, torch_stream()))
        xstat["bytes_off_rank"] += 8 * (n_dim_total - dim_sizes[rank])
        return out

    def step_partition():
        # partitioned hash join (north_star): dim and kept fact rows are hash-partitioned on the join key and
        # exchanged; every rank then owns a disjoint key range and its local join + group-by result is final
        t0 = time.perf_counter()
        dk, = exchange([[dim_key]], [abi.INT64], D_shard(n_dim_total, rank, world) * 2 + 1024)
        if fused_exchange:
            fk, fv = exchange_filtered_fact()
        else:  # SQLRS_BENCH_EXCHANGE_FUSED=0: Filter operator, stable partition, contiguous all_to_all_single
            fk, fv = exchange(filtered_chunks(), [abi.INT64, abi.FLOAT64], int(fact_key.numel() * 0.75) + 1024)
        if xstat["on"]:
            torch.cuda.synchronize()
            be.synchronize()
            xstat["exchange_ms"] += (time.perf_counter() - t0) * 1e3
        out = pipe.join_agg(device_batch(abi, [dk], [abi.INT64]),
                            device_batch(abi, [fk, fv], [abi.INT64, abi.FLOAT64]))
        be.synchronize()
        return out

    # partial aggregation of a rank's fact slice: HashAgg[GROUP BY key; COUNT(val), SUM(val)](Filter(fact)), any group order
    from sqlrs_amd.expr import Constant as _Constant
    _part_pred = (InputRef(1) > _Constant(args.threshold, abi.FLOAT64)).pack()
    _part_gb, _pk1 = abi.pack_exprs([InputRef(0)])
    _pkeep = []
    _part_aggs = (abi.AggFunc * 2)(_AggFunc("count", InputRef(1), abi.INT64).abi_struct(_pkeep),
                                   _AggFunc("sum", InputRef(1), abi.FLOAT64).abi_struct(_pkeep))
    # merge on the owning rank: HashJoinAgg(dim partition, partials) GROUP BY d.key; SUM(count), SUM(sum) — the Inner
    # join drops the keys without a build partner (joined schema: d.key | p.key, p.count, p.sum)
    _m_lk, _mk1 = abi.pack_exprs([InputRef(0)])
    _m_rk, _mk2 = abi.pack_exprs([InputRef(0)])
    _m_gb, _mk3 = abi.pack_exprs([InputRef(0)])
    _m_rd = (C.c_int32 * 3)(abi.INT64, abi.INT64, abi.FLOAT64)
    _mkeep2 = []
    _m_aggs = (abi.AggFunc * 2)(_AggFunc("sum", InputRef(2), abi.INT64).abi_struct(_mkeep2),
                                _AggFunc("sum", InputRef(3), abi.FLOAT64).abi_struct(_mkeep2))

    def step_combine():
        # partitioned hash join, the group-by's partial aggregation below the exchange (eager aggregation across ranks):
        # 1. dim keys hash-partitioned + exchanged  2. local HashAgg(Filter(fact slice)) by key = partial aggregates
        # 3. partials hash-partitioned + exchanged  4. HashJoinAgg(dim partition, partials) merges and joins them
        t0 = time.perf_counter()
        dk, = exchange([[dim_key]], [abi.INT64], D_shard(n_dim_total, rank, world) * 2 + 1024)
        if xstat["on"]:
            torch.cuda.synchronize()
            be.synchronize()
            xstat["exchange_ms"] += (time.perf_counter() - t0) * 1e3
        a = C.c_void_p()
        be.check(be.fn("hash_agg_create")(be.ctx, 1, _part_gb, 2, _part_aggs, C.byref(a)))
        be.check(be.fn("hash_agg_set_group_order")(a, abi.GROUP_ORDER_ANY))
        be.check(be.fn("hash_agg_set_filter")(a, C.byref(_part_pred.abi)))
        fb = device_batch(abi, [fact_key, fact_val], [abi.INT64, abi.FLOAT64])
        be.check(be.fn("hash_agg_push")(a, fb.ptr))
        po = C.POINTER(abi.Batch)()
        be.check(be.fn("hash_agg_finish")(a, abi.MEM_DEVICE, C.byref(po)))
        be.fn("hash_agg_destroy")(a)
        part = be.wrap(po)
        g = part.num_rows
        cols = [_tensor_view(torch, part.column(i).values, g, t, dev)
                for i, t in enumerate((torch.int64, torch.int64, torch.float64))]
        t0 = time.perf_counter()
        rk, rc, rs = exchange([cols], [abi.INT64, abi.INT64, abi.FLOAT64], g + 1024)
        if xstat["on"]:
            torch.cuda.synchronize()
            be.synchronize()
            xstat["exchange_ms"] += (time.perf_counter() - t0) * 1e3
        part.release()
        ja = C.c_void_p()
        be.check(be.fn("join_agg_create")(be.ctx, 1, _m_lk, _m_rk, 1, 3, _m_rd, 1, _m_gb, 2, _m_aggs, C.byref(ja)))
        # a global first-seen order does not exist across ranks: no ordering pass
        be.check(be.fn("join_agg_set_group_order")(ja, abi.GROUP_ORDER_ANY))
        db = device_batch(abi, [dk], [abi.INT64])
        mb = device_batch(abi, [rk, rc, rs], [abi.INT64, abi.INT64, abi.FLOAT64])
        be.check(be.fn("join_agg_build_push")(ja, db.ptr))
        be.check(be.fn("join_agg_build_finish")(ja))
        be.check(be.fn("join_agg_probe_push")(ja, mb.ptr))
        ao = C.POINTER(abi.Batch)()
        be.check(be.fn("join_agg_finish")(ja, abi.MEM_DEVICE, C.byref(ao)))
        pipe.fused_batches = be.fn("join_agg_fused_batches")(ja)
        be.fn("join_agg_destroy")(ja)
        be.synchronize()
        return be.wrap(ao)

    def step_broadcast():
        # 1. replicate the small build side  2. local Filter -> HashJoinAgg = PARTIAL aggregates
        t0 = time.perf_counter()
        dk = gather_dim()
        if xstat["on"]:
            torch.cuda.synchronize()
            xstat["exchange_ms"] += (time.perf_counter() - t0) * 1e3
        part = pipe.step(device_batch(abi, [dk], [abi.INT64]),
                         device_batch(abi, [fact_key, fact_val], [abi.INT64, abi.FLOAT64]))
        be.synchronize()
        g = part.num_rows
        cols = [_tensor_view(torch, part.column(i).values, g, t, dev)
                for i, t in enumerate((torch.int64, torch.int64, torch.float64))]
        # 3. exchange the partial aggregates by key  4. merge: SUM(count), SUM(sum) per key
        t0 = time.perf_counter()
        rk, rc, rs = exchange([cols], [abi.INT64, abi.INT64, abi.FLOAT64], g + 1024)
        if xstat["on"]:
            torch.cuda.synchronize()
            be.synchronize()
            xstat["exchange_ms"] += (time.perf_counter() - t0) * 1e3
        part.release()
        a = C.c_void_p()
        be.check(be.fn("hash_agg_create")(be.ctx, 1, merge_gb, 2, merge_aggs, C.byref(a)))
        # a global first-seen order does not exist across ranks: no ordering pass
        be.check(be.fn("hash_agg_set_group_order")(a, abi.GROUP_ORDER_ANY))
        mb = device_batch(abi, [rk, rc, rs], [abi.INT64, abi.INT64, abi.FLOAT64])
        be.check(be.fn("hash_agg_push")(a, mb.ptr))
        ao = C.POINTER(abi.Batch)()
        be.check(be.fn("hash_agg_finish")(a, abi.MEM_DEVICE, C.byref(ao)))
        be.fn("hash_agg_destroy")(a)
        be.synchronize()
        return be.wrap(ao)

    def one_step():
        if multi:
            for old in abi_x["keep"]:
                old.release()  # what the previous step received through the ABI exchange
            abi_x["keep"] = []
            xstat["steps"] += 1
            return step_broadcast() if strategy == "broadcast" else step_combine() if strategy == "combine" else step_partition()
        out = pipe.step(device_batch(abi, [dim_key], [abi.INT64]),
                        device_batch(abi, [fact_key, fact_val], [abi.INT64, abi.FLOAT64]))
        be.synchronize()
        return out

    def barrier():
        torch.cuda.synchronize()
        be.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- warmup (also validates the result once)
    out = None
    for _ in range(max(args.warmup, 1)):
        if out is not None:
            out.release()
        out = one_step()
    out_groups_local = out.num_rows
    ok, ngroups, got_rows, msg = check_groups(torch, dist if multi else None, dev, out, exp_cnt, exp_sum, has_dim)
    if rank == 0:
        log(f"[bench] check (per group: keys, COUNT bit-exact, SUM 1e-9 rel): {msg} -> {'OK' if ok else 'MISMATCH'}")
    if not ok:
        raise SystemExit("bench result check failed")
    out.release()

    # ---- timed region: exactly K steps between barriers
    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        one_step().release()
    barrier()
    elapsed = time.perf_counter() - t_start
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if multi:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = el.item()
    ms_per_step = elapsed / args.steps * 1e3
    value = n_fact_total / (elapsed / args.steps) / 1e6
    # (review r05 #5) the same K steps once more, each timed on its own: median / min / max beside the contract's mean
    step_times = None
    if not multi:
        ts = []
        for _ in range(args.steps):
            t_one = time.perf_counter()
            one_step().release()
            be.synchronize()
            ts.append((time.perf_counter() - t_one) * 1e3)
        ts.sort()
        step_times = {"median": round(ts[len(ts) // 2], 3), "min": round(ts[0], 3), "max": round(ts[-1], 3), "steps": len(ts),
                      "note": "K further steps, a stream synchronisation behind each (value / ms_per_step stay the K steps between two barriers)"}

    # ---- per-kernel device time (HIP events on the ctx stream), separate profiled steps
    be.profile(True)
    xstat.update(on=multi, exchange_ms=0.0, bytes_off_rank=0)
    torch.cuda.synchronize()
    t_prof = time.perf_counter()
    # (N = 1: as many profiled steps as timed ones, so that `roofline` rests on the MEAN of K launches of the dominant
    #  kernel — two launches read like the minimum of a rocprofv3 trace; N > 1 keeps two: the exchange statistics are per 2)
    nprof = 2 if multi else max(2, args.steps)
    for _ in range(nprof):
        one_step().release()
    torch.cuda.synchronize()
    be.synchronize()
    prof_step_ms = (time.perf_counter() - t_prof) * 1e3 / nprof
    xstat["on"] = False
    prof = be.profile_read()
    be.profile(False)
    exchange_info = None
    if multi:  # SURVEY.md §8e scaling report: exchange vs local time, bytes over xGMI, rate per link
        x_ms, x_bytes = xstat["exchange_ms"] / 2, xstat["bytes_off_rank"] / 2
        exchange_info = {"strategy": strategy, "chunks": n_chunks if strategy == "partition" else 1,
                         "impl": ("C ABI (exchange.hip, RCCL on the ctx stream, no torch on the data path): sqlrs_exchange_all_to_all for "
                                  "single-chunk exchanges, sqlrs_exchange_begin / send_chunk / finish for the chunked fact rows"
                                  if abi_x["h"] is not None else "torch.distributed"),
                         "fused_filter_partition": bool(fused_exchange and strategy == "partition"),
                         "step_ms_profiled": round(prof_step_ms, 3), "exchange_ms": round(x_ms, 3),
                         "local_ms": round(prof_step_ms - x_ms, 3), "bytes_off_rank_per_step": int(x_bytes),
                         "GBps_per_rank": round(x_bytes / max(x_ms, 1e-9) / 1e6, 1),
                         "GBps_per_link": round(x_bytes / max(x_ms, 1e-9) / 1e6 / max(world - 1, 1), 1),
                         "ranks": rank_info,
                         "note": "rank 0, two profiled steps with a sync behind the exchange phase (filter + partition "
                                 "kernels + collectives, chunk k's all-to-all overlapping chunk k+1's kernels); split "
                                 "sizes travel over a gloo side group; xGMI link peak 153 GB/s"}
        # the other strategies, timed with the same wall clock over fewer steps (never the headline value)
        main_strategy = strategy
        for slot, alt in zip(("alternative", "alternative2"), [x for x in ("partition", "broadcast", "combine") if x != main_strategy]):
            try:
                strategy = alt
                pipe.partial = strategy == "broadcast"
                o = one_step()
                ok_alt, _, _, msg_alt = check_groups(torch, dist, dev, o, exp_cnt, exp_sum, has_dim)
                o.release()
                barrier()
                t_alt = time.perf_counter()
                for _ in range(max(2, args.steps // 4)):
                    one_step().release()
                barrier()
                el_alt = torch.tensor([time.perf_counter() - t_alt], dtype=torch.float64, device=dev)
                dist.all_reduce(el_alt, op=dist.ReduceOp.MAX)
                exchange_info[slot] = {"strategy": alt, "ms_per_step": round(el_alt.item() / max(2, args.steps // 4) * 1e3, 3),
                                       "check": "OK" if ok_alt else msg_alt}
            except Exception as e:  # the alternatives are informational: they must never take the headline line down
                exchange_info[slot] = {"strategy": alt, "error": repr(e)[:300]}
            finally:
                strategy = main_strategy
                pipe.partial = strategy == "broadcast"
    workload = {"fact_rows": f_hi - f_lo, "dim_rows": d_hi - d_lo, "selectivity": expected_kept / max(f_hi - f_lo, 1),
                "matches": expected_kept, "groups": out_groups_local}
    if multi:  # after the exchange every rank holds about 1/N of everything
        workload = {"fact_rows": n_fact_total // world, "dim_rows": n_dim_total // world,
                    "selectivity": got_rows / n_fact_total, "matches": got_rows // world,
                    "groups": ngroups // world}
    roofline = None
    if rank == 0:
        rows = sorted(prof.items(), key=lambda kv: -kv[1][0])
        tot = sum(v[0] for _, v in rows) or 1.0
        log(f"[bench] device time per kernel class ({nprof} profiled steps):")
        for name, (ms, launches) in rows:
            ab = algorithmic_bytes(name, workload)
            extra = f"  {ab / (ms / launches) / 1e6:8.0f} GB/s algorithmic" if (ab and launches) else ""
            log(f"    {name:22s} {ms / nprof:9.3f} ms/step  {launches // nprof:4d} launches/step  {100 * ms / tot:5.1f}%{extra}")
        for name, (ms, launches) in rows:
            ab = algorithmic_bytes(name, workload)
            if ab and launches:
                per_launch_ms = ms / launches
                ach = ab / per_launch_ms / 1e6
                traffic, traffic_src = pmc_traffic(name, n_fact_total, n_dim_total, world, args)
                # SURVEY.md §8d operator-level figure: 16 B per probe row + 16 B per build row read, 24 B per
                # group written, over the WHOLE step (temporaries excluded) — per GPU
                pipe_bytes = (16 * n_fact_total + 16 * n_dim_total) // world + 24 * int(ngroups) // world
                pipe_gbps = pipe_bytes / ms_per_step / 1e6
                sb = survey_bytes(name, workload) or ab
                s_ach = sb / per_launch_ms / 1e6
                prof_avg_ms, prof_src = profile_average_ms(name, n_fact_total, n_dim_total, world, args)
                roofline = {"kernel": name, "bound": "hbm", "achieved": round(s_ach, 1), "peak": HBM_PEAK_GBPS,
                            "unit": "GB/s", "frac": round(s_ach / HBM_PEAK_GBPS, 4),
                            "traffic": traffic, "traffic_source": traffic_src,
                            "ms_per_launch": round(per_launch_ms, 4), "launches_averaged": int(launches),
                            # the same fraction from the newest COMMITTED rocprofv3 kernel trace of this command (its average
                            # duration for the kernel): another box, another day — the two must agree to a few percent
                            "frac_profile": (round(sb / prof_avg_ms / 1e6 / HBM_PEAK_GBPS, 4) if prof_avg_ms else None),
                            "ms_per_launch_profile": prof_avg_ms, "profile_source": prof_src,
                            "algorithmic_bytes": int(sb),
                            "kernel_own_bytes": int(ab), "kernel_own_GBps": round(ach, 1),
                            "kernel_own_frac": round(ach / HBM_PEAK_GBPS, 4),
                            "pipeline_bytes": int(pipe_bytes), "pipeline_GBps": round(pipe_gbps, 1),
                            "pipeline_frac": round(pipe_gbps / HBM_PEAK_GBPS, 4),
                            "note": "achieved/frac: SURVEY 8d bytes of the dominant kernel's launch (16 B per probe row it "
                                    "processes, temporaries excluded) / its HIP-event time per launch; kernel_own_*: the same "
                                    "launch priced with the intermediate it writes (its real traffic model); pipeline_*: "
                                    "SURVEY 8d operator bytes (16 nP + 16 nB + 24 G per GPU) / ms_per_step"}
                break

    if args.child:  # a child of --processes: one number, nothing else
        print(json.dumps({"ms_per_step": round(ms_per_step, 3)}), flush=True)
        return
    processes = None
    if rank == 0 and not multi and args.processes > 1:
        processes = measure_in_fresh_processes(args, ms_per_step)

    variants = None
    if rank == 0 and not multi and not args.no_operators and not args.unfused:
        variants = bench_variants(be, abi, torch, dev, args, fact_key, fact_val, dim_key, n_dim_total,
                                  exp_cnt, exp_sum, has_dim, ms_headline=ms_per_step)

    operators = None
    if rank == 0 and not args.no_operators and not multi:
        del fact_key, fact_val, dim_key
        be.fn("ctx_pool_trim")(be.ctx)
        torch.cuda.empty_cache()
        operators = bench_operators(be, abi, datagen, torch, dev)

    cpu = None
    if rank == 0 and not args.no_cpu_baseline and not multi:  # (the CPU leg is an N = 1 figure)
        cpu = cpu_baseline(args, abi, datagen, n_dim_total)

    if rank == 0:
        line = {
            "metric": "Mrows/sec for filter->hash-join->group-by pipeline; achieved HBM GB/s vs peak",
            "value": round(value, 1), "unit": "Mrows/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "int64 keys / f64 sum", "data": "synthetic",
            "config": {"workload": "C5 filter(val>0.5) -> hash-join(fact x dim on int64 key) -> group-by(key) COUNT,SUM(f64)",
                       "fact_rows": n_fact_total, "dim_rows": n_dim_total, "groups": int(ngroups),
                       "selectivity": round(got_rows / n_fact_total, 4),
                       "operators": "Filter -> HashJoin -> HashAgg (3 operators)" if args.unfused else
                       "Filter -> HashJoinAgg (HashAgg fused over the Inner HashJoin)" if args.separate_filter else
                       "HashJoinAgg with the probe-side Filter handed to it (sqlrs_join_agg_set_probe_filter): "
                       f"filter evaluated inside the first partition pass = {bool(pipe.filter_fused_batches)}",
                       "parallelism": ("single GPU" if not multi else
                                       f"x{world}: all-gather dim, local partial aggregation, all-to-all of partial aggregates, merge"
                                       if strategy == "broadcast" else
                                       f"x{world}: partitioned hash join, partial aggregation below the exchange — dim hash-partitioned on the "
                                       "join key + RCCL all-to-all; HashAgg(Filter(fact slice)) by key on every rank; partial aggregates "
                                       "hash-partitioned + RCCL all-to-all; HashJoinAgg(dim partition, partials) on the owning rank"
                                       if strategy == "combine" else
                                       f"x{world}: partitioned hash join — Filter below the exchange, fact + dim hash-partitioned on "
                                       f"the join key, RCCL all-to-all in {n_chunks} overlapped chunks, local HashJoinAgg"
                                       + ("; Filter + partition fused in one pass (sqlrs_hash_partition_filter)" if fused_exchange else ""))},
            "roofline": roofline, "cpu_baseline": cpu,
            # probe batches of the last step that took the library's fused join + aggregate route / had the Filter
            # evaluated inside the first partition pass (0 = the operators were composed: the 2.5x slower form)
            "fused_batches": int(pipe.fused_batches), "filter_fused_batches": int(pipe.filter_fused_batches),
        }
        if step_times:
            line["ms_per_step_median"] = step_times["median"]
            line["ms_per_step_each"] = step_times
        if processes:
            line["ms_per_step_processes"] = processes
        if operators:
            line["operators"] = operators
        if variants:
            line["c5_variants"] = variants
        if exchange_info:
            line["exchange"] = exchange_info
        if json_fd is not None:
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(line) + "\n").encode())
        else:
            print(json.dumps(line), flush=True)
    if multi:
        if abi_x["h"] is not None:
            for old in abi_x["keep"]:
                old.release()
            be.fn("exchange_destroy")(abi_x["h"])
        dist.destroy_process_group()


def measure_in_fresh_processes(args, own_ms):
    """ms_per_step of the same command in fresh processes (this one's timed region + args.processes - 1 children, run one
    after the other while this process sits idle): kernel times repeat to 0.2 % inside a process and differ by 5-15 %
    between processes (identical VIRTUAL addresses re-allocated give different times, tools/placement_log.py: it follows
    the physical backing of the buffers), so one process's number is a draw; the line reports min / median / max."""
    import subprocess
    vals = [round(own_ms, 3)]
    cmd = [sys.executable, os.path.abspath(__file__), "--child", "--no-operators", "--no-cpu-baseline", "--processes", "1",
           "--steps", str(args.steps), "--warmup", str(args.warmup), "--rows", str(args.rows), "--dim-rows", str(args.dim_rows),
           "--threshold", str(args.threshold)]
    cmd += (["--unfused"] if args.unfused else []) + (["--separate-filter"] if args.separate_filter else [])
    for _ in range(args.processes - 1):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            vals.append(float(json.loads(r.stdout.strip().splitlines()[-1])["ms_per_step"]))
        except Exception as e:  # informational: never takes the headline line down
            log(f"[bench] a --processes child failed: {e!r}")
    sv = sorted(vals)
    return {"n": len(vals), "min": sv[0], "median": sv[len(sv) // 2], "max": sv[-1], "values": vals,
            "note": "ms_per_step of the same command in fresh processes (values[0] = this process = `ms_per_step`)"}


def pmc_traffic(kernel, n_fact, n_dim, world, args):
    """(HBM bytes per launch of `kernel`, where that number comes from).  The counters are NOT collected
    in this run: they come from the newest committed profiles/*_pmc_traffic.json, written by
    tools/profile_round.sh (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this very command,
    HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024).  Only valid for the default workload; otherwise null."""
    if not (n_fact == 1_000_000_000 and n_dim == 10_000_000 and world == 1 and args.threshold == 0.5
            and not args.unfused and not args.force_exchange):
        return None, None
    try:
        import glob
        newest = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")) if "_ops_" not in f)[-1]
        with open(newest) as f:
            v = json.load(f)["kernels"][kernel]["hbm_bytes_per_launch"]
        return v, (f"profiles/{os.path.basename(newest)} (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                   "this command on another box; NOT measured in this run)")
    except (OSError, KeyError, ValueError, IndexError):
        return None, None


def ops_traffic(substrings):
    """HBM bytes per launch of the named kernels (counter traffic, summed) from the newest committed
    profiles/*_ops_pmc_traffic.json (tools/profile_round.sh: rocprofv3 --pmc passes of `--operators`), and that file's name."""
    try:
        import glob
        newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_ops_pmc_traffic.json")))[-1]
        with open(newest) as f:
            ks = json.load(f)["kernels"]
        tot, seen = 0, []
        for sub in substrings:
            hit = [(k, v) for k, v in ks.items() if sub in k]
            if not hit:
                return None, None
            k, v = max(hit, key=lambda kv: kv[1]["hbm_bytes_per_launch"])
            tot += int(v["hbm_bytes_per_launch"])
            seen.append(k.split("(")[0][-48:])
        return tot, f"profiles/{os.path.basename(newest)}: " + " + ".join(seen) + " (committed counter passes; NOT measured in this run)"
    except (OSError, KeyError, ValueError, IndexError):
        return None, None


# kernel classes of the profile (ProfScope labels) -> the kernel symbol a rocprofv3 trace lists
PROFILE_SYMBOL = {"rp_chunk_scatter_filter": "rp_chunk_scatter_", "rp_chunk_scatter": "rp_chunk_scatter_", "rp_scatter": "rp_scatter_",
                  "lds_agg": "lds_agg_", "filter_cmp_const": "filter_cmp_const"}


def profile_average_ms(kernel, n_fact, n_dim, world, args):
    """(average duration in ms of the dominant kernel in the newest committed profiles/*_kernel_stats.csv, that file's name):
    the rocprofv3 --kernel-trace --stats summary of this very command, written by tools/profile_round.sh.  Default workload
    only; the row with the largest total duration among the kernels whose symbol carries the class's name."""
    if not (n_fact == 1_000_000_000 and n_dim == 10_000_000 and world == 1 and args.threshold == 0.5
            and not args.unfused and not args.force_exchange) or kernel not in PROFILE_SYMBOL:
        return None, None
    try:
        import csv
        import glob
        newest = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_kernel_stats.csv")) if "_ops_" not in f)[-1]
        best = None
        with open(newest, newline="") as f:
            for r in csv.DictReader(f):
                if PROFILE_SYMBOL[kernel] in r["Name"].split("(")[0] and (best is None or float(r["TotalDurationNs"]) > best[0]):
                    best = (float(r["TotalDurationNs"]), float(r["AverageNs"]))
        if best is None:
            return None, None
        return round(best[1] / 1e6, 4), f"profiles/{os.path.basename(newest)} (AverageNs of the kernel's launches; another run of this command)"
    except (OSError, KeyError, ValueError, IndexError):
        return None, None


def bench_variants(be, abi, torch, dev, args, fact_key, fact_val, dim_key, n_dim, exp_cnt, exp_sum, has_dim, ms_headline=None):
    """The headline query off its three data-dependent specialisations, timed by the same wall clock
    (barrier-free, N = 1) and checked per group like the headline:
      * `three_operators`: Filter, HashJoin, HashAgg as separate operators (joined batch materialised);
      * `sparse_keys`: the same rows with keys k -> k * A + B (A odd: a bijection of int64, the keys no
        longer cover a dense range), so the join builds its general hash table / hashed LDS buckets and
        the group-by partitions by hash instead of by key range."""
    res = {}

    def timed(pipe, dim_b, fact_b, steps, warm):
        # (the variants run after a pool trim with few steps: every step is timed on its own and the MEDIAN reported — a mean of
        #  three let one step that met a slow re-allocation (profiles/r05zz_placement_draws.txt: up to seconds) read as +30 %,
        #  profiles/r05zzzzz_bench_default.json `sorted_fact`; the headline keeps the contract's K steps between two syncs)
        for _ in range(warm):
            pipe.step(dim_b(), fact_b()).release()
        be.synchronize()
        ts = []
        for _ in range(max(steps, 3)):
            t = time.perf_counter()
            pipe.step(dim_b(), fact_b()).release()
            be.synchronize()
            ts.append((time.perf_counter() - t) * 1e3)
        ts.sort()
        return ts[len(ts) // 2]

    def batches(dk, fk, fv):
        return (lambda: device_batch(abi, [dk], [abi.INT64])), (lambda: device_batch(abi, [fk, fv], [abi.INT64, abi.FLOAT64]))

    n = fact_key.numel()
    # ---- three separate operators
    pipe = Pipeline(be, abi, args.threshold, fused=False)
    db, fb = batches(dim_key, fact_key, fact_val)
    out = pipe.step(db(), fb())
    be.synchronize()
    ok, groups, rows, msg = check_groups(torch, None, dev, out, exp_cnt, exp_sum, has_dim)
    out.release()
    ms = timed(pipe, db, fb, 3, 1)
    res["three_operators"] = {"ms_per_step": round(ms, 3), "Mrows_s": round(n / ms / 1e3, 1), "check": "OK" if ok else msg}
    be.fn("ctx_pool_trim")(be.ctx)
    # ---- duplicate build keys: half of the dim keys appear twice (1.5e7 build rows, a many-to-many join grouped by its key)
    dim2 = torch.cat([dim_key, dim_key[: n_dim // 2]])
    has2 = torch.bincount(dim2, minlength=n_dim)
    torch.cuda.synchronize()
    pipe = Pipeline(be, abi, args.threshold, fused=True)
    db, fb = batches(dim2, fact_key, fact_val)
    out = pipe.step(db(), fb())
    be.synchronize()
    ok, groups, rows, msg = check_groups(torch, None, dev, out, exp_cnt, exp_sum, has2)
    out.release()
    ms = timed(pipe, db, fb, 3, 1)
    res["duplicate_build_keys"] = {"ms_per_step": round(ms, 3), "Mrows_s": round(n / ms / 1e3, 1), "build_rows": int(dim2.numel()),
                                   "joined_rows": rows, "fused_route": bool(pipe.fused_batches), "check": "OK" if ok else msg}
    del dim2, has2
    be.fn("ctx_pool_trim")(be.ctx)
    # ---- GROUP BY a dim ATTRIBUTE: SELECT d.region, COUNT(f.val), SUM(f.val) ... GROUP BY d.region (region = key mod 1000)
    res["group_by_dim_attribute"] = bench_group_by_attribute(be, abi, torch, dev, args, fact_key, fact_val, dim_key, n_dim,
                                                             exp_cnt, exp_sum, has_dim)
    be.fn("ctx_pool_trim")(be.ctx)
    # ---- adversarial inputs for the same fused route (review r03 #9: what the data-dependent parts cost on the inputs they
    #      like least; `ms_fast_route` = the headline's ms on its uniformly random keys):
    #      * sorted_fact: the fact rows SORTED by key — every level-1 tile lands in one digit, every level-2 tile in one
    #        bucket, a workgroup's chunks of one digit fill back to back (early closes), first rows in key order;
    #      * hot_key: 30 % of the fact rows carry ONE key — one bucket holds a third of the kept rows, its accumulator slot takes
    #        them all.  Same per-group check as the headline (expectation adjusted for the moved rows).
    adv = {}
    order = torch.sort(fact_key, stable=False).indices
    fk_s, fv_s = fact_key[order], fact_val[order]
    del order
    torch.cuda.synchronize()
    pipe = Pipeline(be, abi, args.threshold, fused=True)
    db, fb = batches(dim_key, fk_s, fv_s)
    out = pipe.step(db(), fb())
    be.synchronize()
    ok, groups, rows, msg = check_groups(torch, None, dev, out, exp_cnt, exp_sum, has_dim)
    out.release()
    ms = timed(pipe, db, fb, 3, 1)
    adv["sorted_fact"] = {"ms_per_step": round(ms, 3), "fused_route": bool(pipe.fused_batches), "check": "OK" if ok else msg}
    del fk_s, fv_s
    be.fn("ctx_pool_trim")(be.ctx)
    torch.cuda.empty_cache()
    k0 = int(dim_key[n_dim // 3].item())
    moved = (torch.arange(n, device=dev, dtype=torch.int64) * 0x9E3779B1 % 10) < 3
    mp = moved & (fact_val > args.threshold)
    e_cnt, e_sum = exp_cnt.clone(), exp_sum.clone()
    e_cnt.index_add_(0, fact_key[mp], torch.full((int(mp.sum().item()),), -1, dtype=torch.int64, device=dev))
    e_sum.index_add_(0, fact_key[mp], -fact_val[mp])
    e_cnt[k0] += int(mp.sum().item())
    e_sum[k0] += fact_val[mp].sum()
    # (a sum rebuilt by subtraction carries the rounding of both sums: groups that lost all their rows are exactly zero)
    e_sum = torch.where(e_cnt == 0, torch.zeros_like(e_sum), e_sum)
    fk_h = torch.where(moved, torch.full_like(fact_key, k0), fact_key)
    del moved, mp
    torch.cuda.synchronize()
    pipe = Pipeline(be, abi, args.threshold, fused=True)
    db, fb = batches(dim_key, fk_h, fact_val)
    out = pipe.step(db(), fb())
    be.synchronize()
    ok, groups, rows, msg = check_groups(torch, None, dev, out, e_cnt, e_sum, has_dim)
    out.release()
    ms = timed(pipe, db, fb, 3, 1)
    adv["hot_key"] = {"ms_per_step": round(ms, 3), "fused_route": bool(pipe.fused_batches), "check": "OK" if ok else msg}
    del fk_h, e_cnt, e_sum
    be.fn("ctx_pool_trim")(be.ctx)
    torch.cuda.empty_cache()
    worst = max(v["ms_per_step"] for v in adv.values())
    res["adversarial"] = {"ms_per_step": worst, "ms_fast_route": None if ms_headline is None else round(ms_headline, 3),
                          "ratio": None if not ms_headline else round(worst / ms_headline, 3), "shapes": adv,
                          "check": "OK" if all(v["check"] == "OK" for v in adv.values()) else "; ".join(v["check"] for v in adv.values())}
    # ---- sparse keys: k -> k * A + B in place (wrapping int64 arithmetic), inverse for the check
    A, B = 0x9E3779B97F4A7C15, 0x632BE59BD9B4E019
    A_s = A - (1 << 64)
    A_inv = pow(A, -1, 1 << 64)
    A_inv_s = A_inv - (1 << 64) if A_inv >= (1 << 63) else A_inv
    B_s = B - (1 << 64) if B >= (1 << 63) else B
    for t in (fact_key, dim_key):
        for lo in range(0, t.numel(), 1 << 27):
            t[lo:lo + (1 << 27)].mul_(A_s).add_(B_s)
    torch.cuda.synchronize()
    key_of = lambda k: (k - B_s) * A_inv_s
    try:
        pipe = Pipeline(be, abi, args.threshold, fused=True)
        db, fb = batches(dim_key, fact_key, fact_val)
        out = pipe.step(db(), fb())
        be.synchronize()
        ok, groups, rows, msg = check_groups(torch, None, dev, out, exp_cnt, exp_sum, has_dim, key_of=key_of)
        out.release()
        ms = timed(pipe, db, fb, 3, 1)
        res["sparse_keys"] = {"ms_per_step": round(ms, 3), "Mrows_s": round(n / ms / 1e3, 1),
                              "fused_route": bool(pipe.fused_batches), "check": "OK" if ok else msg}
    finally:
        for t in (fact_key, dim_key):  # restore the dense keys
            for lo in range(0, t.numel(), 1 << 27):
                t[lo:lo + (1 << 27)].sub_(B_s).mul_(A_inv_s)
        torch.cuda.synchronize()
    be.fn("ctx_pool_trim")(be.ctx)
    for k_, v_ in res.items():
        log(f"[bench] C5 variant {k_}: {v_}")
    if any(v["check"] != "OK" for v in res.values()):
        raise SystemExit("bench variant result check failed")
    return res


def bench_group_by_attribute(be, abi, torch, dev, args, fact_key, fact_val, dim_key, n_dim, exp_cnt, exp_sum, has_dim, regions=1000):
    """The headline query grouped by a column of the DIM side instead of the join key — the usual star-join shape:
    HashAgg[group_by = d.region](HashJoin(dim[key, region], Filter(fact))) through sqlrs_join_agg_*.  The operator
    groups by join key first and re-aggregates the per-key rows by region (eager aggregation, include/sqlrs_hip.h:
    sqlrs_join_agg_eager_groups); `ms_composed` is the same plan with that route switched off (SQLRS_EAGER_AGG=0,
    read at operator creation): the joined batch is materialised and aggregated.  Checked per region against torch."""
    from sqlrs_amd.expr import AggFunc, Constant, InputRef
    region = dim_key % regions
    lk, _k1 = abi.pack_exprs([InputRef(0)])
    rk, _k2 = abi.pack_exprs([InputRef(0)])
    gb, _k3 = abi.pack_exprs([InputRef(1)])  # d.region
    keep = []
    aggs = (abi.AggFunc * 2)(AggFunc("count", InputRef(3), abi.INT64).abi_struct(keep),
                             AggFunc("sum", InputRef(3), abi.FLOAT64).abi_struct(keep))
    right_dtypes = (C.c_int32 * 2)(abi.INT64, abi.FLOAT64)
    pred = (InputRef(1) > Constant(args.threshold, abi.FLOAT64)).pack()
    # expectation per region from the per-key expectation
    joined = (has_dim > 0)
    e_cnt = torch.zeros(regions, dtype=torch.int64, device=dev).index_add_(0, torch.arange(n_dim, device=dev) % regions, exp_cnt * joined)
    e_sum = torch.zeros(regions, dtype=torch.float64, device=dev).index_add_(0, torch.arange(n_dim, device=dev) % regions,
                                                                              exp_sum * joined.to(torch.float64))
    torch.cuda.synchronize()  # (`region` was produced on torch's stream; the library reads it on its own)

    def step():
        ja = C.c_void_p()
        be.check(be.fn("join_agg_create")(be.ctx, 1, lk, rk, 2, 2, right_dtypes, 1, gb, 2, aggs, C.byref(ja)))
        be.check(be.fn("join_agg_set_probe_filter")(ja, C.byref(pred.abi)))
        db = device_batch(abi, [dim_key, region], [abi.INT64, abi.INT64])
        fb = device_batch(abi, [fact_key, fact_val], [abi.INT64, abi.FLOAT64])
        be.check(be.fn("join_agg_build_push")(ja, db.ptr))
        be.check(be.fn("join_agg_build_finish")(ja))
        be.check(be.fn("join_agg_probe_push")(ja, fb.ptr))
        ao = C.POINTER(abi.Batch)()
        be.check(be.fn("join_agg_finish")(ja, abi.MEM_DEVICE, C.byref(ao)))
        eg = be.fn("join_agg_eager_groups")(ja)
        be.fn("join_agg_destroy")(ja)
        be.synchronize()
        return be.wrap(ao), eg

    def check(out):
        g = out.num_rows
        r = _tensor_view(torch, out.column(0).values, g, torch.int64, dev)
        c = _tensor_view(torch, out.column(1).values, g, torch.int64, dev)
        sm = _tensor_view(torch, out.column(2).values, g, torch.float64, dev)
        ok = g == int((e_cnt > 0).sum().item()) and bool((torch.bincount(r, minlength=regions) <= 1).all().item())
        ok = ok and bool((c == e_cnt[r]).all().item()) and bool(((sm - e_sum[r]).abs() <= 1e-9 * e_sum[r].abs().clamp_min(1e-300)).all().item())
        return ok, g

    def timed(steps=3):
        step()[0].release()
        be.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            step()[0].release()
        be.synchronize()
        return (time.perf_counter() - t) / steps * 1e3

    out, eg = step()
    ok, groups = check(out)
    out.release()
    ms = timed()
    old = os.environ.get("SQLRS_EAGER_AGG")
    os.environ["SQLRS_EAGER_AGG"] = "0"
    try:
        out, _ = step()
        ok2, _g = check(out)
        out.release()
        ms_c = timed(2)
    finally:
        if old is None:
            del os.environ["SQLRS_EAGER_AGG"]
        else:
            os.environ["SQLRS_EAGER_AGG"] = old
    n = fact_key.numel()
    return {"ms_per_step": round(ms, 3), "Mrows_s": round(n / ms / 1e3, 1), "groups": groups, "eager_groups": int(eg),
            "ms_composed": round(ms_c, 3), "check": "OK" if (ok and ok2) else f"mismatch (eager {ok}, composed {ok2})"}


def bench_operators(be, abi, datagen, torch, dev, reps=3):
    """Configs C2 / C3 / C4 of BASELINE.json, one operator at a time, HBM-resident inputs,
    timed with HIP events on the ctx stream.  GB/s = algorithmic bytes of SURVEY.md §8d / time."""
    from sqlrs_amd.expr import AggFunc, Constant, InputRef

    def profile_of(fn, label):
        be.profile(True)
        fn()
        pr = be.profile_read()
        be.profile(False)
        rows = sorted(pr.items(), key=lambda kv: -kv[1][0])
        log(f"[bench]   {label} kernel classes: " + ", ".join(f"{k} {v[0]:.3f}ms x{v[1]}" for k, v in rows if v[0] > 0.01))

    def timed(fn):
        fn()  # warm-up (also warms the memory pool)
        t = C.c_void_p()
        be.check(be.fn("timer_create")(be.ctx, C.byref(t)))
        best = 1e30
        for _ in range(reps):
            be.check(be.fn("timer_start")(t))
            fn()
            be.check(be.fn("timer_stop")(t))
            ms = C.c_double()
            be.check(be.fn("timer_elapsed_ms")(t, C.byref(ms)))
            best = min(best, ms.value)
        be.fn("timer_destroy")(t)
        return best

    res = {}
    D = abi.MEM_DEVICE
    # ---- C2: SELECT v1 FROM t WHERE v1 > k, 1e8 int64 rows, selectivity 0.5
    n = 100_000_000
    v1 = datagen.fill_chunks(torch.empty(n, dtype=torch.int64, device=dev),
                             lambda i: datagen._lsr(datagen.splitmix64_t(0xC2, i), 33))  # mod 2^31
    torch.cuda.synchronize()  # inputs are produced on torch's stream, consumed on the ctx stream
    for sel, k in ((0.5, 1 << 30), (0.01, int((1 << 31) * 0.99)), (0.99, int((1 << 31) * 0.01))):
        e = (InputRef(0) > Constant(k, abi.INT64)).pack()
        b = device_batch(abi, [v1], [abi.INT64])
        kept = [0]

        def run():
            f = C.c_void_p()
            be.check(be.fn("filter_create")(be.ctx, C.byref(e.abi), C.byref(f)))
            o = C.POINTER(abi.Batch)()
            be.check(be.fn("filter_push")(f, b.ptr, D, C.byref(o)))
            kept[0] = o.contents.num_rows
            be.fn("batch_release")(o)
            be.fn("filter_destroy")(f)
        ms = timed(run)
        by = 8 * n + 8 * kept[0]
        res[f"C2_filter_s{sel}"] = {"rows": n, "kept": kept[0], "ms": round(ms, 3), "Mrows_s": round(n / ms / 1e3, 1),
                                    "GBps": round(by / ms / 1e6, 1), "frac": round(by / ms / 1e6 / HBM_PEAK_GBPS, 4)}
    del v1
    # ---- C3: 1e8 fact JOIN 1e6 dim on int64 key (Inner, index-pair output)
    nP, nB = 100_000_000, 1_000_000
    dim_key = datagen.fill_chunks(torch.empty(nB, dtype=torch.int64, device=dev), lambda i: datagen.dim_key_t(i, nB))
    A_s = 0x9E3779B97F4A7C15 - (1 << 64)  # odd multiplier: sparse 64-bit keys, same join result
    for hit, mod, sparse in (("all_hit", nB, False), ("half_hit", 2 * nB, False), ("sparse_keys_all_hit", nB, True)):
        fk = datagen.fill_chunks(torch.empty(nP, dtype=torch.int64, device=dev), lambda i: datagen.key_t(0xF1, i, mod))
        dk = dim_key
        if sparse:  # general hash table instead of the direct-address table
            fk.mul_(A_s).add_(12345)
            dk = dim_key * A_s + 12345
        torch.cuda.synchronize()
        db, fb = device_batch(abi, [dk], [abi.INT64]), device_batch(abi, [fk], [abi.INT64])
        lk, _k1 = abi.pack_exprs([InputRef(0)])
        rk, _k2 = abi.pack_exprs([InputRef(0)])
        rd = (C.c_int32 * 1)(abi.INT64)
        m = [0]
        j = C.c_void_p()

        def build():
            be.check(be.fn("hash_join_create")(be.ctx, abi.JOIN_INNER, 1, lk, rk, None, 1, rd, C.byref(j)))
            be.check(be.fn("hash_join_build_push")(j, db.ptr))
            be.check(be.fn("hash_join_build_finish")(j))

        def probe():
            o = C.POINTER(abi.Batch)()
            be.check(be.fn("hash_join_probe_indices")(j, fb.ptr, D, C.byref(o)))
            m[0] = o.contents.num_rows
            be.fn("batch_release")(o)

        def both():
            build()
            probe()
            be.fn("hash_join_destroy")(j)
        ms_all = timed(both)
        build()
        ms_probe = timed(probe)
        profile_of(probe, f"C3 probe {hit}")
        be.fn("hash_join_destroy")(j)
        by = 8 * nB + 8 * nP + 12 * m[0]
        res[f"C3_join_{hit}"] = {"probe_rows": nP, "build_rows": nB, "pairs": m[0], "ms_build_probe": round(ms_all, 3),
                                 "ms_probe": round(ms_probe, 3), "probe_Mrows_s": round(nP / ms_probe / 1e3, 1),
                                 "GBps": round(by / ms_all / 1e6, 1), "frac": round(by / ms_all / 1e6 / HBM_PEAK_GBPS, 4),
                                 # what bounds the probe: one random L2 lookup per row, not the HBM stream
                                 "bound": ("one random L2 lookup per probe row: key stream + lookups + pair stores are ADDITIVE on "
                                           "the CU's vector memory path (profiles/r02_probe_pmc_ta.txt); the lookups read a BIT-PACKED "
                                           "table (20 bits per possible key for 1e6 build rows: 2.4 MiB, inside one XCD's 4 MiB L2 next to "
                                           "the streams) — the composite microbenchmark with nothing but that memory work takes 0.520 ms per "
                                           "1e8 rows on it = ceiling_frac (0.654 with 4-byte entries; profiles/r05b_ubench2.txt)" if not sparse else
                                           "general keys: blocked partitioned join on LDS tables (per-range partition -> per-bucket LDS "
                                           "probe -> per-range LDS un-permute + compaction); each of the three passes is latency- not "
                                           "bandwidth-bound (5.2 GB moved for 2.0 GB algorithmic)"),
                                 "ceiling_frac": (round(by / 0.520 / 1e6 / HBM_PEAK_GBPS, 4) if not sparse and hit == "all_hit" else None),
                                 "ms_probe_composite_ubench": (0.520 if not sparse and hit == "all_hit" else None)}
        if sparse:  # (review r05 #5) the general-key leg carries its counter traffic against the algorithmic bytes
            tb, tsrc = ops_traffic(["lds_join_partition_kernel", "lds_join_probe_kernel", "lds_join_restore_kernel"])
            res[f"C3_join_{hit}"].update({"algorithmic_bytes": int(by), "traffic": tb, "traffic_ratio": (round(tb / by, 2) if tb else None),
                                          "traffic_source": tsrc})
        del fk
    # ---- general-key join shapes at the C3 size (review r04 #5; not BASELINE configs, no roofline claim): every build key FOUR times
    #      (hash_join.rs:172-177 insertion-order chains, :225-234 probe-major pairs — 4e8 pairs out of 1e8 probe rows), and a
    #      TWO-column key (a.x = b.x AND a.y = b.y: matched by the combined row hash like the reference, hash_utils.rs).
    #      `check` (torch, on the device): pair count = sum of the build multiplicities of the probe keys, pairs probe-row
    #      major, build rows of one probe row ascending (= insertion order), every pair's build key equal to its probe key.
    def join_shape(label, build_cols, probe_cols, nkeys, expect_pairs, key_of_build, key_of_probe):
        db = device_batch(abi, build_cols, [abi.INT64] * len(build_cols))
        fb = device_batch(abi, probe_cols, [abi.INT64] * len(probe_cols))
        lk, _k1 = abi.pack_exprs([InputRef(i) for i in range(nkeys)])
        rk, _k2 = abi.pack_exprs([InputRef(i) for i in range(nkeys)])
        rd = (C.c_int32 * len(probe_cols))(*([abi.INT64] * len(probe_cols)))
        j = C.c_void_p()
        m, chk = [0], ["not run"]

        def both(check=False):
            be.check(be.fn("hash_join_create")(be.ctx, abi.JOIN_INNER, nkeys, lk, rk, None, len(probe_cols), rd, C.byref(j)))
            be.check(be.fn("hash_join_build_push")(j, db.ptr))
            be.check(be.fn("hash_join_build_finish")(j))
            o = C.POINTER(abi.Batch)()
            be.check(be.fn("hash_join_probe_indices")(j, fb.ptr, D, C.byref(o)))
            m[0] = o.contents.num_rows
            if check:
                be.synchronize()
                li = _tensor_view(torch, o.contents.columns[0].values, m[0], torch.int64, dev)
                ri = _tensor_view(torch, o.contents.columns[1].values, m[0], torch.int32, dev).to(torch.int64)
                ok = m[0] == expect_pairs
                ok = ok and bool((ri[1:] >= ri[:-1]).all())                                   # probe-row major
                ok = ok and bool(((ri[1:] != ri[:-1]) | (li[1:] > li[:-1])).all())            # insertion order inside a probe row
                ok = ok and bool((key_of_build(li) == key_of_probe(ri)).all())                # every pair joins equal keys
                chk[0] = "OK" if ok else "MISMATCH"
                del li, ri
            be.fn("batch_release")(o)
            be.fn("hash_join_destroy")(j)
        ms = timed(both)
        both(check=True)
        profile_of(both, label)
        res[label] = {"probe_rows": nP, "build_rows": nB, "pairs": m[0], "ms_build_probe": round(ms, 3),
                      "probe_Mrows_s": round(nP / ms / 1e3, 1), "pairs_Mrows_s": round(m[0] / ms / 1e3, 1), "check": chk[0]}

    gq = torch.Generator(device=dev).manual_seed(44)
    dupk = torch.randint(0, nB // 4, (nB,), dtype=torch.int64, device=dev, generator=gq)
    pk4 = datagen.fill_chunks(torch.empty(nP, dtype=torch.int64, device=dev), lambda i: datagen.key_t(0xF1, i, nB // 4))
    mult = torch.bincount(dupk, minlength=nB // 4)
    expect = int(mult[pk4].sum().item())
    torch.cuda.synchronize()
    join_shape("C3_join_dup_build_keys_x4", [dupk], [pk4], 1, expect, lambda li: dupk[li], lambda ri: pk4[ri])
    del dupk, mult
    pk = datagen.fill_chunks(torch.empty(nP, dtype=torch.int64, device=dev), lambda i: datagen.key_t(0xF1, i, nB))
    # (the second column's values lie outside the first one's: the reference combines the column hashes symmetrically —
    #  hash_utils.rs, (x, y) and (y, x) are ONE key to it — and with both columns over 0..999 every probe row finds two partners)
    bx, by_ = dim_key % 1000, dim_key // 1000 + 10_000
    px, py = pk % 1000, pk // 1000 + 10_000
    torch.cuda.synchronize()
    join_shape("C3_join_two_column_keys", [bx, by_], [px, py], 2, nP, lambda li: bx[li] + 1000 * by_[li], lambda ri: px[ri] + 1000 * py[ri])
    del pk, pk4, bx, by_, px, py
    # (every BASELINE config starts from an EMPTY pool, like a fresh process: what the earlier legs left cached decides where the
    #  next leg's regions land physically, and the scatter kernels' time follows that — profiles/r06e_placement.txt)
    if os.environ.get("SQLRS_BENCH_TRIM_LEGS", "1") != "0":
        be.synchronize()
        be.fn("ctx_pool_trim")(be.ctx)
        torch.cuda.empty_cache()
    # ---- C4: 2e8 rows, 1e6 int64 groups, COUNT(val), SUM(val) f64
    n, G = 200_000_000, 1_000_000
    key = datagen.fill_chunks(torch.empty(n, dtype=torch.int64, device=dev), lambda i: datagen.key_t(0xA1, i, G))
    val = datagen.fill_chunks(torch.empty(n, dtype=torch.float64, device=dev), lambda i: datagen.val_t(0xF2, i))
    torch.cuda.synchronize()
    b = device_batch(abi, [key, val], [abi.INT64, abi.FLOAT64])
    gb, _k3 = abi.pack_exprs([InputRef(0)])
    keep = []
    aggs = (abi.AggFunc * 2)(AggFunc("count", InputRef(1), abi.INT64).abi_struct(keep),
                             AggFunc("sum", InputRef(1), abi.FLOAT64).abi_struct(keep))
    groups = [0]

    def run_agg():
        a = C.c_void_p()
        be.check(be.fn("hash_agg_create")(be.ctx, 1, gb, 2, aggs, C.byref(a)))
        be.check(be.fn("hash_agg_push")(a, b.ptr))
        o = C.POINTER(abi.Batch)()
        be.check(be.fn("hash_agg_finish")(a, D, C.byref(o)))
        groups[0] = o.contents.num_rows
        be.fn("batch_release")(o)
        be.fn("hash_agg_destroy")(a)
    ms = timed(run_agg)
    profile_of(run_agg, "C4 agg")
    by = 16 * n + 24 * groups[0]
    res["C4_agg"] = {"rows": n, "groups": groups[0], "ms": round(ms, 3), "Mrows_s": round(n / ms / 1e3, 1),
                     "GBps": round(by / ms / 1e6, 1), "frac": round(by / ms / 1e6 / HBM_PEAK_GBPS, 4)}
    # (review r05 #3d / #5) counter traffic against the algorithmic bytes, and the same leg in two FRESH processes: the
    # claimed level's time follows the physical placement of its output regions (profiles/r06e_placement.txt: 1.45-1.97 ms
    # for one binary), so one process's figure is a draw like the headline's
    tb, tsrc = ops_traffic(["rp_claim_scatter", "lds_agg_dense_slim"])
    res["C4_agg"].update({"algorithmic_bytes": int(by), "traffic": tb, "traffic_ratio": (round(tb / by, 2) if tb else None),
                          "traffic_source": tsrc})
    if os.environ.get("SQLRS_BENCH_C4_PROCESSES", "1") != "0":
        import re
        import subprocess
        vals = [round(ms, 3)]
        for _ in range(2):
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "c4_agg.py")], capture_output=True, text=True, timeout=300,
                                   env=dict(os.environ, ROUNDS="1", REPS="5"))
                vals.append(float(re.search(r"C4 .*?: ([0-9.]+) ms", r.stdout).group(1)))
            except Exception as e:  # informational
                log(f"[bench] a C4 child process failed: {e!r}")
        sv = sorted(vals)
        res["C4_agg"]["ms_processes"] = {"n": len(vals), "min": sv[0], "median": sv[len(sv) // 2], "max": sv[-1], "values": vals,
                                         "note": "the same leg (tools/c4_agg.py, event-timed best of 5) in fresh processes; values[0] = this process"}
    # ---- C4 over GENERAL int64 keys (round 6: the keys k -> k * A + B no longer fill a range — the groups of a real id
    #      column): hashed buckets, probing LDS tables, 20-byte rows through two partition levels instead of the dense keys'
    #      12-byte rows through one.  Not a BASELINE config; the fraction is of the same algorithmic bytes as C4's.
    key_sp = key * (0x9E3779B97F4A7C15 - (1 << 64)) + 12345
    torch.cuda.synchronize()
    b_dense, b = b, device_batch(abi, [key_sp, val], [abi.INT64, abi.FLOAT64])
    ms_sp = timed(run_agg)
    profile_of(run_agg, "C4 agg sparse keys")
    by_sp = 16 * n + 24 * groups[0]
    res["C4_agg_sparse_keys"] = {"rows": n, "groups": groups[0], "ms": round(ms_sp, 3), "Mrows_s": round(n / ms_sp / 1e3, 1),
                                 "GBps": round(by_sp / ms_sp / 1e6, 1), "frac": round(by_sp / ms_sp / 1e6 / HBM_PEAK_GBPS, 4),
                                 "note": "general (non-dense) int64 group keys: two hashed partition levels + probing LDS tables"}
    b = b_dense
    del key_sp
    # ---- C4 with a WHERE: HashAgg(Filter(scan)), val > 0.5 — the filter handed to the aggregate (sqlrs_hash_agg_set_filter:
    #      evaluated by the partition pass) against the same plan as two operators (filter.rs:13-25 feeding hash_agg.rs:44)
    pred = (InputRef(1) > Constant(0.5, abi.FLOAT64)).pack()
    kept = [0]

    def run_agg_where(fused):
        a = C.c_void_p()
        be.check(be.fn("hash_agg_create")(be.ctx, 1, gb, 2, aggs, C.byref(a)))
        fo = None
        if fused:
            be.check(be.fn("hash_agg_set_filter")(a, C.byref(pred.abi)))
            be.check(be.fn("hash_agg_push")(a, b.ptr))
        else:
            f = C.c_void_p()
            be.check(be.fn("filter_create")(be.ctx, C.byref(pred.abi), C.byref(f)))
            fo = C.POINTER(abi.Batch)()
            be.check(be.fn("filter_push")(f, b.ptr, D, C.byref(fo)))
            be.fn("filter_destroy")(f)
            kept[0] = fo.contents.num_rows
            be.check(be.fn("hash_agg_push")(a, fo))
        o = C.POINTER(abi.Batch)()
        be.check(be.fn("hash_agg_finish")(a, D, C.byref(o)))
        groups[0] = o.contents.num_rows
        ff = be.fn("hash_agg_filter_fused_batches")(a)
        be.fn("batch_release")(o)
        if fo is not None:
            be.fn("batch_release")(fo)
        be.fn("hash_agg_destroy")(a)
        return ff
    ms_two = timed(lambda: run_agg_where(False))
    ms_fused = timed(lambda: run_agg_where(True))
    ff = run_agg_where(True)
    profile_of(lambda: run_agg_where(True), "C4 agg where")
    by = 16 * n + 24 * groups[0]
    res["C4_agg_where"] = {"rows": n, "kept": kept[0], "groups": groups[0], "ms": round(ms_fused, 3), "ms_two_operators": round(ms_two, 3),
                           "filter_fused_batches": int(ff), "Mrows_s": round(n / ms_fused / 1e3, 1),
                           "GBps": round(by / ms_fused / 1e6, 1), "frac": round(by / ms_fused / 1e6 / HBM_PEAK_GBPS, 4)}
    # ---- C4 with a WIDE aggregate list (three argument columns, six aggregates: the TPC-H Q1 shape): run as parts of
    #      <= 2 argument columns each (hashagg_op.hip); `ms_unsplit` = the same operator as ONE part (SQLRS_AGG_SPLIT=0,
    #      read at operator creation), whose rows take the row route's global atomics
    val2 = datagen.fill_chunks(torch.empty(n, dtype=torch.float64, device=dev), lambda i: datagen.val_t(0xF3, i))
    qty = datagen.fill_chunks(torch.empty(n, dtype=torch.int64, device=dev), lambda i: datagen.key_t(0xF4, i, 50))
    torch.cuda.synchronize()
    bw = device_batch(abi, [key, val, val2, qty], [abi.INT64, abi.FLOAT64, abi.FLOAT64, abi.INT64])
    keep_w = []
    aggs_w = (abi.AggFunc * 6)(AggFunc("count", InputRef(1), abi.INT64).abi_struct(keep_w), AggFunc("sum", InputRef(1), abi.FLOAT64).abi_struct(keep_w),
                               AggFunc("sum", InputRef(2), abi.FLOAT64).abi_struct(keep_w), AggFunc("sum", InputRef(3), abi.INT64).abi_struct(keep_w),
                               AggFunc("min", InputRef(2), abi.FLOAT64).abi_struct(keep_w), AggFunc("max", InputRef(3), abi.INT64).abi_struct(keep_w))
    chk = [None]

    def run_agg_wide():
        a = C.c_void_p()
        be.check(be.fn("hash_agg_create")(be.ctx, 1, gb, 6, aggs_w, C.byref(a)))
        be.check(be.fn("hash_agg_push")(a, bw.ptr))
        o = C.POINTER(abi.Batch)()
        be.check(be.fn("hash_agg_finish")(a, D, C.byref(o)))
        groups[0] = o.contents.num_rows
        if chk[0] is None:  # once: SUM(qty) and COUNT(val) over all groups against torch (bit exact), group count
            be.synchronize()
            w_ = be.wrap(o)
            cnt = _tensor_view(torch, w_.column(1).values, groups[0], torch.int64, dev)
            sq = _tensor_view(torch, w_.column(4).values, groups[0], torch.int64, dev)
            chk[0] = bool(int(cnt.sum().item()) == n and int(sq.sum().item()) == int(qty.sum().item()))
            w_.release()
        else:
            be.fn("batch_release")(o)
        be.fn("hash_agg_destroy")(a)
    ms_w = timed(run_agg_wide)
    profile_of(run_agg_wide, "C4 agg wide")
    g_w, ok_w = groups[0], chk[0]
    os.environ["SQLRS_AGG_SPLIT"] = "0"
    try:
        chk[0] = True
        run_agg_wide()
        be.synchronize()
        t0 = time.perf_counter()
        run_agg_wide()
        be.synchronize()
        ms_u = (time.perf_counter() - t0) * 1e3
    finally:
        del os.environ["SQLRS_AGG_SPLIT"]
    by = 32 * n + 56 * g_w
    res["C4_agg_wide_3cols_6aggs"] = {"rows": n, "groups": g_w, "ms": round(ms_w, 3), "ms_unsplit": round(ms_u, 3), "Mrows_s": round(n / ms_w / 1e3, 1),
                                      "GBps": round(by / ms_w / 1e6, 1), "frac": round(by / ms_w / 1e6 / HBM_PEAK_GBPS, 4),
                                      "check": "OK" if ok_w and g_w == G else "mismatch"}
    del val2, qty, bw
    # ---- what a MISS of the optimistic key statistics costs (review r03 #9): two keys far outside the range of the others,
    #      hidden in 6144-row chunks the sample does not read.  They go to the outlier list (row route, radix_part.hpp
    #      key_out_of_range) instead of failing the attempt; `ms_exact` = the same batch with the statistics from a full pass
    #      (SQLRS_KEY_STATS_EXACT=1, read per call), `ms_plain` = the batch without the outliers
    if "C4_agg" in res:
        saved = key[[3 * 6144 + 17, 11 * 6144 + 5]].clone()
        key[3 * 6144 + 17] = 10 ** 12
        key[11 * 6144 + 5] = -(10 ** 12)
        torch.cuda.synchronize()
        ms_out = timed(run_agg)
        g_out = groups[0]
        os.environ["SQLRS_KEY_STATS_EXACT"] = "1"
        ms_exact = timed(run_agg)
        os.environ.pop("SQLRS_KEY_STATS_EXACT", None)
        key[3 * 6144 + 17], key[11 * 6144 + 5] = saved[0], saved[1]
        torch.cuda.synchronize()
        res["C4_agg_outlier_keys"] = {"rows": n, "groups": g_out, "ms": round(ms_out, 3), "ms_exact": round(ms_exact, 3),
                                      "ms_plain": res["C4_agg"]["ms"], "ratio_to_exact": round(ms_out / ms_exact, 3),
                                      "check": "OK" if g_out == res["C4_agg"]["groups"] + 2 else f"groups {g_out}"}
    # ---- C4 with Zipf(1.1) keys over the same 1e6 groups (hot groups: contention / bucket skew)
    import numpy as _np
    w = _np.arange(1, G + 1, dtype=_np.float64) ** -1.1
    cdf = torch.from_numpy(_np.cumsum(w) / w.sum()).to(dev)
    perm_a = datagen._coprime_multiplier(G)
    key = datagen.fill_chunks(key, lambda i: (torch.searchsorted(cdf, datagen.val_t(0xA7, i)).clamp_(max=G - 1) * perm_a + 7) % G)
    torch.cuda.synchronize()
    del cdf
    ms = timed(run_agg)
    profile_of(run_agg, "C4 agg zipf")
    by = 16 * n + 24 * groups[0]
    res["C4_agg_zipf1.1"] = {"rows": n, "groups": groups[0], "ms": round(ms, 3), "Mrows_s": round(n / ms / 1e3, 1),
                             "GBps": round(by / ms / 1e6, 1), "frac": round(by / ms / 1e6 / HBM_PEAK_GBPS, 4)}
    del key, val, b
    # ---- C2 with 10 % NULLs in the predicate column (validity bitmap read, NULL rows dropped)
    n = 100_000_000
    v1 = datagen.fill_chunks(torch.empty(n, dtype=torch.int64, device=dev),
                             lambda i: datagen._lsr(datagen.splitmix64_t(0xC2, i), 33))
    nwords = (n + 63) // 64
    bits = torch.zeros(nwords, dtype=torch.int64, device=dev)
    for bit in range(64):  # bit set = valid with probability 0.9
        r = datagen.val_t(0xC3 + bit, torch.arange(nwords, dtype=torch.int64, device=dev)) < 0.9
        bits |= r.to(torch.int64) << bit
    torch.cuda.synchronize()
    col = abi.device_column(abi.INT64, n, v1.data_ptr(), validity_ptr=bits.data_ptr(), null_count=-1)
    bnull = abi.RawBatch([col], n, keepalive=[v1, bits])
    e = (InputRef(0) > Constant(1 << 30, abi.INT64)).pack()
    kept = [0]

    def run_null():
        f = C.c_void_p()
        be.check(be.fn("filter_create")(be.ctx, C.byref(e.abi), C.byref(f)))
        o = C.POINTER(abi.Batch)()
        be.check(be.fn("filter_push")(f, bnull.ptr, D, C.byref(o)))
        kept[0] = o.contents.num_rows
        be.fn("batch_release")(o)
        be.fn("filter_destroy")(f)
    ms = timed(run_null)
    by = 8 * n + n // 8 + 8 * kept[0]
    res["C2_filter_s0.5_null10"] = {"rows": n, "kept": kept[0], "ms": round(ms, 3), "Mrows_s": round(n / ms / 1e3, 1),
                                    "GBps": round(by / ms / 1e6, 1), "frac": round(by / ms / 1e6 / HBM_PEAK_GBPS, 4)}
    del bnull, bits
    # ---- Order: ORDER BY v1 (int64, 31 significant bits) carrying one f64 column, 1e8 rows
    val = datagen.fill_chunks(torch.empty(n, dtype=torch.float64, device=dev), lambda i: datagen.val_t(0xF2, i))
    torch.cuda.synchronize()
    bo = device_batch(abi, [v1, val], [abi.INT64, abi.FLOAT64])
    pk = InputRef(0).pack()
    obs = (abi.OrderBy * 1)(abi.OrderBy(pk.abi, 1, 0))

    def run_order():
        h = C.c_void_p()
        be.check(be.fn("order_create")(be.ctx, 1, obs, C.byref(h)))
        be.check(be.fn("order_push_retained")(h, bo.ptr))  # `bo` outlives the sort (order.rs:19-26 keeps Arcs)
        o = C.POINTER(abi.Batch)()
        be.check(be.fn("order_finish")(h, D, C.byref(o)))
        be.fn("batch_release")(o)
        be.fn("order_destroy")(h)
    ms = timed(run_order)
    profile_of(run_order, "Order")
    # (round 5: the two split passes run in their look-back form — one histogram of the column, passes chained over their tiles;
    #  `ms_counting_form` = the same call with SQLRS_ORDER_LB=0, read per call: per-tile count matrices + scans as before)
    os.environ["SQLRS_ORDER_LB"] = "0"
    try:
        ms_counting = timed(run_order)
    finally:
        os.environ.pop("SQLRS_ORDER_LB", None)
    # 16 N (read key, write permuted key) + 8 N per carried column = 24 B/row is what `frac` is computed from; the
    # carried column is read AND written, so the operator's streams are 32 B/row: `frac_streams_32B`, a differently named field
    by = 16 * n + 8 * n
    res["Order_int64_1col"] = {"rows": n, "ms": round(ms, 3), "Mrows_s": round(n / ms / 1e3, 1),
                               "GBps": round(by / ms / 1e6, 1), "frac": round(by / ms / 1e6 / HBM_PEAK_GBPS, 4),
                               "bytes_per_row": 24, "frac_streams_32B": round(32 * n / ms / 1e6 / HBM_PEAK_GBPS, 4),
                               "ms_counting_form": round(ms_counting, 3)}
    # ---- what a MISS of the optimistic key range costs (review r03 #9): one key far outside the sampled range in a chunk the
    #      sample skips; the raw pass's histogram kernel notices, the split passes behind it return at once, the exact form
    #      runs.  `ms_exact` = the same column with SQLRS_ORDER_SAMPLE=0 (read per call: key range from a full pass)
    saved1 = v1[5 * 2048 + 3].clone()
    v1[5 * 2048 + 3] = (1 << 40) + 5
    torch.cuda.synchronize()
    ms_out = timed(run_order)
    profile_of(run_order, "Order_outlier")
    os.environ["SQLRS_ORDER_SAMPLE"] = "0"  # (read per call: the exact min / max pass)
    ms_exact = timed(run_order)
    os.environ.pop("SQLRS_ORDER_SAMPLE", None)
    v1[5 * 2048 + 3] = saved1
    torch.cuda.synchronize()
    res["Order_outlier_in_unsampled_chunk"] = {"rows": n, "ms": round(ms_out, 3), "ms_exact": round(ms_exact, 3),
                                               "ms_plain": res["Order_int64_1col"]["ms"], "ratio_to_exact": round(ms_out / ms_exact, 3)}
    # ---- few distinct keys spread over many bits (50 values over 2^26): a group of equal top bits is far larger than the
    #      in-LDS finish takes; the call is redone with all key bits through HBM passes (`ms`), against the general path
    #      (`ms_general`, SQLRS_ORDER_FAST=0 read per call) it used to fall to after the wasted attempt
    g50 = torch.Generator(device=dev).manual_seed(50)
    heavy = torch.randint(0, 1 << 26, (50,), dtype=torch.int64, device=dev, generator=g50)[torch.randint(0, 50, (n,), device=dev, generator=g50)]
    bo_h = device_batch(abi, [heavy, val], [abi.INT64, abi.FLOAT64])

    def run_order_heavy():
        h = C.c_void_p()
        be.check(be.fn("order_create")(be.ctx, 1, obs, C.byref(h)))
        be.check(be.fn("order_push_retained")(h, bo_h.ptr))
        o = C.POINTER(abi.Batch)()
        be.check(be.fn("order_finish")(h, D, C.byref(o)))
        be.fn("batch_release")(o)
        be.fn("order_destroy")(h)
    ms_h = timed(run_order_heavy)
    os.environ["SQLRS_ORDER_FAST"] = "0"
    ms_hg = timed(run_order_heavy)
    os.environ.pop("SQLRS_ORDER_FAST", None)
    res["Order_50_distinct_keys"] = {"rows": n, "ms": round(ms_h, 3), "ms_general": round(ms_hg, 3), "ms_plain": res["Order_int64_1col"]["ms"]}
    del bo_h, heavy
    # ---- keys with more than 32 varying bits (order_fast.hip, order_wide: splitters from a sorted sample): ORDER BY the f64
    #      column (uniform doubles) carrying v1, and ORDER BY a column of random 63-bit integers carrying the f64 column.
    #      `ms_general` = the same call with SQLRS_ORDER_WIDE=0 (read per call): LSD radix sort of (key, row id) + gathers.
    #      `check`: key and carried column against torch.sort(stable=True) of the same column (ties keep input order)
    wide = torch.randint(-(1 << 62), 1 << 62, (n,), dtype=torch.int64, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
    for leg, kcol, ccol, kt, ct in (("Order_f64_random", val, v1, abi.FLOAT64, abi.INT64), ("Order_int64_63bit", wide, val, abi.INT64, abi.FLOAT64)):
        batch = device_batch(abi, [kcol, ccol], [kt, ct])
        checked = [None]

        def run_order_wide(batch=batch, kcol=kcol, ccol=ccol):
            h = C.c_void_p()
            be.check(be.fn("order_create")(be.ctx, 1, obs, C.byref(h)))
            be.check(be.fn("order_push_retained")(h, batch.ptr))
            o = C.POINTER(abi.Batch)()
            be.check(be.fn("order_finish")(h, D, C.byref(o)))
            if checked[0] is None:
                be.synchronize()
                w_ = be.wrap(o)
                got_k = _tensor_view(torch, w_.column(0).values, n, kcol.dtype, dev)
                got_c = _tensor_view(torch, w_.column(1).values, n, ccol.dtype, dev)
                exp_k, perm = torch.sort(kcol, stable=True)
                checked[0] = bool(w_.num_rows == n and torch.equal(got_k, exp_k) and torch.equal(got_c, ccol[perm]))
                del exp_k, perm, got_k, got_c
                w_.release()
            else:
                be.fn("batch_release")(o)
            be.fn("order_destroy")(h)
        ms_w = timed(run_order_wide)
        if leg == "Order_f64_random":
            profile_of(run_order_wide, "Order_f64")
        os.environ["SQLRS_ORDER_WIDE"] = "0"
        ms_g = timed(run_order_wide)
        os.environ.pop("SQLRS_ORDER_WIDE", None)
        res[leg] = {"rows": n, "ms": round(ms_w, 3), "Mrows_s": round(n / ms_w / 1e3, 1), "GBps": round(by / ms_w / 1e6, 1),
                    "frac": round(by / ms_w / 1e6 / HBM_PEAK_GBPS, 4), "bytes_per_row": 24,
                    "frac_streams_32B": round(32 * n / ms_w / 1e6 / HBM_PEAK_GBPS, 4), "ms_general": round(ms_g, 3),
                    "check": "OK" if checked[0] else "mismatch"}
        del batch
        torch.cuda.empty_cache()
    del wide
    # ---- ORDER BY a, b (order.rs:27-66 lexsort): a = 1000 distinct int64 values, b = v1 (31 bits), carrying the f64 column —
    #      one composite key of 41 bits through the single-key routes, both key columns decoded from the sorted composite
    #      (order_fast.hip, order_composite); `ms_general` = SQLRS_ORDER_COMPOSITE=0 (read per call): one stable radix sort of
    #      (key, row id) per key + a gather per column; `check`: against torch's stable sorts composed last key first
    ka = torch.randint(0, 1000, (n,), dtype=torch.int64, device=dev, generator=torch.Generator(device=dev).manual_seed(11))
    bo2 = device_batch(abi, [ka, v1, val], [abi.INT64, abi.INT64, abi.FLOAT64])
    pk0, pk1 = InputRef(0).pack(), InputRef(1).pack()
    obs2 = (abi.OrderBy * 2)(abi.OrderBy(pk0.abi, 1, 0), abi.OrderBy(pk1.abi, 1, 0))
    checked2 = [None]

    def run_order2():
        h = C.c_void_p()
        be.check(be.fn("order_create")(be.ctx, 2, obs2, C.byref(h)))
        be.check(be.fn("order_push_retained")(h, bo2.ptr))
        o = C.POINTER(abi.Batch)()
        be.check(be.fn("order_finish")(h, D, C.byref(o)))
        if checked2[0] is None:
            be.synchronize()
            w_ = be.wrap(o)
            got = [_tensor_view(torch, w_.column(i).values, n, t, dev) for i, t in enumerate((torch.int64, torch.int64, torch.float64))]
            p1 = torch.sort(v1, stable=True).indices
            perm = p1[torch.sort(ka[p1], stable=True).indices]
            checked2[0] = bool(w_.num_rows == n and torch.equal(got[0], ka[perm]) and torch.equal(got[1], v1[perm]) and torch.equal(got[2], val[perm]))
            del got, p1, perm
            w_.release()
        else:
            be.fn("batch_release")(o)
        be.fn("order_destroy")(h)
    ms_2 = timed(run_order2)
    profile_of(run_order2, "Order two keys")
    os.environ["SQLRS_ORDER_COMPOSITE"] = "0"
    ms_2g = timed(run_order2)
    os.environ.pop("SQLRS_ORDER_COMPOSITE", None)
    by2 = 2 * 16 * n + 8 * n  # (§8d: 16 N per key column + 8 N per carried column = 40 B/row; the streams are 48)
    res["Order_two_int64_keys"] = {"rows": n, "ms": round(ms_2, 3), "Mrows_s": round(n / ms_2 / 1e3, 1), "GBps": round(by2 / ms_2 / 1e6, 1),
                                   "frac": round(by2 / ms_2 / 1e6 / HBM_PEAK_GBPS, 4), "bytes_per_row": 40,
                                   "frac_streams_48B": round(48 * n / ms_2 / 1e6 / HBM_PEAK_GBPS, 4), "ms_general": round(ms_2g, 3),
                                   "check": "OK" if checked2[0] else "mismatch"}
    del bo2, ka
    torch.cuda.empty_cache()
    # ---- ORDER BY v1 LIMIT 100 — PhysicalLimit(PhysicalOrder(scan)): offset + limit handed to the sort (sqlrs_order_set_limit),
    #      which sorts the candidates below a sampled threshold only; checked against torch.topk on the same column
    K = 100
    first = [None]

    def run_order_limit():
        h = C.c_void_p()
        be.check(be.fn("order_create")(be.ctx, 1, obs, C.byref(h)))
        be.check(be.fn("order_set_limit")(h, K))
        be.check(be.fn("order_push_retained")(h, bo.ptr))
        o = C.POINTER(abi.Batch)()
        be.check(be.fn("order_finish")(h, D, C.byref(o)))
        lim = C.c_void_p()
        be.check(be.fn("limit_create")(be.ctx, 1, K, 0, 0, C.byref(lim)))
        lo = C.POINTER(abi.Batch)()
        fin = C.c_int()
        be.check(be.fn("limit_push")(lim, o, D, C.byref(lo), C.byref(fin)))
        if first[0] is None:
            be.synchronize()
            w_ = be.wrap(lo)
            got_k = _tensor_view(torch, w_.column(0).values, w_.num_rows, torch.int64, dev).clone()
            exp_k = torch.topk(v1, K, largest=False, sorted=True).values
            first[0] = (bool(w_.num_rows == K and torch.equal(got_k, exp_k)), int(be.fn("order_topk_candidates")(h)))
            w_.release()
        else:
            be.fn("batch_release")(lo)
        be.fn("limit_destroy")(lim)
        be.fn("batch_release")(o)
        be.fn("order_destroy")(h)
    ms_k = timed(run_order_limit)
    profile_of(run_order_limit, "Order limit")
    res["Order_int64_limit100"] = {"rows": n, "limit": K, "ms": round(ms_k, 3), "ms_full_sort": round(ms, 3), "candidates": first[0][1],
                                   "Mrows_s": round(n / ms_k / 1e3, 1), "check": "OK" if first[0][0] else "mismatch"}
    del bo, v1, val
    be.fn("ctx_pool_trim")(be.ctx)
    torch.cuda.empty_cache()
    if not os.environ.get("SQLRS_BENCH_SKIP_HOST"):  # (tools/profile_round.sh: the counter passes want the device legs only)
        res.update(bench_host_resident(be, abi, datagen, torch, dev))
    for k_, v_ in res.items():
        log(f"[bench] {k_}: {v_}")
    return res


def bench_host_resident(be, abi, datagen, torch, dev):
    """What a drop-in actually feeds (north_star: "Arrow column buffers move to HBM once per pipeline"): HOST-resident
    inputs, PCIe inclusive, wall clock.  (a) C5_host_1e8: the C5 query over 1e8 fact rows x 1e7 dim rows held in PINNED
    host memory, handed to the fused operator in 2^24-row batches (each column crosses the link once: 1.76 GB);
    (b) C4_host_batches_1024: HashAgg fed the reference's CSV batch shape (1024 rows, storage/csv.rs:105) from pageable
    host memory through the library's host staging.  Never part of `value`."""
    from sqlrs_amd.expr import AggFunc, Constant, InputRef
    PCIE_GBPS = 63.0  # MI355X_MICROARCH.md: PCIe Gen5 x16
    H = abi.MEM_HOST
    out = {}
    # ---- (a)
    nP, nB, step_rows = 100_000_000, 10_000_000, 1 << 24
    fk = datagen.fill_chunks(torch.empty(nP, dtype=torch.int64, device=dev), lambda i: datagen.key_t(0xF1, i, nB)).cpu().pin_memory()
    fv = datagen.fill_chunks(torch.empty(nP, dtype=torch.float64, device=dev), lambda i: datagen.val_t(0xF2, i)).cpu().pin_memory()
    dk = datagen.fill_chunks(torch.empty(nB, dtype=torch.int64, device=dev), lambda i: datagen.dim_key_t(i, nB)).cpu().pin_memory()
    torch.cuda.synchronize()

    def host_batch(tensors, dtypes, lo, hi):
        cols = [abi.Column(dt, H, hi - lo, 0, t.data_ptr() + lo * t.element_size(), None, None) for t, dt in zip(tensors, dtypes)]
        return abi.RawBatch(cols, hi - lo, keepalive=tensors)
    lk, _k1 = abi.pack_exprs([InputRef(0)])
    rk, _k2 = abi.pack_exprs([InputRef(0)])
    gb, _k3 = abi.pack_exprs([InputRef(0)])
    keep = []
    aggs = (abi.AggFunc * 2)(AggFunc("count", InputRef(2), abi.INT64).abi_struct(keep), AggFunc("sum", InputRef(2), abi.FLOAT64).abi_struct(keep))
    rd = (C.c_int32 * 2)(abi.INT64, abi.FLOAT64)
    pf = (InputRef(1) > Constant(0.5, abi.FLOAT64)).pack()
    groups = [0]

    def run_c5_host():
        ja = C.c_void_p()
        be.check(be.fn("join_agg_create")(be.ctx, 1, lk, rk, 1, 2, rd, 1, gb, 2, aggs, C.byref(ja)))
        be.check(be.fn("join_agg_set_probe_filter")(ja, C.byref(pf.abi)))
        be.check(be.fn("join_agg_build_push")(ja, host_batch([dk], [abi.INT64], 0, nB).ptr))
        be.check(be.fn("join_agg_build_finish")(ja))
        for lo in range(0, nP, step_rows):
            be.check(be.fn("join_agg_probe_push")(ja, host_batch([fk, fv], [abi.INT64, abi.FLOAT64], lo, min(nP, lo + step_rows)).ptr))
        o = C.POINTER(abi.Batch)()
        be.check(be.fn("join_agg_finish")(ja, abi.MEM_DEVICE, C.byref(o)))
        groups[0] = o.contents.num_rows
        be.fn("batch_release")(o)
        be.fn("join_agg_destroy")(ja)
        be.synchronize()
    run_c5_host()
    best = 1e30
    for _ in range(2):
        t = time.perf_counter()
        run_c5_host()
        best = min(best, time.perf_counter() - t)
    moved = 16 * nP + 8 * nB
    out["C5_host_1e8"] = {"fact_rows": nP, "dim_rows": nB, "groups": groups[0], "ms": round(best * 1e3, 2),
                          "Mrows_s": round(nP / best / 1e6, 1), "pcie_GBps": round(moved / best / 1e9, 1),
                          "pcie_frac_of_63GBps": round(moved / best / 1e9 / PCIE_GBPS, 3),
                          "note": "pinned host columns -> sqlrs_join_agg_* with the probe Filter, 2^24-row probe batches, result left in HBM; wall clock"}
    del fk, fv, dk
    # ---- (b)
    import pyarrow as pa
    n, B, G = 20_000_000, 1024, 1_000_000
    idx = np.arange(n, dtype=np.int64)
    fact = pa.RecordBatch.from_arrays([pa.array(datagen.key_np(0xA1, idx, G)), pa.array(datagen.val_np(0xF2, idx))], names=["key", "val"])
    fb = [abi.HostBatch(fact.slice(lo, B)) for lo in range(0, n, B)]  # marshalled once: the binding's cost is not the library's
    aggs4 = (abi.AggFunc * 2)(AggFunc("count", InputRef(1), abi.INT64).abi_struct(keep), AggFunc("sum", InputRef(1), abi.FLOAT64).abi_struct(keep))

    def run_c4_host():
        a = C.c_void_p()
        be.check(be.fn("hash_agg_create")(be.ctx, 1, gb, 2, aggs4, C.byref(a)))
        push = be.fn("hash_agg_push")
        for b in fb:
            be.check(push(a, b.ptr))
        o = C.POINTER(abi.Batch)()
        be.check(be.fn("hash_agg_finish")(a, H, C.byref(o)))
        groups[0] = o.contents.num_rows
        be.fn("batch_release")(o)
        be.fn("hash_agg_destroy")(a)
    run_c4_host()
    t = time.perf_counter()
    run_c4_host()
    dt = time.perf_counter() - t
    out["C4_host_batches_1024"] = {"rows": n, "batches": len(fb), "groups": groups[0], "ms": round(dt * 1e3, 1),
                                   "Mrows_s": round(n / dt / 1e6, 1), "pcie_GBps": round(16 * n / dt / 1e9, 2),
                                   "pcie_frac_of_63GBps": round(16 * n / dt / 1e9 / PCIE_GBPS, 4),
                                   "note": "19532 pageable 1024-row host batches (storage/csv.rs:105) through sqlrs_hash_agg_push's host "
                                           "staging (one upload per 2^22 rows), result on the host; wall clock incl. one ctypes call per batch"}
    # the same leg from a NATIVE caller (host/bench_host_batches.cpp: a C call per batch, what a Rust drop-in pays)
    exe = os.path.join(ROOT, "host", "bench_host_batches")
    if os.path.exists(exe):
        try:
            import subprocess
            r = subprocess.run([exe, str(n), "1000000", "1024"], capture_output=True, text=True, timeout=300)
            out["C4_host_batches_1024_native"] = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:  # informational leg: never takes the line down
            out["C4_host_batches_1024_native"] = {"error": repr(e)[:200]}
        # the STREAMING operators at the same batch shape (round-3 review: "a real drop-in would be slower than the CPU there,
        # and the line hides it"): Filter per batch and through sqlrs_filter_push_many; HashJoin probe per batch and _many
        # (round 6: Project — SELECT v, v * 3 + 1, v > k — per batch and through sqlrs_project_push_async)
        for leg, mode in (("C2_host_batches_1024", "filter"), ("C3_probe_host_batches_1024", "probe"), ("Project_host_batches_1024", "project")):
            try:
                r = subprocess.run([exe, mode], capture_output=True, text=True, timeout=600)
                out[leg] = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception as e:
                out[leg] = {"error": repr(e)[:200]}
    return out


def _tensor_view(torch, ptr, n, dtype, dev):
    """zero-copy torch view over a library-owned device buffer"""
    if n == 0:
        return torch.empty(0, dtype=dtype, device=dev)
    itemsize = torch.empty(0, dtype=dtype).element_size()

    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (n,), "typestr": {torch.int64: "<i8", torch.float64: "<f8", torch.int32: "<i4"}[dtype],
                                  "data": (int(ptr), False), "version": 2, "strides": (itemsize,)}
    return torch.as_tensor(h, device=dev)


def cpu_baseline(args, abi, datagen, n_dim_total):
    """The CPU path timed on this box's host cores, same query, same generator, dim kept at its full size:
    (main) the all-core "fair" port — oracle/cpu_fair.cpp: OpenMP over every host core, radix partitioning,
    flat open-addressing tables — and (`faithful`) the single-threaded restatement of what the reference
    does today (oracle/sqlrs_oracle.cpp; the reference never spawns a thread).  Both on bounded samples
    of the fact table.  Test/bench infrastructure only."""
    import pyarrow as pa
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_fair
    from oracle_backend import load_oracle
    from sqlrs_amd.executor import FilterExecutor, HashAggExecutor, HashJoinExecutor
    from sqlrs_amd.expr import AggFunc, Constant, InputRef, JoinCondition

    threads = args.cpu_threads or cpu_fair.max_threads()
    # ---- fair: grow the sample until one run takes ~2 s (or 4e8 rows = 6.4 GB of columns), best of 3
    n = int(args.cpu_sample_rows) or 50_000_000
    while True:
        fk, fv, dk = cpu_fair.gen_c5(n, n_dim_total)
        _, _, _, sec = cpu_fair.run_c5(fk, fv, dk, args.threshold, threads)
        if args.cpu_sample_rows or sec >= 1.5 or n >= 400_000_000:
            break
        n = int(min(400_000_000, max(2 * n, n * 2.0 / max(sec, 1e-3))))
        del fk, fv, dk
    best = sec
    for _ in range(2):
        keys, cnt, sm, sec = cpu_fair.run_c5(fk, fv, dk, args.threshold, threads)
        best = min(best, sec)
    fair = n / best / 1e6
    kept = int((fv > args.threshold).sum())
    if int(cnt.sum()) != kept:  # every key has a partner in this workload
        raise SystemExit(f"cpu_baseline (fair) result check failed: {int(cnt.sum())} joined rows, expected {kept}")
    log(f"[bench] cpu_baseline fair: {n:,} fact rows x {n_dim_total:,} dim rows in {best:.2f}s on {threads} threads = {fair:.1f} Mrows/s")
    del fk, fv, dk

    # ---- faithful: 1 thread, unordered_map keyed by hash, per-batch per-group take + accumulate
    # (bounded: the std::unordered_map build over all 1e7 dim keys alone takes ~45 s, so this leg keeps
    #  the dim at 1e6 rows; the fair leg above runs against the full dim)
    oracle = load_oracle()
    n1, nd1 = 10_000_000, min(n_dim_total, 1_000_000)
    idx = np.arange(n1, dtype=np.int64)
    fact = pa.RecordBatch.from_arrays([pa.array(datagen.key_np(0xF1, idx, nd1)), pa.array(datagen.val_np(0xF2, idx))],
                                      names=["key", "val"])
    dim = pa.RecordBatch.from_arrays([pa.array(datagen.dim_key_np(np.arange(nd1, dtype=np.int64), nd1))], names=["key"])
    schema = pa.schema([("d.key", pa.int64()), ("f.key", pa.int64()), ("f.val", pa.float64())])
    t = time.perf_counter()
    filt = FilterExecutor(oracle, InputRef(1) > Constant(args.threshold, abi.FLOAT64), [fact])
    join = HashJoinExecutor(oracle, [dim], filt.execute(), "inner", JoinCondition([(InputRef(0), InputRef(0))]), schema, 1)
    agg = HashAggExecutor(oracle, [AggFunc("count", InputRef(2), abi.INT64), AggFunc("sum", InputRef(2), abi.FLOAT64)],
                          [InputRef(0)], join.execute())
    (out,) = list(agg.execute())
    dt = time.perf_counter() - t
    faithful = n1 / dt / 1e6
    log(f"[bench] cpu_baseline faithful: {n1:,} fact rows x {nd1:,} dim rows in {dt:.1f}s on 1 thread = {faithful:.3f} Mrows/s")
    return {"value": round(fair, 2), "unit": "Mrows/s", "cores": threads, "kind": "port",
            "sample": f"first {n} fact rows x all {n_dim_total} dim rows, single batch, same query and generator, "
                      f"oracle/libsqlrs_cpu_fair.so (OpenMP, {threads} threads = omp_get_max_threads() = the "
                      f"{len(os.sched_getaffinity(0))} cpus this process is allowed to run on (sched_getaffinity) of the host's "
                      f"{os.cpu_count()}: more threads than that would share cores; --cpu-threads overrides, radix-partitioned "
                      "flat tables), best of 3",
            "faithful": {"value": round(faithful, 4), "unit": "Mrows/s", "cores": 1, "kind": "port",
                         "sample": f"first {n1} fact rows x {nd1} dim rows, single batch, oracle/libsqlrs_oracle.so: the "
                                   "reference's algorithm as it is (1 thread, unordered_map keyed by hash, per-batch "
                                   "per-group take + accumulate)"}}


if __name__ == "__main__":
    main()
