"""Multi-GPU exchange for the partitioned hash join / group-by.

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI).  Equi-join and
group-by couple rows only through equal keys, so the path shards with ONE exchange step:

    1. every rank hash-partitions its slice on the key   (device: ``sqlrs_hash_partition``)
    2. slices are exchanged with one all-to-all per column (xGMI is a full mesh: every pairwise
       slice rides its own link, so an all-to-all is the topology's best case)
    3. every rank runs the ordinary operators on what it received; with group key = join key the
       per-rank results are disjoint and the global result is their concatenation.

``partition_of`` restates the device partition function in numpy so that the exchange logic is
testable on CPU (``gloo``, world_size 2) and so that tests can check device and host agree.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

_MASK = (1 << 64) - 1


def mix64_np(x: np.ndarray) -> np.ndarray:
    """device_utils.hpp mix64 (murmur3 finaliser), vectorised on uint64"""
    with np.errstate(over="ignore"):
        x = x.astype(np.uint64)
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xFF51AFD7ED558CCD)
        x ^= x >> np.uint64(33)
        x *= np.uint64(0xC4CEB9FE1A85EC53)
        x ^= x >> np.uint64(33)
        return x


def partition_of(keys: np.ndarray, parts: int, valid: Optional[np.ndarray] = None) -> np.ndarray:
    """partition.hip part_of(): p = ((mix64(key ^ C) >> 32) * parts) >> 32; NULL keys -> 0"""
    h = mix64_np(keys.view(np.uint64) ^ np.uint64(0x5851F42D4C957F2D))
    p = (((h >> np.uint64(32)) * np.uint64(parts)) >> np.uint64(32)).astype(np.uint32)
    if valid is not None:
        p = np.where(valid, p, np.uint32(0))
    return p


def partition_numpy(columns: Sequence[np.ndarray], parts: int, valid: Optional[np.ndarray] = None):
    """Host restatement of sqlrs_hash_partition on column 0: rows permuted so that partition p is
    contiguous (input order kept inside a partition) + the parts+1 offsets."""
    p = partition_of(columns[0], parts, valid)
    order = np.argsort(p, kind="stable")
    counts = np.bincount(p, minlength=parts)
    offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    return [c[order] for c in columns], offsets.tolist()


def partition_filter_numpy(columns: Sequence[np.ndarray], parts: int, keep: Optional[np.ndarray] = None):
    """Host restatement of sqlrs_hash_partition_filter's output layout on column 0: every partition owns a
    region of ``cap`` = len rows (rounded up to 64), filled from its start with the kept rows of that
    partition; returns (region columns, part_start, part_rows).  The device fills a region in tile-claim
    order; here input order is kept (any order inside a partition is a valid output)."""
    n = len(columns[0])
    cap = (max(n, 1) + 63) // 64 * 64
    p = partition_of(columns[0], parts)
    sel = np.ones(n, dtype=bool) if keep is None else keep.astype(bool)
    outs = [np.zeros(parts * cap, dtype=c.dtype) for c in columns]
    starts, rows = [], []
    for q in range(parts):
        idx = np.nonzero(sel & (p == q))[0]
        for o, c in zip(outs, columns):
            o[q * cap:q * cap + len(idx)] = c[idx]
        starts.append(q * cap)
        rows.append(len(idx))
    return outs, starts, rows


def all_to_all_columns(dist, columns, offsets: Sequence[int], world: int, torch):
    """Exchanges partitioned columns: slice p of every column goes to rank p.  ``columns`` are
    torch tensors (device or CPU) already in partition order with ``offsets`` (world+1 ints).
    Returns the received columns (concatenation of every rank's slice for this rank)."""
    dev = columns[0].device
    send = torch.tensor([offsets[p + 1] - offsets[p] for p in range(world)], dtype=torch.int64, device=dev)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send)
    sc, rc = send.tolist(), recv.tolist()
    outs = []
    for t in columns:
        dst = torch.empty(int(sum(rc)), dtype=t.dtype, device=dev)
        dist.all_to_all_single(dst, t.contiguous(), output_split_sizes=rc, input_split_sizes=sc)
        outs.append(dst)
    return outs


class ChunkedExchange:
    """All-to-all of hash-partitioned column chunks with the exchange of chunk k overlapping the
    partitioning of chunk k+1.

    * the per-chunk split sizes travel over ``count_group`` (a CPU / gloo group): the host learns how
      much it will receive without touching the device stream, so the payload collectives are never
      waited for inside the loop;
    * payload columns go through ``dist.all_to_all_single(..., async_op=True)`` on ``data_group``
      (RCCL over xGMI on the GPU box) straight into slices of ONE receive buffer per column, so the
      operators downstream see a single batch (no concatenation copy);
    * ``finish()`` waits for the collectives and returns the received columns.

    ``wire_out`` / ``wire_in`` adapt tensors for the data group (identity on RCCL; ``.cpu()`` / ``.to(dev)``
    when a test drives GPU tensors through gloo)."""

    def __init__(self, dist, torch, world: int, dtypes, device, capacity_rows: int, data_group=None, count_group=None,
                 wire_out=None, wire_in=None):
        self.dist, self.torch, self.world = dist, torch, world
        self.data_group, self.count_group = data_group, count_group
        self.wire_out = wire_out or (lambda t: t)
        self.wire_in = wire_in or (lambda t: t)
        self.device = device
        self.dtypes = list(dtypes)
        self.cap = max(int(capacity_rows), 1)
        self.bufs = [torch.empty(self.cap, dtype=dt, device=device) for dt in self.dtypes]
        self.filled = 0
        self.pending = []   # (work, keepalive, staged) per payload collective
        self.bytes_off_rank = 0
        self.rank = dist.get_rank()

    def _grow(self, need: int):
        self._wait()  # the old buffers are targets of in-flight collectives
        new_cap = max(need, self.cap * 2)
        nb = [self.torch.empty(new_cap, dtype=dt, device=self.device) for dt in self.dtypes]
        for o, n in zip(self.bufs, nb):
            n[:self.filled].copy_(o[:self.filled])
        self.bufs, self.cap = nb, new_cap

    def _wait(self):
        for work, keep, staged in self.pending:
            work.wait()
            if staged is not None:  # the data group moved host tensors: land them in the receive buffer
                dst, src = staged
                dst.copy_(self.wire_in(src))
        self.pending = []

    def send_chunk(self, columns, offsets):
        """columns: tensors of one chunk in partition order, offsets: world + 1 row offsets (host ints)"""
        torch, dist, W = self.torch, self.dist, self.world
        sc = [int(offsets[p + 1] - offsets[p]) for p in range(W)]
        send = torch.tensor(sc, dtype=torch.int64)
        recv = torch.empty(W, dtype=torch.int64)
        dist.all_to_all_single(recv, send, group=self.count_group)  # CPU group: no device synchronisation
        rc = [int(x) for x in recv.tolist()]
        total = sum(rc)
        if self.filled + total > self.cap:
            self._grow(self.filled + total)
        for ci, col in enumerate(columns):
            dst = self.bufs[ci][self.filled:self.filled + total]
            src = self.wire_out(col.contiguous())
            if src.device == dst.device:
                work = dist.all_to_all_single(dst, src, output_split_sizes=rc, input_split_sizes=sc, group=self.data_group,
                                              async_op=True)
                self.pending.append((work, (src, col), None))
            else:
                tmp = torch.empty(total, dtype=src.dtype, device=src.device)
                work = dist.all_to_all_single(tmp, src, output_split_sizes=rc, input_split_sizes=sc, group=self.data_group,
                                              async_op=True)
                self.pending.append((work, (src, col), (dst, tmp)))
            self.bytes_off_rank += (sum(sc) - sc[self.rank]) * col.element_size()
        self.filled += total

    def send_regions(self, columns, starts, rows):
        """Like send_chunk for the output of ``sqlrs_hash_partition_filter``: partition p of every column is
        ``col[starts[p] : starts[p] + rows[p]]`` — regions with padding between them, so the slices travel as
        a LIST all-to-all (RCCL: one grouped send/recv per peer straight out of the regions and into the
        receive buffer; gloo has no list all-to-all: the same exchange spelled as isend / irecv pairs)."""
        torch, dist, W = self.torch, self.dist, self.world
        sc = [int(r) for r in rows]
        send = torch.tensor(sc, dtype=torch.int64)
        recv = torch.empty(W, dtype=torch.int64)
        dist.all_to_all_single(recv, send, group=self.count_group)  # CPU group: no device synchronisation
        rc = [int(x) for x in recv.tolist()]
        total = sum(rc)
        if self.filled + total > self.cap:
            self._grow(self.filled + total)
        backend = dist.get_backend(self.data_group)
        for ci, col in enumerate(columns):
            dst = self.bufs[ci][self.filled:self.filled + total]
            ins = [self.wire_out(col[int(starts[p]):int(starts[p]) + sc[p]]) for p in range(W)]
            staged = None
            if ins[0].device == dst.device:
                outs = list(torch.split(dst, rc))
            else:  # the data group moves host tensors: land them in the receive buffer afterwards
                tmp = torch.empty(total, dtype=ins[0].dtype, device=ins[0].device)
                outs, staged = list(torch.split(tmp, rc)), (dst, tmp)
            if backend == "nccl":
                work = dist.all_to_all(outs, ins, group=self.data_group, async_op=True)
                self.pending.append((work, (ins, outs, col), staged))
            else:
                outs[self.rank].copy_(ins[self.rank])
                ops = []
                for r in range(W):
                    if r == self.rank:
                        continue
                    if sc[r]:
                        ops.append(dist.P2POp(dist.isend, ins[r], r, group=self.data_group))
                    if rc[r]:
                        ops.append(dist.P2POp(dist.irecv, outs[r], r, group=self.data_group))
                works = dist.batch_isend_irecv(ops) if ops else []
                for k, wk in enumerate(works):
                    self.pending.append((wk, (ins, outs, col), staged if k == len(works) - 1 else None))
                if not works and staged is not None:
                    staged[0].copy_(self.wire_in(staged[1]))
            self.bytes_off_rank += (sum(sc) - sc[self.rank]) * col.element_size()
        self.filled += total

    def finish(self):
        """-> received columns (views of the receive buffers), valid on the current stream"""
        self._wait()
        return [b[:self.filled] for b in self.bufs]


def shard_bounds(total: int, rank: int, world: int):
    """contiguous slice [lo, hi) of a table owned by ``rank`` before the exchange"""
    return total * rank // world, total * (rank + 1) // world
