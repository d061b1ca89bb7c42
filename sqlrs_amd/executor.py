"""Host-side mirror of the reference's operator structs over the C ABI.

Each class has the reference struct's fields and an ``execute()`` that yields
RecordBatches, like the ``#[try_stream] execute(self) -> BoxedExecutor`` generators of
src/executor/{filter.rs:7-25, join/hash_join.rs:16-23,146-323,
aggregate/hash_agg.rs:15-19,32-150, order.rs:8-67}.  A child is any iterable of batches
(the reference's tests fake children with ``futures::stream::iter(vec_of_batches)``,
hash_join.rs:407-414); a batch is a ``pyarrow.RecordBatch`` (host) or an ``abi.LibBatch``
(already HBM resident, e.g. the output of an upstream operator run with
``out_mem=abi.MEM_DEVICE``), so a Filter -> HashJoin -> HashAgg chain moves its columns
to HBM once.

``backend`` is an ``abi.Backend``: the HIP library in the product (``sqlrs_amd.hip()``)
— the parity tests pass the oracle's Backend to run the identical plan on the CPU
restatement.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, List, Optional, Sequence

import pyarrow as pa

from . import abi
from .expr import AggFunc, BoundExpr, JoinCondition, OrderBy

JoinType = {"inner": abi.JOIN_INNER, "left": abi.JOIN_LEFT, "right": abi.JOIN_RIGHT,
            "full": abi.JOIN_FULL}


def _emit(backend: abi.Backend, p, out_mem: int, names: Optional[Sequence[str]]):
    lb = backend.wrap(p)
    if lb is None:
        return None
    if out_mem == abi.MEM_DEVICE:
        return lb
    try:
        return lb.to_arrow(names)
    finally:
        lb.release()


def _async_stream(be, child, depth, push, names_of):
    """the stream adaptor of the async entry points: `depth` tickets in flight, batches handed out in input order"""
    from collections import deque
    q = deque()

    def wait_one():
        t, names = q.popleft()
        out = C.POINTER(abi.Batch)()
        be.check(be.fn("batch_wait")(t, C.byref(out)))
        return _emit(be, out, abi.MEM_HOST, names)
    try:
        for batch in child:
            b = abi.as_batch(batch)
            t = C.c_void_p()
            be.check(push(b, C.byref(t)))
            q.append((t, names_of(batch)))
            if len(q) > depth:
                r = wait_one()
                if r is not None:
                    yield r
        while q:
            r = wait_one()
            if r is not None:
                yield r
    finally:
        while q:  # (an error / an abandoned generator: every ticket is consumed)
            t, _ = q.popleft()
            out = C.POINTER(abi.Batch)()
            if be.fn("batch_wait")(t, C.byref(out)) == abi.OK and out:
                be.fn("batch_release")(out)


def _names_of(batch) -> Optional[List[str]]:
    if isinstance(batch, pa.RecordBatch):
        return list(batch.schema.names)
    return getattr(batch, "names", None)


class FilterExecutor:
    """``FilterExecutor { expr, child }`` (filter.rs:7-10)."""

    def __init__(self, backend: abi.Backend, expr: BoundExpr, child: Iterable,
                 out_mem: int = abi.MEM_HOST, many: int = 0, depth: int = 0):
        self.backend, self.expr, self.child, self.out_mem = backend, expr, child, out_mem
        # depth > 0: sqlrs_filter_push_async with that many tickets in flight — the same stream of HOST batches, the
        # operator polled one batch at a time as the reference does, no stream synchronisation per batch
        self.depth = depth
        # many > 1: pull that many batches of the child, hand them to sqlrs_filter_push_many together and yield its
        # outputs one by one — the same stream of batches (one per input batch, filter.rs:15-24), fewer uploads
        self.many = many

    def _execute_many(self, be, h):
        group = []

        def flush():
            n = len(group)
            hb = [abi.as_batch(b) for b in group]
            ins = (C.POINTER(abi.Batch) * n)(*[C.pointer(b.abi) if hasattr(b, "abi") else b.ptr for b in hb])
            outs = (C.POINTER(abi.Batch) * n)()
            be.check(be.fn("filter_push_many")(h, n, ins, self.out_mem, outs))
            res = [_emit(be, outs[i], self.out_mem, _names_of(group[i])) for i in range(n)]
            group.clear()
            return res
        for batch in self.child:
            group.append(batch)
            if len(group) == self.many:
                yield from flush()
        if group:
            yield from flush()

    def execute(self):
        be = self.backend
        packed = self.expr.pack()
        h = C.c_void_p()
        be.check(be.fn("filter_create")(be.ctx, C.byref(packed.abi), C.byref(h)))
        try:
            if self.many > 1 and getattr(be.lib, be.prefix + "filter_push_many", None) is not None:
                yield from self._execute_many(be, h)
                return
            if self.depth > 0 and self.out_mem == abi.MEM_HOST and getattr(be.lib, be.prefix + "filter_push_async", None) is not None:
                yield from _async_stream(be, self.child, self.depth, lambda b, t: be.fn("filter_push_async")(h, b.ptr, t), _names_of)
                return
            for batch in self.child:  # filter.rs:15-24
                b = abi.as_batch(batch)
                out = C.POINTER(abi.Batch)()
                be.check(be.fn("filter_push")(h, b.ptr, self.out_mem, C.byref(out)))
                yield _emit(be, out, self.out_mem, _names_of(batch))
        finally:
            be.fn("filter_destroy")(h)


class HashJoinExecutor:
    """``HashJoinExecutor { left_child, right_child, join_type, join_condition,
    join_output_schema }`` (hash_join.rs:16-23).  ``join_output_schema`` is a pyarrow
    schema: field names ``"{table_id}.{column_id}"`` and forced nullability come from the
    planner (catalog/mod.rs:131-137, logical_join.rs:82-116); the library only needs the
    right-hand dtypes (all-NULL tail columns, hash_join.rs:309-317)."""

    def __init__(self, backend: abi.Backend, left_child: Iterable, right_child: Iterable,
                 join_type: str, join_condition: JoinCondition, join_output_schema: pa.Schema,
                 num_left_columns: int, out_mem: int = abi.MEM_HOST, many: int = 0, depth: int = 0):
        self.backend = backend
        self.depth = depth  # > 0: probe batches through sqlrs_hash_join_probe_push_async, that many tickets in flight
        # many > 1: that many probe batches go to sqlrs_hash_join_probe_push_many together (same stream of joined batches)
        self.many = many
        self.left_child, self.right_child = left_child, right_child
        self.join_type, self.join_condition = join_type, join_condition
        self.join_output_schema = join_output_schema
        self.num_left_columns = num_left_columns
        self.out_mem = out_mem

    def _create(self):
        be = self.backend
        keep = []
        lk, k1 = abi.pack_exprs([l for l, _ in self.join_condition.on])
        rk, k2 = abi.pack_exprs([r for _, r in self.join_condition.on])
        keep += [lk, rk, k1, k2]
        filt = None
        if self.join_condition.filter is not None:
            pf = self.join_condition.filter.pack()
            keep.append(pf)
            filt = C.byref(pf.abi)
        right_fields = list(self.join_output_schema)[self.num_left_columns:]
        rd = (C.c_int32 * max(len(right_fields), 1))(*[abi.dtype_of(f.type) for f in right_fields])
        h = C.c_void_p()
        be.check(be.fn("hash_join_create")(
            be.ctx, JoinType[self.join_type.lower()], len(self.join_condition.on), lk, rk, filt,
            len(right_fields), rd, C.byref(h)))
        return h, keep

    def execute(self, indices_only: bool = False):
        be = self.backend
        h, keep = self._create()
        names = list(self.join_output_schema.names)
        try:
            for batch in self.left_child:  # build phase, hash_join.rs:161-181
                b = abi.as_batch(batch)
                be.check(be.fn("hash_join_build_push")(h, b.ptr))
            be.check(be.fn("hash_join_build_finish")(h))
            if self.many > 1 and not indices_only and getattr(be.lib, be.prefix + "hash_join_probe_push_many", None) is not None:
                group = []

                def flush():
                    n = len(group)
                    hb = [abi.as_batch(b) for b in group]
                    ins = (C.POINTER(abi.Batch) * n)(*[C.pointer(b.abi) if hasattr(b, "abi") else b.ptr for b in hb])
                    outs = (C.POINTER(abi.Batch) * n)()
                    be.check(be.fn("hash_join_probe_push_many")(h, n, ins, self.out_mem, outs))
                    group.clear()
                    return [r for r in (_emit(be, outs[i], self.out_mem, names) for i in range(n)) if r is not None]
                for batch in self.right_child:
                    group.append(batch)
                    if len(group) == self.many:
                        yield from flush()
                if group:
                    yield from flush()
                out = C.POINTER(abi.Batch)()  # tail, hash_join.rs:296-322
                be.check(be.fn("hash_join_finish")(h, self.out_mem, C.byref(out)))
                r = _emit(be, out, self.out_mem, names)
                if r is not None:
                    yield r
                return
            if self.depth > 0 and not indices_only and self.out_mem == abi.MEM_HOST and \
                    getattr(be.lib, be.prefix + "hash_join_probe_push_async", None) is not None:
                yield from _async_stream(be, self.right_child, self.depth,
                                         lambda b, t: be.fn("hash_join_probe_push_async")(h, b.ptr, t), lambda _b: names)
                out = C.POINTER(abi.Batch)()  # tail, hash_join.rs:296-322
                be.check(be.fn("hash_join_finish")(h, self.out_mem, C.byref(out)))
                r = _emit(be, out, self.out_mem, names)
                if r is not None:
                    yield r
                return
            for batch in self.right_child:  # probe phase, hash_join.rs:207-292
                b = abi.as_batch(batch)
                out = C.POINTER(abi.Batch)()
                if indices_only:
                    be.check(be.fn("hash_join_probe_indices")(h, b.ptr, self.out_mem, C.byref(out)))
                    r = _emit(be, out, self.out_mem, ["left_indices", "right_indices"])
                else:
                    be.check(be.fn("hash_join_probe_push")(h, b.ptr, self.out_mem, C.byref(out)))
                    r = _emit(be, out, self.out_mem, names)
                if r is not None:
                    yield r
            if not indices_only:
                out = C.POINTER(abi.Batch)()  # tail, hash_join.rs:296-322
                be.check(be.fn("hash_join_finish")(h, self.out_mem, C.byref(out)))
                r = _emit(be, out, self.out_mem, names)
                if r is not None:
                    yield r
        finally:
            be.fn("hash_join_destroy")(h)


class HashAggExecutor:
    """``HashAggExecutor { agg_funcs, group_by, child }`` (hash_agg.rs:15-19)."""

    def __init__(self, backend: abi.Backend, agg_funcs: List[AggFunc], group_by: List[BoundExpr],
                 child: Iterable, out_mem: int = abi.MEM_HOST,
                 output_names: Optional[Sequence[str]] = None, child_filter: Optional[BoundExpr] = None):
        self.backend, self.agg_funcs, self.group_by = backend, agg_funcs, group_by
        self.child, self.out_mem, self.output_names = child, out_mem, output_names
        # FilterExecutor{expr = child_filter, child} directly below the operator (filter.rs:7-25), handed to the
        # library (sqlrs_hash_agg_set_filter; HIP library only): same result as wrapping `child` in a FilterExecutor
        self.child_filter = child_filter
        self.filter_fused_batches = 0

    def execute(self):
        be = self.backend
        keep = []
        gb, k = abi.pack_exprs(self.group_by)
        keep += [gb, k]
        aggs = (abi.AggFunc * max(len(self.agg_funcs), 1))(
            *[a.abi_struct(keep) for a in self.agg_funcs])
        h = C.c_void_p()
        be.check(be.fn("hash_agg_create")(be.ctx, len(self.group_by), gb, len(self.agg_funcs), aggs,
                                          C.byref(h)))
        try:
            if self.child_filter is not None:
                pf = self.child_filter.pack()
                keep.append(pf)
                be.check(be.fn("hash_agg_set_filter")(h, C.byref(pf.abi)))
            for batch in self.child:  # hash_agg.rs:44-122
                b = abi.as_batch(batch)
                be.check(be.fn("hash_agg_push")(h, b.ptr))
            out = C.POINTER(abi.Batch)()  # hash_agg.rs:124-149: exactly one output batch
            be.check(be.fn("hash_agg_finish")(h, self.out_mem, C.byref(out)))
            if self.child_filter is not None:
                self.filter_fused_batches = be.fn("hash_agg_filter_fused_batches")(h)
            yield _emit(be, out, self.out_mem, self.output_names)
        finally:
            be.fn("hash_agg_destroy")(h)


class HashJoinAggExecutor:
    """``HashAggExecutor`` directly over an Inner ``HashJoinExecutor`` without join filter — the
    physical rewrite of ``PhysicalHashAgg(PhysicalHashJoin(left, right))`` (hash_agg.rs:32-150
    consuming hash_join.rs:146-323).  Same result as running the two operators back to back;
    ``group_by`` / aggregate arguments index the join output schema (left columns, then right
    columns).  HIP library only (the oracle composes the two operators)."""

    def __init__(self, backend: abi.Backend, left_child: Iterable, right_child: Iterable,
                 join_condition: JoinCondition, join_output_schema: pa.Schema, num_left_columns: int,
                 agg_funcs: List[AggFunc], group_by: List[BoundExpr], out_mem: int = abi.MEM_HOST,
                 output_names: Optional[Sequence[str]] = None, probe_filter: Optional[BoundExpr] = None):
        self.backend = backend
        # FilterExecutor{expr = probe_filter, child = right_child} directly below the probe side
        # (filter.rs:7-25): same result as wrapping right_child in a FilterExecutor
        self.probe_filter = probe_filter
        self.filter_fused_batches = 0
        self.left_child, self.right_child = left_child, right_child
        self.join_condition, self.join_output_schema = join_condition, join_output_schema
        self.num_left_columns = num_left_columns
        self.agg_funcs, self.group_by = agg_funcs, group_by
        self.out_mem, self.output_names = out_mem, output_names
        self.fused_batches = 0
        self.eager_groups = 0  # partial groups re-aggregated by build-side GROUP BY columns (0 = route not taken)

    def execute(self):
        be = self.backend
        keep = []
        lk, k1 = abi.pack_exprs([l for l, _ in self.join_condition.on])
        rk, k2 = abi.pack_exprs([r for _, r in self.join_condition.on])
        gb, k3 = abi.pack_exprs(self.group_by)
        keep += [lk, rk, gb, k1, k2, k3]
        aggs = (abi.AggFunc * max(len(self.agg_funcs), 1))(*[a.abi_struct(keep) for a in self.agg_funcs])
        right_fields = list(self.join_output_schema)[self.num_left_columns:]
        rd = (C.c_int32 * max(len(right_fields), 1))(*[abi.dtype_of(f.type) for f in right_fields])
        h = C.c_void_p()
        be.check(be.fn("join_agg_create")(be.ctx, len(self.join_condition.on), lk, rk, self.num_left_columns,
                                          len(right_fields), rd, len(self.group_by), gb, len(self.agg_funcs),
                                          aggs, C.byref(h)))
        try:
            if self.probe_filter is not None:
                pf = self.probe_filter.pack()
                keep.append(pf)
                be.check(be.fn("join_agg_set_probe_filter")(h, C.byref(pf.abi)))
            for batch in self.left_child:
                b = abi.as_batch(batch)  # keep the marshalled batch alive across the call
                be.check(be.fn("join_agg_build_push")(h, b.ptr))
            be.check(be.fn("join_agg_build_finish")(h))
            for batch in self.right_child:
                b = abi.as_batch(batch)
                be.check(be.fn("join_agg_probe_push")(h, b.ptr))
            out = C.POINTER(abi.Batch)()
            be.check(be.fn("join_agg_finish")(h, self.out_mem, C.byref(out)))
            self.fused_batches = be.fn("join_agg_fused_batches")(h)
            self.filter_fused_batches = be.fn("join_agg_filter_fused_batches")(h)
            self.eager_groups = be.fn("join_agg_eager_groups")(h)
            yield _emit(be, out, self.out_mem, self.output_names)
        finally:
            be.fn("join_agg_destroy")(h)


class OrderExecutor:
    """``OrderExecutor { order_by, child }`` (order.rs:8-11).  ``retain_inputs``: the child's DEVICE batches are
    kept alive by this executor until the sort is done instead of being copied by the library (what the reference
    does with its Arc'd arrays, order.rs:19-26; ``sqlrs_order_push_retained``)."""

    def __init__(self, backend: abi.Backend, order_by: List[OrderBy], child: Iterable,
                 out_mem: int = abi.MEM_HOST, retain_inputs: bool = False, limit_hint: Optional[int] = None):
        self.backend, self.order_by, self.child, self.out_mem = backend, order_by, child, out_mem
        self.retain_inputs = retain_inputs
        # LimitExecutor{offset, limit} directly above (PhysicalLimit(PhysicalOrder(child))): only the first
        # offset + limit rows will be read — the operator may return a prefix of the sorted result (sqlrs_order_set_limit)
        self.limit_hint = limit_hint
        self.topk_candidates = 0  # rows the sort actually took because of the hint (0 = all of them)

    def execute(self):
        be = self.backend
        keep = []
        obs = []
        for ob in self.order_by:
            p = ob.expr.pack()
            keep.append(p)
            obs.append(abi.OrderBy(p.abi, int(ob.asc), 0))
        arr = (abi.OrderBy * max(len(obs), 1))(*obs)
        h = C.c_void_p()
        be.check(be.fn("order_create")(be.ctx, len(obs), arr, C.byref(h)))
        names = None
        hinted = self.limit_hint is not None and hasattr(be.lib, be.prefix + "order_set_limit")  # (HIP library only)
        try:
            if hinted:
                be.check(be.fn("order_set_limit")(h, int(self.limit_hint)))
            for batch in self.child:  # order.rs:19-26
                names = names or _names_of(batch)
                b = abi.as_batch(batch)
                if self.retain_inputs:
                    keep.append((batch, b))  # alive and unchanged until order_finish has returned
                    be.check(be.fn("order_push_retained")(h, b.ptr))
                else:
                    be.check(be.fn("order_push")(h, b.ptr))
            out = C.POINTER(abi.Batch)()
            be.check(be.fn("order_finish")(h, self.out_mem, C.byref(out)))
            if hinted:
                self.topk_candidates = be.fn("order_topk_candidates")(h)
            yield _emit(be, out, self.out_mem, names)
        finally:
            be.fn("order_destroy")(h)


class ProjectExecutor:
    """``ProjectExecutor { exprs, child }`` (project.rs:6-9)."""

    def __init__(self, backend: abi.Backend, exprs: List[BoundExpr], child: Iterable, out_mem: int = abi.MEM_HOST,
                 output_names: Optional[Sequence[str]] = None, many: int = 0, depth: int = 0):
        self.backend, self.exprs, self.child, self.out_mem, self.output_names = backend, exprs, child, out_mem, output_names
        self.depth = depth  # > 0: sqlrs_project_push_async with that many tickets in flight (see FilterExecutor)
        # many > 1: that many batches of the child go to sqlrs_project_push_many together (the same stream of output batches,
        # one per input batch, project.rs:15-27)
        self.many = many

    def execute(self):
        be = self.backend
        arr, keep = abi.pack_exprs(self.exprs)
        h = C.c_void_p()
        be.check(be.fn("project_create")(be.ctx, len(self.exprs), arr, C.byref(h)))
        try:
            if self.many > 1 and getattr(be.lib, be.prefix + "project_push_many", None) is not None:
                group = []

                def flush():
                    n = len(group)
                    hb = [abi.as_batch(b) for b in group]
                    ins = (C.POINTER(abi.Batch) * n)(*[C.pointer(b.abi) if hasattr(b, "abi") else b.ptr for b in hb])
                    outs = (C.POINTER(abi.Batch) * n)()
                    be.check(be.fn("project_push_many")(h, n, ins, self.out_mem, outs))
                    res = [_emit(be, outs[i], self.out_mem, self.output_names) for i in range(n)]
                    group.clear()
                    return res
                for batch in self.child:
                    group.append(batch)
                    if len(group) == self.many:
                        yield from flush()
                if group:
                    yield from flush()
                return
            if self.depth > 0 and self.out_mem == abi.MEM_HOST and getattr(be.lib, be.prefix + "project_push_async", None) is not None:
                yield from _async_stream(be, self.child, self.depth, lambda b, t: be.fn("project_push_async")(h, b.ptr, t),
                                         lambda _b: self.output_names)
                return
            for batch in self.child:  # project.rs:14-27
                b = abi.as_batch(batch)
                out = C.POINTER(abi.Batch)()
                be.check(be.fn("project_push")(h, b.ptr, self.out_mem, C.byref(out)))
                yield _emit(be, out, self.out_mem, self.output_names)
        finally:
            be.fn("project_destroy")(h)


class CrossJoinExecutor:
    """``CrossJoinExecutor { left_child, right_child, join_output_schema }`` (cross_join.rs:8-13): what the binder turns an
    uncorrelated scalar subquery into (binder/table/subquery.rs:120-167).  One output batch per (right batch, left row)
    like the reference (cross_join.rs:39-55): the library returns the batches of one right batch as a single batch in
    the same row order, sliced back here (host output only)."""

    def __init__(self, backend: abi.Backend, left_child: Iterable, right_child: Iterable, join_output_schema: pa.Schema,
                 max_rows_per_call: int = (1 << 31) - 1):
        self.backend, self.left_child, self.right_child, self.schema = backend, left_child, right_child, join_output_schema
        self.max_rows_per_call = max_rows_per_call

    def execute(self):
        be = self.backend
        h = C.c_void_p()
        be.check(be.fn("cross_join_create")(be.ctx, C.byref(h)))
        try:
            self._left_rows = 0
            for batch in self.left_child:  # cross_join.rs:30
                b = abi.as_batch(batch)
                self._left_rows += b.abi.num_rows if hasattr(b, "abi") else b.num_rows
                be.check(be.fn("cross_join_build_push")(h, b.ptr))
            for batch in self.right_child:  # cross_join.rs:38-56
                b = abi.as_batch(batch)
                r = b.abi.num_rows if hasattr(b, "abi") else b.num_rows
                # a library batch holds < 2^31 rows: the left rows are taken in ranges of at most that many output rows
                # (one range for anything but a very large product; max_rows_per_call is a test hook)
                per_call = max(1, self.max_rows_per_call // max(r, 1))
                for l0 in range(0, max(self._left_rows, 1), per_call):
                    out = C.POINTER(abi.Batch)()
                    if per_call >= self._left_rows:
                        be.check(be.fn("cross_join_probe_push")(h, b.ptr, abi.MEM_HOST, C.byref(out)))
                    else:
                        be.check(be.fn("cross_join_probe_push_range")(h, b.ptr, l0, min(per_call, self._left_rows - l0), abi.MEM_HOST, C.byref(out)))
                    whole = _emit(be, out, abi.MEM_HOST, list(self.schema.names))
                    if whole is None:
                        continue
                    if r == 0:  # one EMPTY batch per left row (cross_join.rs:39-55 emits it all the same)
                        for _ in range(self._left_rows):
                            yield whole.slice(0, 0)
                        break
                    for i in range(whole.num_rows // r):
                        yield whole.slice(i * r, r)
        finally:
            be.fn("cross_join_destroy")(h)


class LimitExecutor:
    """``LimitExecutor { limit, offset, child }`` (limit.rs:4-8); ``None`` = no LIMIT / OFFSET clause."""

    def __init__(self, backend: abi.Backend, limit: Optional[int], offset: Optional[int], child: Iterable,
                 out_mem: int = abi.MEM_HOST):
        self.backend, self.limit, self.offset, self.child, self.out_mem = backend, limit, offset, child, out_mem

    def execute(self):
        be = self.backend
        h = C.c_void_p()
        be.check(be.fn("limit_create")(be.ctx, int(self.limit is not None), int(self.limit or 0),
                                       int(self.offset is not None), int(self.offset or 0), C.byref(h)))
        try:
            if self.limit is not None and self.limit == 0:
                return  # limit.rs:29-31: returns before the child is polled
            for batch in self.child:  # limit.rs:35-79
                b = abi.as_batch(batch)
                out = C.POINTER(abi.Batch)()
                done = C.c_int(0)
                be.check(be.fn("limit_push")(h, b.ptr, self.out_mem, C.byref(out), C.byref(done)))
                r = _emit(be, out, self.out_mem, _names_of(batch))
                if r is not None:
                    yield r
                if done.value:
                    break
        finally:
            be.fn("limit_destroy")(h)


class SimpleAggExecutor:
    """``SimpleAggExecutor { agg_funcs, child }`` (simple_agg.rs:10-13)."""

    def __init__(self, backend: abi.Backend, agg_funcs: List[AggFunc], child: Iterable, out_mem: int = abi.MEM_HOST,
                 output_names: Optional[Sequence[str]] = None):
        self.backend, self.agg_funcs, self.child, self.out_mem, self.output_names = backend, agg_funcs, child, out_mem, output_names

    def execute(self):
        be = self.backend
        keep = []
        aggs = (abi.AggFunc * max(len(self.agg_funcs), 1))(*[a.abi_struct(keep) for a in self.agg_funcs])
        h = C.c_void_p()
        be.check(be.fn("simple_agg_create")(be.ctx, len(self.agg_funcs), aggs, C.byref(h)))
        try:
            for batch in self.child:  # simple_agg.rs:35-56
                b = abi.as_batch(batch)
                be.check(be.fn("simple_agg_push")(h, b.ptr))
            out = C.POINTER(abi.Batch)()
            be.check(be.fn("simple_agg_finish")(h, self.out_mem, C.byref(out)))
            yield _emit(be, out, self.out_mem, self.output_names)
        finally:
            be.fn("simple_agg_destroy")(h)


class CsvScan:
    """``CsvTable`` + ``CsvTransaction`` (storage/csv.rs:108-241): schema inferred from the first records,
    batches of ``batch_size`` rows; ``bounds = (offset, limit)``, ``projection`` = column indices.
    HIP library only (CSV ingest is host work that lands in HBM with ``out_mem=MEM_DEVICE``)."""

    def __init__(self, backend: abi.Backend, path: str, has_header: bool = True, delimiter: str = ",",
                 batch_size: int = 1024, infer_max_records: int = 10, bounds=None, projection=None,
                 out_mem: int = abi.MEM_HOST):
        self.backend, self.path, self.out_mem = backend, path, out_mem
        self.cfg = (has_header, delimiter, batch_size, infer_max_records)
        self.bounds, self.projection = bounds, projection
        self.names: List[str] = []
        self.dtypes: List[int] = []

    def execute(self):
        be = self.backend
        h = C.c_void_p()
        has_header, delim, bs, infer = self.cfg
        be.check(be.fn("csv_open")(be.ctx, self.path.encode(), int(has_header), delim.encode()[:1], bs, infer, C.byref(h)))
        try:
            n = be.fn("csv_num_columns")(h)
            names = [be.fn("csv_column_name")(h, k).decode().lower() for k in range(n)]  # infer_catalog lowercases (:137)
            dtypes = [be.fn("csv_column_dtype")(h, k) for k in range(n)]
            if self.projection is not None:
                arr = (C.c_int32 * max(len(self.projection), 1))(*self.projection)
                be.check(be.fn("csv_set_projection")(h, len(self.projection), arr))
                names = [names[k] for k in self.projection]
                dtypes = [dtypes[k] for k in self.projection]
            self.names, self.dtypes = names, dtypes
            if self.bounds is not None:
                be.check(be.fn("csv_set_bounds")(h, int(self.bounds[0]), -1 if self.bounds[1] is None else int(self.bounds[1])))
            while True:
                out = C.POINTER(abi.Batch)()
                be.check(be.fn("csv_next_batch")(h, self.out_mem, C.byref(out)))
                if not out:
                    break
                yield _emit(be, out, self.out_mem, names)
        finally:
            be.fn("csv_close")(h)


def eval_column(backend: abi.Backend, expr: BoundExpr, batch, out_mem: int = abi.MEM_HOST):
    """``BoundExpr::eval_column`` (evaluator.rs:13-28) -> one-column batch."""
    packed = expr.pack()
    b = abi.as_batch(batch)
    out = C.POINTER(abi.Batch)()
    backend.check(backend.fn("eval_expr")(backend.ctx, C.byref(packed.abi), b.ptr, out_mem,
                                          C.byref(out)))
    return _emit(backend, out, out_mem, ["expr"])


def try_collect(executor) -> List:
    """``try_collect`` (executor/mod.rs:58-64)."""
    return list(executor.execute())
