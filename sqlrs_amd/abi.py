"""ctypes mirror of include/sqlrs_hip.h plus pyarrow <-> sqlrs_batch_t marshalling.

The same struct layouts are used by the HIP library (prefix ``sqlrs_``) and by the
test-only CPU oracle (prefix ``oracle_``, loaded from ``oracle/`` by the tests, never
from here).  ``Backend`` is parameterised by (shared object, prefix) so that the parity
tests drive both through one code path.

Reference vocabulary: a batch is an ``arrow::record_batch::RecordBatch``
(executor/mod.rs:34), a column an ``ArrayRef``.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, List, Optional, Sequence

import pyarrow as pa

# ---- enums (include/sqlrs_hip.h) -------------------------------------------------
OK, ERR_ARROW, ERR_INTERNAL, ERR_STORAGE, ERR_DEVICE = 0, 1, 2, 3, 4
NULLTYPE, INT32, INT64, FLOAT64, BOOLEAN, UTF8, UINT32, UINT64 = range(8)
MEM_HOST, MEM_DEVICE = 0, 1
GROUP_ORDER_FIRST_SEEN, GROUP_ORDER_ANY = 0, 1
JOIN_INNER, JOIN_LEFT, JOIN_RIGHT, JOIN_FULL = 0, 1, 2, 3
AGG_COUNT, AGG_SUM, AGG_MIN, AGG_MAX = 0, 1, 2, 3

EXPR_INPUT_REF, EXPR_CONSTANT, EXPR_TYPE_CAST = 1, 2, 3
(EXPR_PLUS, EXPR_MINUS, EXPR_MULTIPLY, EXPR_DIVIDE, EXPR_GT, EXPR_LT, EXPR_GTEQ, EXPR_LTEQ,
 EXPR_EQ, EXPR_NOTEQ, EXPR_AND, EXPR_OR) = range(10, 22)

_PA_TO_DTYPE = {
    pa.int32(): INT32, pa.int64(): INT64, pa.float64(): FLOAT64, pa.bool_(): BOOLEAN,
    pa.string(): UTF8, pa.uint32(): UINT32, pa.uint64(): UINT64,
}
_DTYPE_TO_PA = {v: k for k, v in _PA_TO_DTYPE.items()}
_WIDTH = {INT32: 4, INT64: 8, FLOAT64: 8, UINT32: 4, UINT64: 8}


def dtype_of(t: pa.DataType) -> int:
    try:
        return _PA_TO_DTYPE[t]
    except KeyError:
        raise ExecutorError(ERR_INTERNAL, f"unsupported arrow type {t}")


def pa_type(dtype: int) -> pa.DataType:
    return _DTYPE_TO_PA[dtype]


class ExecutorError(RuntimeError):
    """Mirror of ``ExecutorError`` (executor/mod.rs:67-85)."""

    KINDS = {ERR_ARROW: "Arrow", ERR_INTERNAL: "InternalError", ERR_STORAGE: "Storage",
             ERR_DEVICE: "Device"}

    def __init__(self, status: int, message: str):
        super().__init__(f"{self.KINDS.get(status, status)}: {message}")
        self.status = status
        self.message = message


# ---- structs ---------------------------------------------------------------------
class Column(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("mem", C.c_int32), ("length", C.c_int64),
                ("null_count", C.c_int64), ("values", C.c_void_p), ("validity", C.c_void_p),
                ("offsets", C.c_void_p)]


class Batch(C.Structure):
    _fields_ = [("num_rows", C.c_int64), ("num_columns", C.c_int32), ("reserved", C.c_int32),
                ("columns", C.POINTER(Column)), ("owner", C.c_void_p)]


class ExprNode(C.Structure):
    _fields_ = [("op", C.c_int32), ("dtype", C.c_int32), ("index", C.c_int32),
                ("is_null", C.c_int32), ("i", C.c_int64), ("f", C.c_double), ("s", C.c_char_p)]


class Expr(C.Structure):
    _fields_ = [("nodes", C.POINTER(ExprNode)), ("num_nodes", C.c_int32),
                ("reserved", C.c_int32)]


class AggFunc(C.Structure):
    _fields_ = [("func", C.c_int32), ("distinct", C.c_int32), ("return_dtype", C.c_int32),
                ("reserved", C.c_int32), ("arg", Expr)]


class OrderBy(C.Structure):
    _fields_ = [("expr", Expr), ("asc", C.c_int32), ("reserved", C.c_int32)]


# ---- pyarrow -> ABI --------------------------------------------------------------
def _normalise(arr: pa.Array) -> pa.Array:
    if isinstance(arr, pa.ChunkedArray):
        arr = arr.combine_chunks() if arr.num_chunks != 1 else arr.chunk(0)
    if arr.offset != 0:
        # the ABI requires offset 0: re-materialise the slice
        arr = pa.concat_arrays([arr, arr.slice(0, 0)])
        if arr.offset != 0:
            arr = pa.array(arr.to_pylist(), type=arr.type)
    return arr


class HostBatch:
    """A caller-built sqlrs_batch_t over the buffers of a pyarrow RecordBatch (zero copy)."""

    def __init__(self, rb: pa.RecordBatch):
        self.arrays = [_normalise(rb.column(i)) for i in range(rb.num_columns)]
        self.schema = rb.schema
        n = len(self.arrays)
        self.cols = (Column * max(n, 1))()
        for i, arr in enumerate(self.arrays):
            c = self.cols[i]
            c.dtype = dtype_of(arr.type)
            c.mem = MEM_HOST
            c.length = len(arr)
            c.null_count = arr.null_count
            bufs = arr.buffers()
            c.validity = bufs[0].address if (bufs[0] is not None and arr.null_count) else None
            if c.dtype == UTF8:
                c.offsets = bufs[1].address if bufs[1] is not None else None
                c.values = bufs[2].address if bufs[2] is not None else None
                if c.offsets is None:  # zero-length string array without buffers
                    self._z = (C.c_int32 * 1)(0)
                    c.offsets = C.addressof(self._z)
            else:
                c.values = bufs[1].address if bufs[1] is not None else None
        self.abi = Batch(rb.num_rows, n, 0, C.cast(self.cols, C.POINTER(Column)), None)

    @property
    def ptr(self):
        return C.byref(self.abi)


def device_column(dtype: int, length: int, values_ptr: int, validity_ptr: Optional[int] = None,
                  null_count: int = 0) -> Column:
    """Column descriptor over HBM-resident buffers (e.g. torch tensors' data_ptr())."""
    return Column(dtype, MEM_DEVICE, length, null_count if validity_ptr else 0, values_ptr,
                  validity_ptr, None)


class RawBatch:
    """A caller-built sqlrs_batch_t from explicit Column descriptors (host or device)."""

    def __init__(self, columns: Sequence[Column], num_rows: int, keepalive=None):
        n = len(columns)
        self.cols = (Column * max(n, 1))(*columns)
        self.abi = Batch(num_rows, n, 0, C.cast(self.cols, C.POINTER(Column)), None)
        self.keepalive = keepalive

    def release(self):
        """drops the references that keep the caller's buffers alive (same call as LibBatch.release)"""
        self.keepalive = None

    @property
    def ptr(self):
        return C.byref(self.abi)


class ArrowSchemaC(C.Structure):
    pass


class ArrowArrayC(C.Structure):
    pass


ArrowSchemaC._fields_ = [("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p), ("flags", C.c_int64),
                         ("n_children", C.c_int64), ("children", C.POINTER(C.POINTER(ArrowSchemaC))),
                         ("dictionary", C.POINTER(ArrowSchemaC)), ("release", C.c_void_p), ("private_data", C.c_void_p)]
ArrowArrayC._fields_ = [("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64),
                        ("n_children", C.c_int64), ("buffers", C.POINTER(C.c_void_p)),
                        ("children", C.POINTER(C.POINTER(ArrowArrayC))), ("dictionary", C.POINTER(ArrowArrayC)),
                        ("release", C.c_void_p), ("private_data", C.c_void_p)]

# When set to a Backend that has sqlrs_batch_import_arrow / _export_arrow, pyarrow RecordBatches cross the boundary through
# the Arrow C Data Interface (RecordBatch._export_to_c / _import_from_c: what arrow-rs's arrow::ffi does) instead of through
# hand-built sqlrs_column_t descriptors; tests/test_gpu_arrow_c.py replays the reference's goldens that way.
ARROW_C_BACKEND = None


class ArrowCBatch:
    """A pyarrow RecordBatch handed over through the Arrow C Data Interface: pyarrow exports (array, schema), the library
    imports them WITHOUT copying and owns them until release."""

    def __init__(self, backend: "Backend", rb: pa.RecordBatch):
        self.backend = backend
        arr, sch = ArrowArrayC(), ArrowSchemaC()
        rb._export_to_c(C.addressof(arr), C.addressof(sch))
        out = C.POINTER(Batch)()
        st = backend.fn("batch_import_arrow")(backend.ctx, C.byref(arr), C.byref(sch), C.byref(out))
        if st != OK:  # nothing was consumed: give pyarrow's exports back
            pa.RecordBatch._import_from_c(C.addressof(arr), C.addressof(sch))
            backend.check(st)
        assert not arr.release and not sch.release  # moved
        self.p = out

    @property
    def ptr(self):
        return self.p

    @property
    def num_rows(self) -> int:
        return self.p.contents.num_rows

    @property
    def num_columns(self) -> int:
        return self.p.contents.num_columns

    def release(self):
        if self.p is not None:
            self.backend.fn("batch_release")(self.p)
            self.p = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class LibBatch:
    """A batch owned by a library (host or device resident) until released."""

    def __init__(self, backend: "Backend", p):
        self.backend = backend
        self.p = p  # POINTER(Batch)

    @property
    def ptr(self):
        return self.p

    @property
    def num_rows(self) -> int:
        return self.p.contents.num_rows

    @property
    def num_columns(self) -> int:
        return self.p.contents.num_columns

    def column(self, i: int) -> Column:
        return self.p.contents.columns[i]

    def to_arrow(self, names: Optional[Sequence[str]] = None) -> pa.RecordBatch:
        """Copies a HOST resident library batch into a pyarrow RecordBatch."""
        if ARROW_C_BACKEND is not None and self.backend is ARROW_C_BACKEND:
            # the batch is MOVED into the exported structures; pyarrow's import takes them over (zero copy) and a final
            # copy detaches the result from this library's memory like the path below does
            arr, sch = ArrowArrayC(), ArrowSchemaC()
            nm = None
            if names is not None:
                enc = [n.encode() for n in names]
                nm = (C.c_char_p * len(enc))(*enc)
            p, self.p = self.p, None
            self.backend.check(self.backend.fn("batch_export_arrow")(self.backend.ctx, p, nm, C.byref(arr), C.byref(sch)))
            return pa.RecordBatch._import_from_c(C.addressof(arr), C.addressof(sch))
        b = self.p.contents
        arrays = []
        for i in range(b.num_columns):
            c = b.columns[i]
            if c.mem != MEM_HOST:
                raise ExecutorError(ERR_INTERNAL, "to_arrow on a device batch: copy it to host first")
            n = c.length
            nb = (n + 7) // 8
            validity = None
            if c.validity and c.null_count != 0:
                validity = pa.py_buffer(C.string_at(c.validity, nb))
            t = pa_type(c.dtype)
            if c.dtype == UTF8:
                offs = C.string_at(c.offsets, 4 * (n + 1))
                end = int.from_bytes(offs[-4:], "little") if n >= 0 else 0
                data = C.string_at(c.values, end) if end else b""
                arr = pa.Array.from_buffers(t, n, [validity, pa.py_buffer(offs), pa.py_buffer(data)])
            elif c.dtype == BOOLEAN:
                arr = pa.Array.from_buffers(t, n, [validity, pa.py_buffer(C.string_at(c.values, nb))])
            else:
                arr = pa.Array.from_buffers(
                    t, n, [validity, pa.py_buffer(C.string_at(c.values, _WIDTH[c.dtype] * n))])
            arrays.append(arr)
        if names is None:
            names = [f"c{i}" for i in range(len(arrays))]
        return pa.RecordBatch.from_arrays(arrays, names=list(names))

    def release(self):
        if self.p is not None:
            self.backend.fn("batch_release")(self.p)
            self.p = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def as_batch(x):
    """pyarrow RecordBatch / HostBatch / RawBatch / LibBatch -> object with .ptr"""
    if isinstance(x, pa.RecordBatch):
        return ArrowCBatch(ARROW_C_BACKEND, x) if ARROW_C_BACKEND is not None else HostBatch(x)
    if isinstance(x, pa.Table):
        return HostBatch(x.combine_chunks().to_batches()[0] if x.num_rows else
                         pa.RecordBatch.from_pylist([], schema=x.schema))
    return x


# ---- expressions -----------------------------------------------------------------
class PackedExpr:
    """Owns the node array behind one sqlrs_expr_t."""

    def __init__(self, nodes: List[ExprNode]):
        self.arr = (ExprNode * len(nodes))(*nodes)
        self.abi = Expr(C.cast(self.arr, C.POINTER(ExprNode)), len(nodes), 0)


def pack_exprs(exprs: Iterable) -> "tuple[C.Array, list]":
    packed = [e.pack() for e in exprs]
    arr = (Expr * max(len(packed), 1))(*[p.abi for p in packed])
    return arr, packed


# ---- backend ---------------------------------------------------------------------
class Backend:
    """One loaded implementation of the ABI (HIP library or, in tests, the CPU oracle)."""

    def __init__(self, lib_path: str, prefix: str, ctx_arg: int = 0):
        self.lib = C.CDLL(lib_path)
        self.prefix = prefix
        self.lib_path = lib_path
        self._declare()
        ctx = C.c_void_p()
        st = self.fn("ctx_create")(ctx_arg, C.byref(ctx))
        if st != OK:
            raise ExecutorError(st, f"{prefix}ctx_create({ctx_arg}) failed: no usable device / library")
        self.ctx = ctx

    def fn(self, name: str):
        return getattr(self.lib, self.prefix + name)

    def _declare(self):
        vp, pvp, i = C.c_void_p, C.POINTER(C.c_void_p), C.c_int
        pb, ppb = C.POINTER(Batch), C.POINTER(C.POINTER(Batch))
        pe = C.POINTER(Expr)
        sig = {
            "ctx_create": (i, [i, pvp]),
            "ctx_destroy": (None, [vp]),
            "last_error": (C.c_char_p, [vp]),
            "batch_release": (None, [pb]),
            "filter_create": (i, [vp, pe, pvp]),
            "filter_push": (i, [vp, pb, i, ppb]),
            "filter_destroy": (None, [vp]),
            "eval_expr": (i, [vp, pe, pb, i, ppb]),
            "hash_join_create": (i, [vp, i, i, pe, pe, pe, i, C.POINTER(C.c_int32), pvp]),
            "hash_join_build_push": (i, [vp, pb]),
            "hash_join_build_finish": (i, [vp]),
            "hash_join_probe_push": (i, [vp, pb, i, ppb]),
            "hash_join_probe_indices": (i, [vp, pb, i, ppb]),
            "hash_join_finish": (i, [vp, i, ppb]),
            "hash_join_destroy": (None, [vp]),
            "hash_agg_create": (i, [vp, i, pe, i, C.POINTER(AggFunc), pvp]),
            "hash_agg_push": (i, [vp, pb]),
            "hash_agg_finish": (i, [vp, i, ppb]),
            "hash_agg_destroy": (None, [vp]),
            "order_create": (i, [vp, i, C.POINTER(OrderBy), pvp]),
            "order_push": (i, [vp, pb]),
            "order_finish": (i, [vp, i, ppb]),
            "order_destroy": (None, [vp]),
            "version": (C.c_char_p, []),
        }
        for name, (res, args) in sig.items():
            f = self.fn(name)
            f.restype = res
            f.argtypes = args
        # HIP-only entry points
        optional = {
            "ctx_synchronize": (i, [vp]),
            "ctx_stream": (vp, [vp]),
            "ctx_wait_stream": (i, [vp, vp]),
            "ctx_release_to_stream": (i, [vp, vp]),
            "order_push_retained": (i, [vp, pb]),
            "order_set_limit": (i, [vp, C.c_int64]),
            "order_topk_candidates": (C.c_int64, [vp]),
            "ctx_pool_bytes": (C.c_int64, [vp]),
            "ctx_pool_trim": (None, [vp]),
            "batch_copy": (i, [vp, pb, i, ppb]),
            "filter_push_async": (i, [vp, pb, pvp]),
            "hash_join_probe_push_async": (i, [vp, pb, pvp]),
            "project_push_async": (i, [vp, pb, pvp]),
            "batch_wait": (i, [vp, ppb]),
            "batch_import_arrow": (i, [vp, C.POINTER(ArrowArrayC), C.POINTER(ArrowSchemaC), ppb]),
            "batch_export_arrow": (i, [vp, pb, C.POINTER(C.c_char_p), C.POINTER(ArrowArrayC), C.POINTER(ArrowSchemaC)]),
            "exchange_unique_id": (i, [vp, vp]),
            "exchange_create": (i, [vp, vp, i, i, pvp]),
            "exchange_all_to_all": (i, [vp, pb, C.POINTER(C.c_int64), C.POINTER(C.c_int64), ppb, C.POINTER(C.c_int64)]),
            "exchange_begin": (i, [vp, i, C.POINTER(C.c_int32), C.c_int64]),
            "exchange_send_chunk": (i, [vp, pb, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
            "exchange_finish": (i, [vp, ppb]),
            "exchange_plan": (i, [i, i, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
            "exchange_bytes_off_rank": (C.c_int64, [vp]),
            "exchange_destroy": (None, [vp]),
            "hash_partition": (i, [vp, pb, pe, i, i, ppb, C.POINTER(C.c_int64)]),
            "hash_partition_filter": (i, [vp, pb, pe, pe, i, i, ppb, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
            "join_agg_create": (i, [vp, i, pe, pe, i, i, C.POINTER(C.c_int32), i, pe, i, C.POINTER(AggFunc), pvp]),
            "join_agg_build_push": (i, [vp, pb]),
            "join_agg_build_finish": (i, [vp]),
            "join_agg_probe_push": (i, [vp, pb]),
            "join_agg_finish": (i, [vp, i, ppb]),
            "join_agg_fused_batches": (C.c_int64, [vp]),
            "join_agg_set_probe_filter": (i, [vp, pe]),
            "join_agg_filter_fused_batches": (C.c_int64, [vp]),
            "join_agg_eager_groups": (C.c_int64, [vp]),
            "join_agg_set_group_order": (i, [vp, i]),
            "hash_agg_set_group_order": (i, [vp, i]),
            "hash_agg_set_filter": (i, [vp, pe]),
            "hash_agg_filter_fused_batches": (C.c_int64, [vp]),
            "join_agg_destroy": (None, [vp]),
            "filter_push_many": (i, [vp, i, C.POINTER(pb), i, C.POINTER(pb)]),
            "hash_join_probe_push_many": (i, [vp, i, C.POINTER(pb), i, C.POINTER(pb)]),
            "cross_join_create": (i, [vp, pvp]),
            "cross_join_build_push": (i, [vp, pb]),
            "cross_join_probe_push": (i, [vp, pb, i, ppb]),
            "cross_join_probe_push_range": (i, [vp, pb, C.c_int64, C.c_int64, i, ppb]),
            "cross_join_left_rows": (C.c_int64, [vp]),
            "cross_join_destroy": (None, [vp]),
            "project_create": (i, [vp, i, pe, pvp]),
            "project_push": (i, [vp, pb, i, ppb]),
            "project_push_many": (i, [vp, i, C.POINTER(pb), i, C.POINTER(pb)]),
            "project_destroy": (None, [vp]),
            "limit_create": (i, [vp, i, C.c_int64, i, C.c_int64, pvp]),
            "limit_push": (i, [vp, pb, i, ppb, C.POINTER(C.c_int)]),
            "limit_destroy": (None, [vp]),
            "simple_agg_create": (i, [vp, i, C.POINTER(AggFunc), pvp]),
            "simple_agg_push": (i, [vp, pb]),
            "simple_agg_finish": (i, [vp, i, ppb]),
            "simple_agg_destroy": (None, [vp]),
            "batch_to_string": (i, [vp, pb, C.POINTER(C.c_void_p)]),
            "string_free": (None, [C.c_void_p]),
            "csv_open": (i, [vp, C.c_char_p, i, C.c_char, C.c_int64, C.c_int64, pvp]),
            "csv_num_columns": (i, [vp]),
            "csv_column_name": (C.c_char_p, [vp, i]),
            "csv_column_dtype": (i, [vp, i]),
            "csv_set_bounds": (i, [vp, C.c_int64, C.c_int64]),
            "csv_set_projection": (i, [vp, i, C.POINTER(C.c_int32)]),
            "csv_next_batch": (i, [vp, i, ppb]),
            "csv_close": (None, [vp]),
            "timer_create": (i, [vp, pvp]),
            "timer_start": (i, [vp]),
            "timer_stop": (i, [vp]),
            "timer_elapsed_ms": (i, [vp, C.POINTER(C.c_double)]),
            "timer_destroy": (None, [vp]),
            "ctx_profile_enable": (i, [vp, i]),
            "ctx_profile_reset": (i, [vp]),
            "ctx_profile_read": (i, [vp, i, C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                                     C.POINTER(C.c_int64)]),
        }
        for name, (res, args) in optional.items():
            f = getattr(self.lib, self.prefix + name, None)
            if f is not None:
                f.restype = res
                f.argtypes = args

    def check(self, status: int):
        if status != OK:
            msg = self.fn("last_error")(self.ctx)
            raise ExecutorError(status, (msg or b"").decode("utf-8", "replace"))

    def wrap(self, p) -> Optional[LibBatch]:
        return LibBatch(self, p) if p else None

    def version(self) -> str:
        return self.fn("version")().decode()

    def close(self):
        if self.ctx is not None:
            self.fn("ctx_destroy")(self.ctx)
            self.ctx = None

    # ---- helpers on top of optional entry points
    def synchronize(self):
        self.check(self.fn("ctx_synchronize")(self.ctx))

    def copy(self, batch, out_mem: int) -> LibBatch:
        b = as_batch(batch)
        out = C.POINTER(Batch)()
        self.check(self.fn("batch_copy")(self.ctx, b.ptr, out_mem, C.byref(out)))
        return self.wrap(out)

    def to_device(self, batch) -> LibBatch:
        return self.copy(batch, MEM_DEVICE)

    def to_host(self, batch) -> LibBatch:
        return self.copy(batch, MEM_HOST)

    def hash_partition(self, batch, key_expr, num_parts: int, out_mem: int = MEM_DEVICE):
        """-> (LibBatch permuted by partition, offsets list of num_parts + 1 ints)"""
        b = as_batch(batch)
        packed = key_expr.pack()
        out = C.POINTER(Batch)()
        offs = (C.c_int64 * (num_parts + 1))()
        self.check(self.fn("hash_partition")(self.ctx, b.ptr, C.byref(packed.abi), num_parts, out_mem,
                                             C.byref(out), offs))
        return self.wrap(out), list(offs)

    def hash_partition_filter(self, batch, key_expr, predicate, num_parts: int, out_mem: int = MEM_DEVICE):
        """Filter + hash partition in one pass -> (LibBatch, part_start list, part_rows list): partition p =
        rows [part_start[p], part_start[p] + part_rows[p]) of the batch; `predicate` may be None"""
        b = as_batch(batch)
        packed = key_expr.pack()
        pred = predicate.pack() if predicate is not None else None
        out = C.POINTER(Batch)()
        starts = (C.c_int64 * num_parts)()
        rows = (C.c_int64 * num_parts)()
        self.check(self.fn("hash_partition_filter")(self.ctx, b.ptr, C.byref(packed.abi),
                                                    C.byref(pred.abi) if pred is not None else None, num_parts,
                                                    out_mem, C.byref(out), starts, rows))
        return self.wrap(out), list(starts), list(rows)

    # ---- exchange (RCCL all-to-all of hash partitions behind the C ABI; one process per GPU)
    EXCHANGE_ID_BYTES = 128

    def exchange_unique_id(self) -> bytes:
        """the 128-byte id ONE rank makes (ncclGetUniqueId) and hands to the others out of band"""
        buf = C.create_string_buffer(self.EXCHANGE_ID_BYTES)
        self.check(self.fn("exchange_unique_id")(self.ctx, buf))
        return buf.raw

    def exchange_create(self, unique_id: bytes, rank: int, world: int):
        """collective over the ranks that share `unique_id` -> opaque handle (exchange_destroy it)"""
        assert len(unique_id) == self.EXCHANGE_ID_BYTES
        h = C.c_void_p()
        self.check(self.fn("exchange_create")(self.ctx, C.create_string_buffer(unique_id, self.EXCHANGE_ID_BYTES), rank, world, C.byref(h)))
        return h

    def exchange_all_to_all(self, h, parts, part_start, part_rows):
        """rows [part_start[p], + part_rows[p]) of the DEVICE batch `parts` go to rank p -> (LibBatch of the rows received
        from rank 0, 1, ..., rows received per rank)"""
        b = as_batch(parts)
        w = len(part_rows)
        ps, pr, rr = (C.c_int64 * w)(*part_start), (C.c_int64 * w)(*part_rows), (C.c_int64 * w)()
        out = C.POINTER(Batch)()
        self.check(self.fn("exchange_all_to_all")(h, b.ptr, ps, pr, C.byref(out), rr))
        return self.wrap(out), list(rr)

    def exchange_begin(self, h, dtypes, capacity_rows: int = 0):
        """opens a chunk sequence: columns of `dtypes`, one received batch at the end (sqlrs_exchange_begin)"""
        arr = (C.c_int32 * len(dtypes))(*dtypes)
        self.check(self.fn("exchange_begin")(h, len(dtypes), arr, int(capacity_rows)))

    def exchange_send_chunk(self, h, parts, part_start, part_rows):
        """collective: rows [part_start[p], + part_rows[p]) of the DEVICE batch `parts` are this chunk's share for rank p"""
        b = as_batch(parts)
        w = len(part_rows)
        ps, pr = (C.c_int64 * w)(*[int(v) for v in part_start]), (C.c_int64 * w)(*[int(v) for v in part_rows])
        self.check(self.fn("exchange_send_chunk")(h, b.ptr, ps, pr))

    def exchange_finish(self, h):
        """-> LibBatch (DEVICE) of everything received since exchange_begin"""
        out = C.POINTER(Batch)()
        self.check(self.fn("exchange_finish")(h, C.byref(out)))
        return self.wrap(out)

    def exchange_plan(self, world: int, rank: int, send_rows_all):
        """host arithmetic only: send_rows_all[q][p] = rows rank q sends to rank p -> (recv_rows, recv_start, total)"""
        flat = (C.c_int64 * (world * world))(*[int(v) for row in send_rows_all for v in row])
        rr, rs, tot = (C.c_int64 * world)(), (C.c_int64 * world)(), C.c_int64()
        st = self.fn("exchange_plan")(world, rank, flat, rr, rs, C.byref(tot))
        if st != OK:
            raise ExecutorError(st, "exchange_plan: inconsistent arguments")
        return list(rr), list(rs), tot.value

    def batch_to_string(self, batch) -> str:
        """``record_batch_to_string`` (util/mod.rs:53-80) of a pyarrow / host / device batch"""
        b = as_batch(batch)
        out = C.c_void_p()
        self.check(self.fn("batch_to_string")(self.ctx, b.ptr, C.byref(out)))
        try:
            return C.string_at(out.value).decode("utf-8")
        finally:
            self.fn("string_free")(out)

    def profile(self, on: bool = True):
        self.check(self.fn("ctx_profile_enable")(self.ctx, int(on)))
        self.check(self.fn("ctx_profile_reset")(self.ctx))

    def profile_read(self) -> dict:
        cap = 64
        names = (C.c_char_p * cap)()
        ms = (C.c_double * cap)()
        n_l = (C.c_int64 * cap)()
        n = self.fn("ctx_profile_read")(self.ctx, cap, names, ms, n_l)
        return {names[k].decode(): (ms[k], n_l[k]) for k in range(min(n, cap))}
