/*
 * sqlrs_hip.h — C ABI of the MI355X (gfx950) execution backend for the sqlrs
 * `src/executor` hot path: Filter -> HashJoin (build + probe) -> HashAgg
 * (update + finalize) -> Order.
 *
 * This is the drop-in boundary.  Every entry point replaces one
 * piece of a reference operator; the reference interface it stands in for is
 * cited as  [ref: file:line]  relative to the sqlrs source tree.
 *
 * Data model = Arrow column buffers, passed as plain pointers + sizes:
 *   - fixed-width columns: `values` = length * sizeof(T) little-endian
 *   - BOOLEAN columns:     `values` = Arrow bitmap (LSB-first), ceil(length/8) B
 *   - UTF8 columns:        `values` = bytes, `offsets` = (length+1) int32
 *   - `validity`           = Arrow validity bitmap (LSB-first) or NULL (= all
 *                            valid); array offset must be 0 (slice on the host)
 *   - `mem`                = where the three pointers live (host or HBM)
 * An arrow-rs `ArrayData` maps 1:1 (buffers()[i].as_ptr(), null_buffer()); the
 * Rust shim a sqlrs maintainer would add is shown in INTEGRATION.md.
 *
 * Threading: one ctx = one HIP stream on one GPU; calls on a ctx are not
 * re-entrant (the reference executor is single-threaded: executor/mod.rs:58-64).
 * Distinct ctxs may be used from distinct threads / processes / GPUs.
 *
 * No CPU fallback exists in this library: without a usable gfx950 device
 * sqlrs_ctx_create fails with SQLRS_ERR_DEVICE.
 *
 * Environment.  Three deployment settings are read when a ctx / operator is created: SQLRS_POOL_RESERVE_GB (one up-front
 * device allocation that the ctx pool carves its blocks from), SQLRS_POOL_VMM=<MiB> (pool blocks of that size and more are
 * backed through hipMemCreate / hipMemMap), SQLRS_JOIN_COMPOSITE=1 (several integer join keys compared exactly instead of
 * by the reference's combined hash).  Every other SQLRS_* name (tools/HOOKS.md) is a test / tuning hook and is looked at
 * only in a process started with SQLRS_HOOKS=1; without it the library never calls getenv on an operator's path.
 */
#ifndef SQLRS_HIP_H
#define SQLRS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- status -- */
/* [ref: src/executor/mod.rs:67-85  enum ExecutorError {Storage, Arrow, InternalError}] */
typedef enum sqlrs_status {
  SQLRS_OK = 0,
  SQLRS_ERR_ARROW = 1,    /* ExecutorError::Arrow          (bad buffers, divide by zero, ...) */
  SQLRS_ERR_INTERNAL = 2, /* ExecutorError::InternalError  (unsupported dtype / shape)        */
  SQLRS_ERR_STORAGE = 3,  /* ExecutorError::Storage        (never produced by this path)      */
  SQLRS_ERR_DEVICE = 4    /* HIP runtime failure / no gfx950 device (no reference analogue)   */
} sqlrs_status_t;

/* ----------------------------------------------------------------- types -- */
/* [ref: src/types/mod.rs:23-36  ScalarValue lattice: Null/Boolean/Float64/Int32/Int64/String]
 * UINT32/UINT64 exist only for the join index pair arrays (hash_join.rs:218-219). */
typedef enum sqlrs_dtype {
  SQLRS_NULLTYPE = 0,
  SQLRS_INT32 = 1,
  SQLRS_INT64 = 2,
  SQLRS_FLOAT64 = 3,
  SQLRS_BOOLEAN = 4,
  SQLRS_UTF8 = 5,
  SQLRS_UINT32 = 6,
  SQLRS_UINT64 = 7
} sqlrs_dtype_t;

typedef enum sqlrs_mem { SQLRS_MEM_HOST = 0, SQLRS_MEM_DEVICE = 1 } sqlrs_mem_t;

/* One Arrow array.  [ref: arrow::array::ArrayRef as used by executor/evaluator.rs:13-28] */
typedef struct sqlrs_column {
  int32_t dtype;           /* sqlrs_dtype_t */
  int32_t mem;             /* sqlrs_mem_t: address space of values/validity/offsets */
  int64_t length;          /* rows */
  int64_t null_count;      /* -1 = unknown (library counts bits when it needs to) */
  const void *values;
  const uint8_t *validity; /* NULL = no nulls */
  const int32_t *offsets;  /* UTF8 only */
} sqlrs_column_t;

/* One RecordBatch.  [ref: arrow::record_batch::RecordBatch, the item type of
 * BoxedExecutor, executor/mod.rs:34].  Batches returned by the library are
 * owned by it until sqlrs_batch_release.  A DEVICE batch returned by the library
 * may be pushed, unchanged, into another operator of the same ctx: an operator
 * that retains its input (join build side, HashAgg) then shares the batch's
 * reference-counted buffers instead of copying them, and sqlrs_batch_release
 * only drops the caller's reference (the Arc<RecordBatch> clone of the reference).
 *
 * Borrowing rules for caller-built batches:
 *   - SQLRS_MEM_HOST columns are read completely before the call returns.
 *   - SQLRS_MEM_DEVICE columns are used STREAM-ORDERED on the ctx stream: the call queues kernels that read
 *     them (and operators that retain input queue a private copy) but does not wait for those kernels.
 *     The buffers must (a) hold their data before the ctx stream reaches the call — data produced on
 *     another stream: sqlrs_ctx_wait_stream(ctx, that_stream) first, or synchronise that stream — and
 *     (b) stay valid and unmodified until the work queued by the call has completed:
 *     sqlrs_ctx_synchronize, or sqlrs_ctx_release_to_stream(ctx, s) before stream s reuses / frees them.
 *     (A stream-ordered allocator on another stream — e.g. torch's caching allocator — may hand a freed
 *     block to later work of ITS stream: order that stream behind the ctx with release_to_stream.)
 *   - Device bitmaps (validity, BOOLEAN values) must be 8-byte aligned and readable up to the next multiple
 *     of 8 bytes (kernels read whole 64-bit words); host bitmaps have no such requirement.
 * Size limit, the same for every operator: 0 <= num_rows < 2^31 per batch (row ids travel as 32-bit words); a batch
 * outside that range is refused by every call that takes one with SQLRS_ERR_ARROW, before any column is read.  Larger
 * inputs arrive as several batches — the reference's own batches are 1024 rows (src/storage/csv.rs:105). */
typedef struct sqlrs_batch {
  int64_t num_rows;
  int32_t num_columns;
  int32_t reserved;
  sqlrs_column_t *columns;
  void *owner; /* private; NULL for caller-built batches */
} sqlrs_batch_t;

/* ----------------------------------------------------------- expressions -- */
/* Postfix encoding of the BoundExpr subset evaluated on this path.
 * [ref: src/executor/evaluator.rs:13-28 (eval_column), src/executor/array_compute.rs:70-90
 *  (binary_op), src/binder/expression/mod.rs:18-27 (BoundExpr)] */
typedef enum sqlrs_expr_op {
  SQLRS_EXPR_INPUT_REF = 1, /* BoundExpr::InputRef  -> column `index`                       */
  SQLRS_EXPR_CONSTANT = 2,  /* BoundExpr::Constant  -> dtype + (i | f | s), is_null         */
  SQLRS_EXPR_TYPE_CAST = 3, /* BoundExpr::TypeCast  -> cast top of stack to `dtype`         */
  /* BoundExpr::BinaryOp, sqlparser BinaryOperator names (array_compute.rs:75-88) */
  SQLRS_EXPR_PLUS = 10,
  SQLRS_EXPR_MINUS = 11,
  SQLRS_EXPR_MULTIPLY = 12,
  SQLRS_EXPR_DIVIDE = 13,
  SQLRS_EXPR_GT = 14,
  SQLRS_EXPR_LT = 15,
  SQLRS_EXPR_GTEQ = 16,
  SQLRS_EXPR_LTEQ = 17,
  SQLRS_EXPR_EQ = 18,
  SQLRS_EXPR_NOTEQ = 19,
  SQLRS_EXPR_AND = 20, /* and_kleene */
  SQLRS_EXPR_OR = 21   /* or_kleene  */
} sqlrs_expr_op_t;

typedef struct sqlrs_expr_node {
  int32_t op;      /* sqlrs_expr_op_t */
  int32_t dtype;   /* CONSTANT: value type; TYPE_CAST: target type */
  int32_t index;   /* INPUT_REF: column index in the input batch */
  int32_t is_null; /* CONSTANT: ScalarValue::X(None) */
  int64_t i;       /* CONSTANT int32/int64/boolean payload */
  double f;        /* CONSTANT float64 payload */
  const char *s;   /* CONSTANT utf8 payload (NUL-terminated), else NULL */
} sqlrs_expr_node_t;

typedef struct sqlrs_expr {
  const sqlrs_expr_node_t *nodes; /* postfix order */
  int32_t num_nodes;
  int32_t reserved;
} sqlrs_expr_t;

/* [ref: src/binder/table/join.rs:18-24  enum JoinType] (Cross never reaches HashJoin:
 *  optimizer/physical_rewriter.rs:20-31) */
typedef enum sqlrs_join_type {
  SQLRS_JOIN_INNER = 0,
  SQLRS_JOIN_LEFT = 1,
  SQLRS_JOIN_RIGHT = 2,
  SQLRS_JOIN_FULL = 3
} sqlrs_join_type_t;

/* [ref: src/binder/expression/agg_func.rs:10-15 AggFunc, :29-34 BoundAggFunc] */
typedef enum sqlrs_agg_kind {
  SQLRS_AGG_COUNT = 0,
  SQLRS_AGG_SUM = 1,
  SQLRS_AGG_MIN = 2,
  SQLRS_AGG_MAX = 3
} sqlrs_agg_kind_t;

typedef struct sqlrs_agg_func {
  int32_t func;         /* sqlrs_agg_kind_t */
  int32_t distinct;     /* BoundAggFunc.distinct */
  int32_t return_dtype; /* BoundAggFunc.return_type (SUM accumulates in this type, sum.rs:54) */
  int32_t reserved;
  sqlrs_expr_t arg;     /* BoundAggFunc.exprs[0]  (only exprs[0] is read: hash_agg.rs:65) */
} sqlrs_agg_func_t;

/* [ref: src/binder/statement/mod.rs:26-29  BoundOrderBy {expr, asc}] */
typedef struct sqlrs_order_by {
  sqlrs_expr_t expr;
  int32_t asc;
  int32_t reserved;
} sqlrs_order_by_t;

/* ---------------------------------------------------------------- context -- */
typedef struct sqlrs_ctx sqlrs_ctx_t;

/* Opens GPU `device_id`, creates the ctx stream and the device memory pool.
 * Replaces nothing in the reference (it has no device); it is the handle a
 * replacement ExecutorBuilder would hold [ref: src/executor/mod.rs:36-56]. */
int sqlrs_ctx_create(int device_id, sqlrs_ctx_t **out);
/* Operators (and timers) of the ctx must be destroyed first; batches the ctx returned may be released
 * before or after (their device blocks then go straight back to the driver). */
void sqlrs_ctx_destroy(sqlrs_ctx_t *ctx);
/* Message of the last failed call on this ctx (valid until the next call). */
const char *sqlrs_last_error(const sqlrs_ctx_t *ctx);
/* Blocks until all work queued on the ctx stream is done. */
int sqlrs_ctx_synchronize(sqlrs_ctx_t *ctx);
/* The hipStream_t of the ctx as an opaque pointer (event timing, stream ordering by callers). */
void *sqlrs_ctx_stream(sqlrs_ctx_t *ctx);
/* Stream-ordered hand-over of DEVICE buffers without blocking the host (`stream` = a hipStream_t):
 * wait_stream: work queued on the ctx AFTER this call starts only when everything queued on
 *   `producer_stream` BEFORE it has finished (inputs produced there are complete);
 * release_to_stream: work queued on `consumer_stream` AFTER this call starts only when everything queued
 *   on the ctx BEFORE it has finished (borrowed inputs may be reused, outputs may be read there). */
int sqlrs_ctx_wait_stream(sqlrs_ctx_t *ctx, void *producer_stream);
int sqlrs_ctx_release_to_stream(sqlrs_ctx_t *ctx, void *consumer_stream);
/* Bytes currently held by the ctx memory pool (live + cached). */
int64_t sqlrs_ctx_pool_bytes(const sqlrs_ctx_t *ctx);
/* Frees cached (not live) pool blocks back to the driver. */
void sqlrs_ctx_pool_trim(sqlrs_ctx_t *ctx);

/* Releases a batch returned by any call below (host or device resident). */
void sqlrs_batch_release(sqlrs_batch_t *batch);
/* Copies a batch (host or device) into freshly allocated memory of `out_mem`;
 * this is the "Arrow column buffers move to HBM once per pipeline" step. */
int sqlrs_batch_copy(sqlrs_ctx_t *ctx, const sqlrs_batch_t *in, int out_mem, sqlrs_batch_t **out);

/* --------------------------------------------------- Arrow C Data Interface -- */
/* [ref: src/executor/mod.rs:34  BoxedExecutor = BoxStream<'static, Result<RecordBatch, ExecutorError>> — the item every
 *  operator of the reference consumes and yields is an arrow RecordBatch]  The two structs of the Arrow C Data Interface,
 * exactly as the specification defines them (the guard is the specification's own, so this header coexists with
 * arrow/c/abi.h); arrow-rs binds them as arrow::ffi::{FFI_ArrowArray, FFI_ArrowSchema}, pyarrow as _export_to_c /
 * _import_from_c. */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
#define ARROW_FLAG_DICTIONARY_ORDERED 1
#define ARROW_FLAG_NULLABLE 2
#define ARROW_FLAG_MAP_KEYS_SORTED 4
struct ArrowSchema {
  const char *format;
  const char *name;
  const char *metadata;
  int64_t flags;
  int64_t n_children;
  struct ArrowSchema **children;
  struct ArrowSchema *dictionary;
  void (*release)(struct ArrowSchema *);
  void *private_data;
};
struct ArrowArray {
  int64_t length;
  int64_t null_count;
  int64_t offset;
  int64_t n_buffers;
  int64_t n_children;
  const void **buffers;
  struct ArrowArray **children;
  struct ArrowArray *dictionary;
  void (*release)(struct ArrowArray *);
  void *private_data;
};
#endif
/* A RecordBatch exported as a struct array ("+s": what RecordBatch::into / pyarrow's RecordBatch._export_to_c produce) ->
 * a HOST batch of this library WITHOUT copying its buffers: both structures are MOVED into the call (their `release`
 * fields are NULL afterwards, as the specification's move rule says; the schema is released before the call returns, the
 * array when the batch is: sqlrs_batch_release).  Children of type int32 "i", int64 "l", float64 "g", bool "b", utf8 "u";
 * anything else, a dictionary, or NULLs at the struct level: SQLRS_ERR_ARROW and nothing is consumed.  A child whose
 * offset is not a multiple of eight has its bitmaps re-packed (the only copy). */
int sqlrs_batch_import_arrow(sqlrs_ctx_t *ctx, struct ArrowArray *array, struct ArrowSchema *schema, sqlrs_batch_t **out);
/* The reverse: `batch` (produced by this library; HOST or DEVICE resident — device columns are downloaded first) is MOVED
 * into `out_array` / `out_schema` and must not be used or released by the caller afterwards: the consumer's `release`
 * callbacks give the memory back.  `names`: num_columns field names or NULL ("c0", "c1", ...).  A caller-built batch
 * (owner == NULL) is copied, since its buffers are only borrowed. */
int sqlrs_batch_export_arrow(sqlrs_ctx_t *ctx, sqlrs_batch_t *batch, const char *const *names, struct ArrowArray *out_array,
                             struct ArrowSchema *out_schema);

/* ----------------------------------------------------------------- Filter -- */
/* [ref: src/executor/filter.rs:7-25  FilterExecutor{expr, child}::execute]
 * One output batch per input batch, row order preserved, rows whose predicate
 * is false or NULL dropped, empty batches still returned. */
typedef struct sqlrs_filter sqlrs_filter_t;
int sqlrs_filter_create(sqlrs_ctx_t *ctx, const sqlrs_expr_t *expr, sqlrs_filter_t **out);
int sqlrs_filter_push(sqlrs_filter_t *f, const sqlrs_batch_t *in, int out_mem, sqlrs_batch_t **out);
/* n iterations of the loop in ONE call: out[i] (n entries) is what sqlrs_filter_push(in[i]) returns — one output batch
 * per input batch, in order, empty ones included [ref: filter.rs:15-24] — for the reference's batch shape, 1024-row
 * HOST batches [ref: src/storage/csv.rs:105]: small HOST batches of fixed-width columns are uploaded together, filtered
 * by one launch sequence and the kept rows of every input batch handed out as a batch of its own (out_mem = HOST); any
 * other input runs through sqlrs_filter_push batch by batch.  On error no output batch is left allocated. */
int sqlrs_filter_push_many(sqlrs_filter_t *f, int n, const sqlrs_batch_t *const *in, int out_mem, sqlrs_batch_t **out);
/* The reference's own calling shape — ONE batch per poll (filter.rs:15-24; 1024 rows, storage/csv.rs:105) — without a stream
 * synchronisation per call: push_async queues the batch and returns a TICKET for the HOST batch sqlrs_filter_push(f, in,
 * SQLRS_MEM_HOST, ..) would have returned; sqlrs_batch_wait(ticket, &out) blocks until that batch exists, hands it out and
 * consumes the ticket (also after an error).  `in` is read completely before push_async returns.  Tickets of one ctx
 * complete in issue order; a caller that waits a few batches behind (the stream adaptor of the Rust / C++ mirrors keeps
 * `depth` tickets in flight) never blocks on the device.  HOST batches of <= 4096 rows and <= 12 int32 / int64 / float64 /
 * utf8 columns under ANY predicate over their int32 / int64 / float64 columns (comparisons, + - * / with the evaluator's wrapping
 * and its "Divide by zero error" — reported by sqlrs_batch_wait of that ticket —, casts between the three types, NULL
 * constants, AND / OR; <= 24 expression nodes) share ONE launch with up to three neighbours and take no copy call (pinned,
 * device-mapped ring); every other batch (a predicate that reads a utf8 / boolean column, DEVICE input, more rows or columns)
 * runs the synchronous operator inside push_async: same batches, no speed-up.  Tickets must be waited for before their ctx is
 * destroyed. */
typedef struct sqlrs_ticket sqlrs_ticket_t;
int sqlrs_filter_push_async(sqlrs_filter_t *f, const sqlrs_batch_t *in, sqlrs_ticket_t **ticket);
int sqlrs_batch_wait(sqlrs_ticket_t *ticket, sqlrs_batch_t **out);
void sqlrs_filter_destroy(sqlrs_filter_t *f);

/* Evaluates one expression on a batch -> one-column batch.
 * [ref: src/executor/evaluator.rs:13-28  BoundExpr::eval_column] */
int sqlrs_eval_expr(sqlrs_ctx_t *ctx, const sqlrs_expr_t *expr, const sqlrs_batch_t *in,
                    int out_mem, sqlrs_batch_t **out);

/* --------------------------------------------------------------- HashJoin -- */
/* [ref: src/executor/join/hash_join.rs:16-23  HashJoinExecutor{left_child, right_child,
 *  join_type, join_condition, join_output_schema}; execute :146-323]
 * left = build side, right = probe side (hash_join.rs:162,208). */
typedef struct sqlrs_hash_join sqlrs_hash_join_t;
int sqlrs_hash_join_create(sqlrs_ctx_t *ctx, int join_type, int num_keys,
                           const sqlrs_expr_t *left_keys,  /* JoinCondition::On.on[i].0 */
                           const sqlrs_expr_t *right_keys, /* JoinCondition::On.on[i].1 */
                           const sqlrs_expr_t *filter,     /* JoinCondition::On.filter or NULL */
                           int num_right_columns,          /* right part of join_output_schema */
                           const int32_t *right_dtypes,    /* (types of the all-NULL tail columns) */
                           sqlrs_hash_join_t **out);
/* build phase, one call per left batch  [ref: hash_join.rs:161-181] */
int sqlrs_hash_join_build_push(sqlrs_hash_join_t *j, const sqlrs_batch_t *left);
/* end of left stream: concat + finish the table  [ref: hash_join.rs:183-187] */
int sqlrs_hash_join_build_finish(sqlrs_hash_join_t *j);
/* probe phase, one call per right batch -> one joined batch (possibly 0 rows).
 * If the build side received no batch, *out = NULL ("emit nothing", hash_join.rs:183-185).
 * [ref: hash_join.rs:207-292] */
int sqlrs_hash_join_probe_push(sqlrs_hash_join_t *j, const sqlrs_batch_t *right, int out_mem,
                               sqlrs_batch_t **out);
/* n probe batches in ONE call: out[i] (n entries) is what sqlrs_hash_join_probe_push(right[i]) returns — one joined batch
 * per probe batch, in order [ref: hash_join.rs:207-292] — for the reference's batch shape, 1024-row HOST batches
 * [ref: src/storage/csv.rs:105]: Inner / Left joins without a join filter over small HOST batches of fixed-width columns
 * are probed as one batch (pairs are probe-row major, hash_join.rs:225-234: every input batch's joined rows are one
 * contiguous range) and cut back into n HOST batches; anything else runs batch by batch.  On error no output batch is
 * left allocated. */
int sqlrs_hash_join_probe_push_many(sqlrs_hash_join_t *j, int n, const sqlrs_batch_t *const *right, int out_mem,
                                    sqlrs_batch_t **out);
/* sqlrs_hash_join_probe_push without the wait (see sqlrs_filter_push_async; hash_join.rs:207-292 polled one batch at a
 * time): the fast path takes Inner joins without a join filter over ONE exactly compared INPUT_REF key (int32 / int64 /
 * float64, no NULL probe keys), unique build keys and fixed-width columns on both sides; *out of the wait is NULL for an
 * empty build side, as for probe_push. */
int sqlrs_hash_join_probe_push_async(sqlrs_hash_join_t *j, const sqlrs_batch_t *right, sqlrs_ticket_t **ticket);
/* The index-pair form of one probe batch, before any gather: 2 columns
 * (UINT64 left index, nullable; UINT32 right index), in the reference's order
 * (probe-row major, build insertion order minor), join filter NOT applied.
 * [ref: hash_join.rs:218-253  left_indices / right_indices] */
int sqlrs_hash_join_probe_indices(sqlrs_hash_join_t *j, const sqlrs_batch_t *right, int out_mem,
                                  sqlrs_batch_t **out);
/* end of right stream: the unvisited-left tail batch for Left/Full (always a
 * batch, possibly 0 rows), *out = NULL for Inner/Right or an empty build side.
 * [ref: hash_join.rs:296-322] */
int sqlrs_hash_join_finish(sqlrs_hash_join_t *j, int out_mem, sqlrs_batch_t **out);
void sqlrs_hash_join_destroy(sqlrs_hash_join_t *j);

/* ---------------------------------------------------------------- HashAgg -- */
/* [ref: src/executor/aggregate/hash_agg.rs:15-19  HashAggExecutor{agg_funcs, group_by, child};
 *  execute :32-150; accumulators aggregate/{count,sum,min_max}.rs]
 * Output = one batch [group keys..., aggs...], groups in first-seen order.
 * COUNT accumulates across batches (the reference assigns, count.rs:22: see DESIGN.md). */
typedef struct sqlrs_hash_agg sqlrs_hash_agg_t;
int sqlrs_hash_agg_create(sqlrs_ctx_t *ctx, int num_group_by, const sqlrs_expr_t *group_by,
                          int num_aggs, const sqlrs_agg_func_t *aggs, sqlrs_hash_agg_t **out);
/* [ref: hash_agg.rs:44-122]
 * BLOCKING operators (this one, sqlrs_join_agg_probe_push, sqlrs_order_push, sqlrs_hash_join_build_push) decide when
 * to work: small HOST batches of fixed-width columns — the reference's 1024-row CSV batches — are appended to a host
 * staging area (one memcpy) and uploaded together once per 2^22 rows or at *_finish; device batches are staged by
 * reference.  Consequence for error reporting: an error that only the evaluation of a batch reveals (an expression
 * over a column of the wrong type, "Divide by zero error", a schema that differs from the first batch's) is returned
 * by the call that works on the staged rows — a LATER push or *_finish — not necessarily by the push that delivered
 * the batch; the operator is unusable after an error either way (destroy it), exactly like the reference's stream,
 * which ends at its first Err.  A wide aggregate list (more than two argument columns) runs as several internal
 * operators that each see every batch. */
int sqlrs_hash_agg_push(sqlrs_hash_agg_t *a, const sqlrs_batch_t *in);
/* [ref: hash_agg.rs:124-149]; with no pushed batch the reference panics (:125):
 * here that is SQLRS_ERR_INTERNAL. */
int sqlrs_hash_agg_finish(sqlrs_hash_agg_t *a, int out_mem, sqlrs_batch_t **out);
/* Group order of the output batch.  SQLRS_GROUP_ORDER_FIRST_SEEN (default) is the reference's
 * (hash_agg.rs:98,132).  SQLRS_GROUP_ORDER_ANY is for PARTIAL aggregates that are exchanged and
 * merged again (multi-GPU): the sort of the groups by first row and the gathers
 * that follow it are skipped; the set of groups and every value are unchanged. */
#define SQLRS_GROUP_ORDER_FIRST_SEEN 0
#define SQLRS_GROUP_ORDER_ANY 1
int sqlrs_hash_agg_set_group_order(sqlrs_hash_agg_t *a, int group_order);
/* A FilterExecutor sitting directly below the operator — PhysicalHashAgg(PhysicalFilter(child)), the shape of
 * `SELECT k, sum(v) FROM t WHERE t.col > c GROUP BY k` [ref: filter.rs:13-25 feeding hash_agg.rs:44].  `filter`
 * indexes the columns of the pushed batches.  Results are identical to sqlrs_filter_push on every batch followed
 * by sqlrs_hash_agg_push of its output (groups in first-seen order of the FILTERED rows).  When the predicate is
 * `column OP constant` over an int64 / float64 column without NULLs, the batch is one that is aggregated in
 * place (>= 2^26 rows, first batch) and neither the keys nor the aggregate arguments contain a division, the
 * first partition pass evaluates the predicate itself (no filtered copy of any column is written); otherwise
 * the library runs the Filter operator first.  Must be called before the first batch; NULL / empty = none. */
int sqlrs_hash_agg_set_filter(sqlrs_hash_agg_t *a, const sqlrs_expr_t *filter);
/* batches whose filter was evaluated inside the first partition pass (diagnostics / tests) */
int64_t sqlrs_hash_agg_filter_fused_batches(const sqlrs_hash_agg_t *a);
void sqlrs_hash_agg_destroy(sqlrs_hash_agg_t *a);

/* ------------------------------------------------------------------ Order -- */
/* [ref: src/executor/order.rs:8-67  OrderExecutor{order_by, child}::execute]
 * NULLs first (SortOptions::default().nulls_first, order.rs:37-40), stable. */
typedef struct sqlrs_order sqlrs_order_t;
int sqlrs_order_create(sqlrs_ctx_t *ctx, int num_keys, const sqlrs_order_by_t *order_by,
                       sqlrs_order_t **out);
int sqlrs_order_push(sqlrs_order_t *o, const sqlrs_batch_t *in);
/* The reference's Order keeps the child's batches themselves (order.rs:19-26: Arc'd arrays in a Vec, no copy).
 * sqlrs_order_push must copy device columns it does not own, because a batch is only borrowed for the call
 * (1.6 GB, 0.64 ms for 1e8 rows x 2 columns); with this entry point the CALLER keeps every buffer of `in` alive
 * and unchanged until sqlrs_order_finish has returned (or the operator is destroyed), the stream rules of
 * sqlrs_batch_t apply until then, and nothing is copied.  Host columns are uploaded as usual. */
int sqlrs_order_push_retained(sqlrs_order_t *o, const sqlrs_batch_t *in);
int sqlrs_order_finish(sqlrs_order_t *o, int out_mem, sqlrs_batch_t **out);
/* ORDER BY ... LIMIT k — PhysicalLimit(PhysicalOrder(child)) [ref: limit.rs:12-80 slicing what order.rs:27-66 sorted]:
 * the caller promises to read only the first `rows` (= offset + limit) rows of the result.  sqlrs_order_finish may then
 * return a PREFIX of the sorted result with at least `rows` rows (or everything): it keeps the rows whose key cannot be
 * beyond position `rows` — a threshold from a sample of the keys, ties at the threshold included — and sorts only those,
 * so the prefix is exactly what the full sort would put first, stable ties and all.  Applies when the FIRST key is a plain
 * fixed-width column without NULLs (the threshold is taken on it; further keys only order the candidates) and `rows` is a
 * small part of the input; ignored otherwise (0 = no hint).  The LimitExecutor above
 * slices either result the same way. */
int sqlrs_order_set_limit(sqlrs_order_t *o, int64_t rows);
/* rows the last finish sorted because of the hint; 0 = the hint did not apply (diagnostics / tests) */
int64_t sqlrs_order_topk_candidates(const sqlrs_order_t *o);
void sqlrs_order_destroy(sqlrs_order_t *o);

/* ---------------------------------------------------- HashJoin + HashAgg fused -- */
/* A HashAggExecutor sitting directly on an Inner HashJoinExecutor without join filter
 * [ref: hash_agg.rs:32-150 consuming hash_join.rs:146-323] — what a physical rewrite of
 * PhysicalHashAgg(PhysicalHashJoin(left, right)) instantiates.  Results are identical to running
 * the two operators back to back: group_by / aggregate argument expressions index the JOIN
 * OUTPUT schema (left columns, then right columns), groups come out in first-seen order of the
 * join output.  When the join has one exactly-compared key with unique build keys, the group key
 * is that key and the aggregate arguments only read probe-side columns, the joined batch is never
 * materialised (probe rows are partitioned once, each LDS bucket table is pre-filled with the
 * build keys); otherwise the library composes the two operators on the device itself. */
typedef struct sqlrs_join_agg sqlrs_join_agg_t;
int sqlrs_join_agg_create(sqlrs_ctx_t *ctx, int num_keys, const sqlrs_expr_t *left_keys,
                          const sqlrs_expr_t *right_keys, int num_left_columns, int num_right_columns,
                          const int32_t *right_dtypes, int num_group_by, const sqlrs_expr_t *group_by,
                          int num_aggs, const sqlrs_agg_func_t *aggs, sqlrs_join_agg_t **out);
int sqlrs_join_agg_build_push(sqlrs_join_agg_t *ja, const sqlrs_batch_t *left);
int sqlrs_join_agg_build_finish(sqlrs_join_agg_t *ja);
int sqlrs_join_agg_probe_push(sqlrs_join_agg_t *ja, const sqlrs_batch_t *right);
int sqlrs_join_agg_finish(sqlrs_join_agg_t *ja, int out_mem, sqlrs_batch_t **out);
int sqlrs_join_agg_set_group_order(sqlrs_join_agg_t *ja, int group_order); /* see sqlrs_hash_agg_set_group_order */
/* probe inputs that took the non-materialising route so far (small probe batches are staged and
 * processed together: they count once) */
int64_t sqlrs_join_agg_fused_batches(const sqlrs_join_agg_t *ja);
/* A FilterExecutor sitting directly below the probe side — PhysicalHashAgg(PhysicalHashJoin(left,
 * PhysicalFilter(right))), the shape of `... FROM fact JOIN dim ... WHERE fact.col > k GROUP BY ...`
 * [ref: filter.rs:13-25 feeding hash_join.rs:207].  `filter` indexes the PROBE batch's columns.  Results
 * are identical to sqlrs_filter_push on every probe batch followed by sqlrs_join_agg_probe_push of its
 * output; when the predicate is `column OP constant` over an int64 / float64 column without NULLs and the
 * non-materialising route applies, the first partition pass evaluates it itself (the filtered copies of
 * the probe columns are never written), otherwise the library runs the Filter operator first.  Must be
 * called before the first probe batch; NULL / empty removes the filter. */
int sqlrs_join_agg_set_probe_filter(sqlrs_join_agg_t *ja, const sqlrs_expr_t *filter);
/* probe batches whose filter was evaluated inside the first partition pass (diagnostics / tests) */
int64_t sqlrs_join_agg_filter_fused_batches(const sqlrs_join_agg_t *ja);
/* GROUP BY over columns of the BUILD side (`... FROM fact JOIN dim ON fact.k = dim.k GROUP BY dim.region`,
 * PhysicalHashAgg over PhysicalHashJoin with group_by = InputRefs below num_left_columns) with unique build keys:
 * every build column is a function of the join key, so the operator groups the probe rows by JOIN KEY first (its
 * non-materialising route), finds the build row of every distinct key once, and re-aggregates those partial rows by
 * the requested columns (COUNT -> sum of counts; SUM / MIN / MAX of partials).  Same groups, same first-seen group
 * order [ref: hash_agg.rs:98] and the same aggregates as HashAgg over the materialised join [ref: hash_join.rs:284-291
 * feeding hash_agg.rs:44-122]; SUM(double) within the summation-order tolerance.  Returns the number of partial
 * groups (distinct join keys) the last finish re-aggregated; 0 = the route did not run (diagnostics / tests). */
int64_t sqlrs_join_agg_eager_groups(const sqlrs_join_agg_t *ja);
void sqlrs_join_agg_destroy(sqlrs_join_agg_t *ja);

/* ---------------------------------------- operators either side of the hot path -- */
/* [ref: src/executor/project.rs:6-29  ProjectExecutor{exprs, child}]  One output batch per input batch,
 * column i = exprs[i].eval_column(batch). */
typedef struct sqlrs_project sqlrs_project_t;
int sqlrs_project_create(sqlrs_ctx_t *ctx, int num_exprs, const sqlrs_expr_t *exprs, sqlrs_project_t **out);
int sqlrs_project_push(sqlrs_project_t *p, const sqlrs_batch_t *in, int out_mem, sqlrs_batch_t **out);
/* n input batches in ONE call: out[i] (n entries) is what sqlrs_project_push(in[i]) returns — one output batch per input
 * batch, in order [ref: project.rs:15-27] — for the reference's batch shape, 1024-row HOST batches
 * [ref: src/storage/csv.rs:105]: small HOST batches of fixed-width columns are uploaded together and projected by one
 * launch sequence (an expression's row i depends on row i alone); anything else runs batch by batch.  On error no output
 * batch is left allocated. */
int sqlrs_project_push_many(sqlrs_project_t *p, int n, const sqlrs_batch_t *const *in, int out_mem, sqlrs_batch_t **out);
/* sqlrs_project_push without the wait (see sqlrs_filter_push_async; project.rs:15-27 polled one batch at a time): the fast
 * path takes HOST batches of <= 4096 rows whose output columns are bare column references (int32 / int64 / float64 /
 * boolean / utf8) or expressions over int32 / int64 / float64 columns with an int32 / int64 / float64 / boolean result
 * (<= 6 computed columns, <= 12 columns in and out); sqlrs_batch_wait reports a division by zero of that batch. */
int sqlrs_project_push_async(sqlrs_project_t *p, const sqlrs_batch_t *in, sqlrs_ticket_t **ticket);
void sqlrs_project_destroy(sqlrs_project_t *p);

/* [ref: src/executor/limit.rs:4-81  LimitExecutor{limit, offset, child}]  limit / offset are the constants
 * of the two BoundExpr::Constant (has_* = 0: None).  One call per child batch, in stream order, with the
 * reference's arithmetic across batches: *out = NULL when the batch contributes nothing, otherwise the
 * batch itself or its slice; *done = 1 once no later batch can contribute (the reference `break`s, or
 * returned before polling the child when limit = 0) — the caller stops pulling the child then. */
typedef struct sqlrs_limit sqlrs_limit_t;
int sqlrs_limit_create(sqlrs_ctx_t *ctx, int has_limit, int64_t limit, int has_offset, int64_t offset,
                       sqlrs_limit_t **out);
int sqlrs_limit_push(sqlrs_limit_t *l, const sqlrs_batch_t *in, int out_mem, sqlrs_batch_t **out, int *done);
void sqlrs_limit_destroy(sqlrs_limit_t *l);

/* [ref: src/executor/join/cross_join.rs:8-58  CrossJoinExecutor{left_child, right_child, join_output_schema};
 *  instantiated at src/executor/mod.rs:116-125]  What the binder rewrites an uncorrelated scalar subquery to
 * (src/binder/table/subquery.rs:120-167: base table CROSS JOIN the one-row subquery).  The left child is collected and
 * concatenated (cross_join.rs:30-36); for every right batch and every LEFT ROW the reference yields one batch = that
 * row's values repeated right.num_rows times | the right columns (cross_join.rs:39-55).  probe_push returns those
 * batches of ONE right batch as a single batch in the same row order (left row 0 x right rows, left row 1 x right
 * rows, ...); the host mirrors slice it back into left_rows batches of right.num_rows rows.  *out = NULL when the left
 * child yielded no batch (cross_join.rs:32-34) or no row. */
typedef struct sqlrs_cross_join sqlrs_cross_join_t;
int sqlrs_cross_join_create(sqlrs_ctx_t *ctx, sqlrs_cross_join_t **out);
int sqlrs_cross_join_build_push(sqlrs_cross_join_t *j, const sqlrs_batch_t *left);
int sqlrs_cross_join_probe_push(sqlrs_cross_join_t *j, const sqlrs_batch_t *right, int out_mem, sqlrs_batch_t **out);
/* The same for the left rows [left_begin, left_begin + left_rows) only: a library batch holds < 2^31 rows, the reference's
 * one-batch-per-left-row stream has no such limit, so a caller whose left rows x right rows exceed it takes the left rows in
 * ranges (sqlrs_cross_join_left_rows = rows pushed so far) and gets the same stream, range after range. */
int sqlrs_cross_join_probe_push_range(sqlrs_cross_join_t *j, const sqlrs_batch_t *right, int64_t left_begin, int64_t left_rows,
                                      int out_mem, sqlrs_batch_t **out);
int64_t sqlrs_cross_join_left_rows(const sqlrs_cross_join_t *j);
void sqlrs_cross_join_destroy(sqlrs_cross_join_t *j);

/* [ref: src/executor/aggregate/simple_agg.rs:10-66  SimpleAggExecutor{agg_funcs, child}]  Aggregates
 * without GROUP BY: one accumulator per aggregate over all rows, exactly one output row (COUNT = 0 and
 * NULL for the others when no row arrived); finish without any pushed batch is SQLRS_ERR_INTERNAL (the
 * reference unwraps a None, :63). */
typedef struct sqlrs_simple_agg sqlrs_simple_agg_t;
int sqlrs_simple_agg_create(sqlrs_ctx_t *ctx, int num_aggs, const sqlrs_agg_func_t *aggs, sqlrs_simple_agg_t **out);
int sqlrs_simple_agg_push(sqlrs_simple_agg_t *a, const sqlrs_batch_t *in);
int sqlrs_simple_agg_finish(sqlrs_simple_agg_t *a, int out_mem, sqlrs_batch_t **out);
void sqlrs_simple_agg_destroy(sqlrs_simple_agg_t *a);

/* [ref: src/util/mod.rs:53-80  record_batch_to_string]  The sqllogictest text form of a batch (host or
 * device resident): one line per row, one blank between columns, NULL -> "NULL", empty string ->
 * "(empty)", other values as arrow's array_value_to_string prints them (floats: Rust's Display).
 * *out is a NUL-terminated string owned by the library until sqlrs_string_free. */
int sqlrs_batch_to_string(sqlrs_ctx_t *ctx, const sqlrs_batch_t *in, char **out);
void sqlrs_string_free(char *s);

/* CSV ingest [ref: src/storage/csv.rs:92-106 CsvConfig (header, ',', infer over 10 records, batches of
 * 1024 rows), :124-133 schema inference, :190-241 CsvTransaction::next_batch].  Parsed on the host; with
 * out_mem = SQLRS_MEM_DEVICE every batch is uploaded once and the scan's output is HBM resident.
 * Inferred types: INT64, FLOAT64, BOOLEAN, else UTF8 (date-like columns stay UTF8: the path has no date
 * type); a missing value is NULL in a typed column and the empty string in a UTF8 column. */
typedef struct sqlrs_csv sqlrs_csv_t;
int sqlrs_csv_open(sqlrs_ctx_t *ctx, const char *path, int has_header, char delimiter, int64_t batch_size,
                   int64_t infer_max_records, sqlrs_csv_t **out);
int sqlrs_csv_num_columns(const sqlrs_csv_t *r);
const char *sqlrs_csv_column_name(const sqlrs_csv_t *r, int i);
int sqlrs_csv_column_dtype(const sqlrs_csv_t *r, int i);
/* Bounds (offset, limit) of the scan over the data records, limit < 0 = none [ref: csv.rs:207-223]; a file
 * WITHOUT a header yields limit + 1 records, as the reference does (its (offset, offset + limit + 1) line bounds meet
 * arrow-csv's line counter, which starts one later only when there is a header); call before the first next_batch */
int sqlrs_csv_set_bounds(sqlrs_csv_t *r, int64_t offset, int64_t limit);
/* Projections: indices into the file's columns [ref: csv.rs:224] */
int sqlrs_csv_set_projection(sqlrs_csv_t *r, int num_columns, const int32_t *columns);
/* *out = NULL at the end of the scan */
int sqlrs_csv_next_batch(sqlrs_csv_t *r, int out_mem, sqlrs_batch_t **out);
void sqlrs_csv_close(sqlrs_csv_t *r);

/* --------------------------------------------------------------- exchange -- */
/* Hash-partitions a batch on one key expression for the multi-GPU partitioned join /
 * group-by (no reference analogue: sqlrs is single process).  The output batch
 * holds the input rows permuted so that partition p = rows [offsets[p], offsets[p+1]), input
 * order preserved inside a partition; `offsets` (host, num_parts + 1 entries) is filled by the
 * call.  Rows with equal keys always land in the same partition; NULL keys go to partition 0.
 * The caller exchanges the slices (RCCL all-to-all over xGMI) and feeds what it receives to
 * the ordinary operators above. */
int sqlrs_hash_partition(sqlrs_ctx_t *ctx, const sqlrs_batch_t *in, const sqlrs_expr_t *key,
                         int num_parts, int out_mem, sqlrs_batch_t **out, int64_t *offsets);
/* Filter + hash partition in one pass: the rows of `in` that pass `predicate` (FilterExecutor semantics,
 * [ref: src/executor/filter.rs:13-25]: rows whose mask is false or NULL are dropped; NULL / zero nodes =
 * no predicate) distributed like sqlrs_hash_partition.  Partition p = rows
 * [part_start[p], part_start[p] + part_rows[p]) of the output batch (both arrays: host, num_parts
 * entries, filled by the call); rows outside those ranges are unspecified padding and the row order
 * inside a partition is unspecified — what the equi-join / group-by exchange needs, nothing more.
 * `column OP constant` predicates over an int64 / float64 column without NULLs on batches of <= 3
 * fixed-width 8-byte columns (the filtered fact rows of the partitioned hash join) run as ONE kernel:
 * every row is read once, every kept row written once, into per-partition regions of num_rows rows each
 * (num_parts x the input's size of device memory).  Any other shape = sqlrs_filter_* followed by
 * sqlrs_hash_partition inside the library (same rows per partition). */
int sqlrs_hash_partition_filter(sqlrs_ctx_t *ctx, const sqlrs_batch_t *in, const sqlrs_expr_t *key,
                                const sqlrs_expr_t *predicate, int num_parts, int out_mem,
                                sqlrs_batch_t **out, int64_t *part_start, int64_t *part_rows);

/* The exchange itself: an all-to-all of hash partitions over RCCL (xGMI) on the ctx stream, one process per GPU
 * (no reference analogue: sqlrs is a single process; it slots under the executors the builder instantiates
 * for a partitioned join / aggregate [ref: src/executor/mod.rs:103-114,163-174]: their children are wrapped in an
 * exchange of the children's hash partitions).  RCCL is dlopen'ed at the first call; none of this needs torch.
 *   sqlrs_exchange_unique_id   one rank makes the 128-byte id (ncclGetUniqueId) and hands it to the others out of band;
 *   sqlrs_exchange_create      collective: every rank calls it with the same id, its rank and the world size;
 *   sqlrs_exchange_all_to_all  collective: `in` is a DEVICE batch of fixed-width columns (NULLs allowed: validity travels
 *       as a byte per row beside the values) whose rows [part_start[p], part_start[p] + part_rows[p]) go to rank p (what
 *       sqlrs_hash_partition / _filter return; host arrays of `world` entries); *out = the rows received from rank 0, 1,
 *       ... in this order (DEVICE); recv_rows (optional, host, `world` entries) = rows received per rank.  All columns of
 *       the call travel in ONE ncclGroupStart/End.  A rank whose arguments are unusable (Utf8 / Boolean column, partition
 *       outside the batch) says so in the count all-gather: EVERY rank then returns SQLRS_ERR_INTERNAL from this call and
 *       nothing is sent (*out = NULL).  `in` is read stream-ordered: keep it alive until the ctx stream has passed
 *       (sqlrs_ctx_synchronize / release_to_stream), like every borrowed device batch;
 *   sqlrs_exchange_begin / _send_chunk / _finish  the same exchange for a SEQUENCE of chunks into ONE received batch (the
 *       fact rows of the partitioned join, filtered and partitioned chunk by chunk): begin names the column types (and a
 *       capacity hint in rows; the receive batch grows as needed), every send_chunk is a collective like all_to_all, finish
 *       returns the rows of all chunks (per chunk: from rank 0, 1, ...).  send_chunk(k) puts the count words of chunk k on the
 *       wire and sends the payload of chunk k - 1, whose words reached the host while chunk k was being produced: the
 *       host does not wait for the device inside the loop, and the transfer of chunk k - 1 overlaps the partitioning of
 *       chunk k + 1 on the stream.  A batch this library produced is retained by reference until its payload is queued;
 *       any other DEVICE batch is copied once;
 *   sqlrs_exchange_plan        the receive side's bookkeeping as plain host arithmetic (no device): send_rows_all[q * world
 *       + p] = rows rank q sends to rank p -> rows / start offsets this rank receives per source rank. */
#define SQLRS_EXCHANGE_ID_BYTES 128
typedef struct sqlrs_exchange sqlrs_exchange_t;
int sqlrs_exchange_unique_id(sqlrs_ctx_t *ctx, void *id_out);
int sqlrs_exchange_create(sqlrs_ctx_t *ctx, const void *unique_id, int rank, int world, sqlrs_exchange_t **out);
int sqlrs_exchange_all_to_all(sqlrs_exchange_t *x, const sqlrs_batch_t *in, const int64_t *part_start, const int64_t *part_rows,
                              sqlrs_batch_t **out, int64_t *recv_rows);
int sqlrs_exchange_begin(sqlrs_exchange_t *x, int num_columns, const int32_t *dtypes, int64_t capacity_rows);
int sqlrs_exchange_send_chunk(sqlrs_exchange_t *x, const sqlrs_batch_t *in, const int64_t *part_start, const int64_t *part_rows);
int sqlrs_exchange_finish(sqlrs_exchange_t *x, sqlrs_batch_t **out);
int sqlrs_exchange_plan(int world, int rank, const int64_t *send_rows_all, int64_t *recv_rows, int64_t *recv_start, int64_t *total);
int64_t sqlrs_exchange_bytes_off_rank(const sqlrs_exchange_t *x);
void sqlrs_exchange_destroy(sqlrs_exchange_t *x);

/* ------------------------------------------------- timing of device work -- */
/* HIP-event timing on the ctx stream (bench.py measures the dominant kernel
 * with these: torch.cuda.Event only sees torch's own stream). */
typedef struct sqlrs_timer sqlrs_timer_t;
int sqlrs_timer_create(sqlrs_ctx_t *ctx, sqlrs_timer_t **out);
int sqlrs_timer_start(sqlrs_timer_t *t);
int sqlrs_timer_stop(sqlrs_timer_t *t);
/* synchronises on the stop event */
int sqlrs_timer_elapsed_ms(sqlrs_timer_t *t, double *ms);
void sqlrs_timer_destroy(sqlrs_timer_t *t);

/* Per-kernel-class accumulated device time since the last reset, measured with
 * HIP events around each launch group when profiling is enabled (off by
 * default; enabling it serialises nothing but adds two event records per group). */
int sqlrs_ctx_profile_enable(sqlrs_ctx_t *ctx, int on);
int sqlrs_ctx_profile_reset(sqlrs_ctx_t *ctx);
/* Writes up to `cap` entries; returns the number of classes. Names are static strings. */
int sqlrs_ctx_profile_read(sqlrs_ctx_t *ctx, int cap, const char **names, double *total_ms,
                           int64_t *launches);

/* Library version string ("sqlrs-hip <semver> gfx950"). */
const char *sqlrs_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SQLRS_HIP_H */
