// hashagg_op.hip — HashAggExecutor entry points (src/executor/aggregate/hash_agg.rs:32-150).
//
// push() stages the batch (HashAgg is blocking; a first batch of >= 2^26 rows is taken in place);
// staged rows are aggregated together at finish() / every 2^28 rows by one of two routes:
//   * row route      (small inputs): every row resolves to its group through the global table
//                    and updates the dense accumulators with global atomics (agg.hip);
//   * partition route (>= 2^21 rows, COUNT/SUM/MIN/MAX over <= 2 argument columns): rows are
//                    pre-aggregated per LDS bucket (agg_partition.hip); the resulting groups are
//                    the operator state as they are, or go through the row route carrying
//                    pre-aggregated cells when the table already holds groups.
// Both leave the same state; finish() emits groups in first-seen order (hash_agg.rs:98,132).
#include <cstdlib>

#include "agg_partition.hpp"
#include "agg_state.hpp"
#include "common.hpp"
#include "device_utils.hpp"
#include "host_stage.hpp"
#include "prims.hpp"

using namespace sq;

namespace {

struct AggSpec {
  int func = 0, distinct = 0;
  int32_t return_dtype = 0;
  Expr arg;
  GrowBuf acc, nn;
  bool track_nn = false; // nn maintained (COUNT always; others once a NULL input was seen)
  int32_t acc_dtype = 0; // dtype the accumulator works in
  int argcol = -1;       // index into the per-batch evaluated argument columns
  // MIN/MAX over Utf8 (min_max.rs:21-29 min_string/max_string): the strings of every pushed batch
  // and their rows' group ids are kept; finish ranks all strings once and reduces ranks per group
  struct StrBatch {
    DCol col;
    BufP gid;
    int64_t n;
  };
  std::vector<StrBatch> strs;
};

uint64_t acc_identity(const AggSpec &a) { return a.func == SQLRS_AGG_MIN ? ~0ull : 0ull; }

bool same_expr(const Expr &x, const Expr &y) {
  if (x.nodes.size() != y.nodes.size()) return false;
  for (size_t i = 0; i < x.nodes.size(); i++) {
    const sqlrs_expr_node_t &p = x.nodes[i], &q = y.nodes[i];
    if (p.op != q.op || p.dtype != q.dtype || p.index != q.index || p.is_null != q.is_null || p.i != q.i ||
        std::memcmp(&p.f, &q.f, 8) != 0 || x.strings[i] != y.strings[i])
      return false;
  }
  return true;
}

} // namespace

// Groups of one pre-aggregated batch that have not been merged into the table yet: while the
// operator has seen nothing else they ARE its state, and finish() can emit them directly.
struct PendingGroups {
  bool active = false;
  PartAggOutput po;
  std::vector<DCol> keyvals;          // key column values per group (list order)
  std::vector<int> cnt_of_col, acc_of_agg, col_of_agg;
  std::vector<uint8_t> col_nullable;  // value column carried NULLs in that batch
  std::vector<int32_t> col_dtype;
  bool exact = true;
  int32_t key_dtype = SQLRS_INT64;
};

struct sqlrs_hash_agg {
  Ctx *ctx = nullptr;
  HostStage hstage; // small HOST batches wait here and are uploaded together (host_stage.hpp)
  PendingGroups pending;
  std::vector<Expr> group_by;
  std::vector<AggSpec> aggs;
  AggState st;
  bool saw_batch = false, in_order = true;
  // Batches wait here (evaluated key + argument columns, private copies) and are aggregated as ONE
  // batch: HashAgg is a blocking operator, and pre-aggregating every pushed batch on its own means
  // merging its groups into the global table with ~24 G atomics/s (4 batches of 5e7 rows with 1e7
  // groups: 35 ms, against 7.7 ms for the same rows as one batch).
  struct Staged {
    std::vector<DCol> kcols, acols;
    int64_t n = 0;
  };
  std::vector<Staged> staged;
  int64_t staged_rows = 0;
  bool any_order = false; // SQLRS_GROUP_ORDER_ANY: partial aggregates, no first-seen ordering at finish
  bool strong_keys = false; // internal de-dup stage of a DISTINCT aggregate
  int64_t rows_seen = 0;
  std::vector<int32_t> key_dtypes;
  std::vector<std::vector<DCol>> key_parts; // per key column: values of new groups, per batch
  // distinct argument columns: (expression, cast target) pairs shared by the aggregates
  std::vector<Expr> arg_exprs;
  std::vector<int32_t> arg_cast; // 0 = none
  // DISTINCT aggregates (count.rs:31-58, sum.rs:99-132): one de-duplicating sub-aggregation per
  // aggregate, grouped by (group keys..., argument); re-aggregated per group at finish()
  struct DistinctAgg {
    int func = 0;
    int32_t return_dtype = 0;
    sqlrs_hash_agg *dedup = nullptr;
  };
  std::vector<DistinctAgg> distinct_aggs;
  std::vector<std::pair<int, int>> out_order; // (0 = plain | 1 = distinct, index) per output aggregate
  // FilterExecutor directly below the operator (sqlrs_hash_agg_set_filter)
  bool has_filter = false;
  Expr filter;
  int64_t filter_fused_batches = 0;
  // Wide aggregate lists: the partition route carries at most two 8-byte argument columns and PART_MAX_ACC
  // accumulator cells per bucket-table slot; an operator asking for more (SUM(a), SUM(b), SUM(c), COUNT(*) ...) used to
  // fall to the row route — one global atomic per row and accumulator, 24 G/s.  Such an operator is instead run as
  // several PARTS with the same GROUP BY and disjoint subsets of the aggregates (each within those limits): every
  // part sees every batch, all of them find the same groups in the same first-seen order (hash_agg.rs:98: the order
  // depends on the key columns alone), and finish() puts their aggregate columns side by side.
  std::vector<sqlrs_hash_agg *> parts;
  std::vector<std::pair<int, int>> part_col; // per aggregate, output order: (part, aggregate index inside the part)
  ~sqlrs_hash_agg() {
    for (auto &d : distinct_aggs) delete d.dedup;
    for (auto *p : parts) delete p;
  }
};

extern "C" int sqlrs_hash_agg_push(sqlrs_hash_agg_t *a, const sqlrs_batch_t *in);
extern "C" int sqlrs_hash_agg_finish(sqlrs_hash_agg_t *a, int out_mem, sqlrs_batch_t **out);
extern "C" int sqlrs_hash_agg_create(sqlrs_ctx_t *ctx, int num_group_by, const sqlrs_expr_t *group_by, int num_aggs,
                                     const sqlrs_agg_func_t *aggs, sqlrs_hash_agg_t **out);
extern "C" void sqlrs_hash_agg_destroy(sqlrs_hash_agg_t *a);
extern "C" void sqlrs_batch_release(sqlrs_batch_t *batch);
extern "C" int sqlrs_filter_create(sqlrs_ctx_t *, const sqlrs_expr_t *, sqlrs_filter_t **);
extern "C" int sqlrs_filter_push(sqlrs_filter_t *, const sqlrs_batch_t *, int, sqlrs_batch_t **);
extern "C" void sqlrs_filter_destroy(sqlrs_filter_t *);

namespace sq {
bool fusable_row_filter(const Expr &e, InBatch &ib, RowFilter *rf);

__global__ void gather_u32_kernel(const uint32_t *__restrict__ src, const uint32_t *__restrict__ idx,
                                  int64_t n, uint32_t *__restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = src[idx[i]];
}
__global__ void rowid_from_u32_kernel(const uint32_t *__restrict__ rows, int64_t n, uint64_t offset,
                                      uint64_t *__restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = offset + rows[i];
}

// One evaluated argument column of the current (sub-)batch
struct ArgView {
  const void *values = nullptr;
  const uint64_t *validity = nullptr;
  int32_t dtype = 0;
  const DCol *col = nullptr;
};

// groups for `nk` (rows or pre-aggregated groups) + key values of the groups that are new
static BufP resolve_groups(sqlrs_hash_agg *a, const NKeys &nk, const uint64_t *row_ids,
                           const uint32_t *local_map, const std::vector<DCol> &kcols, int64_t *nnew_out) {
  Ctx *ctx = a->ctx;
  BufP new_rows;
  int64_t nnew = 0;
  BufP row_gid = agg_resolve_rows(ctx, a->st, nk, row_ids, (uint64_t)a->rows_seen, &new_rows, &nnew);
  if (nnew) { // group key values of first sight (hash_agg.rs:90-96)
    const void *idx = new_rows->p;
    BufP mapped;
    if (local_map) {
      mapped = ctx->alloc(4 * (size_t)nnew);
      gather_u32_kernel<<<dim3((unsigned)ceil_div(nnew, 256)), dim3(256), 0, ctx->stream>>>(
          local_map, new_rows->as<uint32_t>(), nnew, mapped->as<uint32_t>());
      SQ_HIP(hipGetLastError());
      idx = mapped->p;
    }
    for (size_t c = 0; c < kcols.size(); c++)
      a->key_parts[c].push_back(gather_column(ctx, kcols[c], idx, false, nullptr, nnew));
  }
  if (row_ids) a->in_order = false; // explicit ids arrive in arbitrary order
  *nnew_out = nnew;
  return row_gid;
}

// has-value tracking starts with the first NULL-bearing batch: until then every existing
// group holds at least one valid value, so its counter is back-filled with 1
static void start_tracking(Ctx *ctx, AggSpec &s, int64_t G, int64_t nnew) {
  if (s.track_nn) return;
  s.track_nn = true;
  s.nn.ensure(ctx, std::max<int64_t>(G, 1), 0);
  if (G - nnew > 0) fill_u64(ctx, s.nn.buf->as<uint64_t>(), G - nnew, 1);
}

// row route: accumulate the argument columns of n rows
// ---- MIN / MAX over Utf8 -----------------------------------------------------------------
__global__ void rank_of_perm_kernel(const uint32_t *__restrict__ perm, int64_t n, uint32_t *__restrict__ rank) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) rank[perm[i]] = (uint32_t)i;
}
// best[g] = min or max over the group's non-NULL rows of (rank + 1); 0 / ~0 = no value yet
__global__ void best_rank_kernel(const uint32_t *__restrict__ gid, const uint32_t *__restrict__ rank,
                                 const uint64_t *__restrict__ validity, int64_t n, int is_min,
                                 uint32_t *__restrict__ best) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (validity && !((validity[i >> 6] >> (i & 63)) & 1)) return;
  if (is_min) atomicMin(&best[gid[i]], rank[i] + 1);
  else atomicMax(&best[gid[i]], rank[i] + 1);
}
// winner row of every group (index into the concatenated strings) + whether the group has one
__global__ void winner_row_kernel(const uint32_t *__restrict__ best, const uint32_t *__restrict__ perm, int64_t G,
                                  int is_min, uint32_t *__restrict__ row, uint64_t *__restrict__ has) {
  int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  uint32_t b = g < G ? best[g] : 0;
  bool ok = g < G && b != (is_min ? 0xffffffffu : 0u);
  if (g < G) row[g] = ok ? perm[b - 1] : 0;
  uint64_t m = __ballot(ok);
  if (lane_id() == 0 && g < G) has[g >> 6] = m;
}

static DCol finalize_utf8_minmax(Ctx *ctx, AggSpec &s, int64_t G) {
  const int is_min = s.func == SQLRS_AGG_MIN;
  int64_t G1 = std::max<int64_t>(G, 1);
  // all retained strings as one column, their group ids as one array
  std::vector<const DCol *> parts;
  int64_t N = 0;
  for (auto &sb : s.strs) {
    parts.push_back(&sb.col);
    N += sb.n;
  }
  if (parts.empty() || N == 0) return make_null_column(ctx, SQLRS_UTF8, G);
  if (N > 0xfffffffell) fail(SQLRS_ERR_INTERNAL, "utf8 min/max: more than 2^32 rows");
  DCol all = concat_columns(ctx, parts);
  BufP gid = ctx->alloc(4 * (size_t)N);
  size_t off = 0;
  for (auto &sb : s.strs) {
    if (sb.n) SQ_HIP(hipMemcpyAsync(gid->as<uint8_t>() + off, sb.gid->p, 4 * (size_t)sb.n, hipMemcpyDeviceToDevice, ctx->stream));
    off += 4 * (size_t)sb.n;
  }
  const uint64_t *valid = (all.validity && all.null_count != 0) ? all.validity : nullptr;
  BufP perm = ctx->alloc(4 * (size_t)N), keys = ctx->alloc(8 * (size_t)N), rank = ctx->alloc(4 * (size_t)N);
  iota_u32(ctx, perm->as<uint32_t>(), N);
  sort_perm_by_utf8(ctx, all, valid, 0, perm->as<uint32_t>(), keys->as<uint64_t>(), N);
  dim3 b(256), gn((unsigned)ceil_div(N, 256));
  rank_of_perm_kernel<<<gn, b, 0, ctx->stream>>>(perm->as<uint32_t>(), N, rank->as<uint32_t>());
  BufP best = ctx->alloc(4 * (size_t)G1);
  SQ_HIP(hipMemsetAsync(best->p, is_min ? 0xff : 0, 4 * (size_t)G1, ctx->stream));
  best_rank_kernel<<<gn, b, 0, ctx->stream>>>(gid->as<uint32_t>(), rank->as<uint32_t>(), valid, N, is_min,
                                               best->as<uint32_t>());
  BufP row = ctx->alloc(4 * (size_t)G1), has = ctx->alloc_zero(bitmap_bytes(G1));
  int64_t G64 = (int64_t)round_up((size_t)G1, 64);
  winner_row_kernel<<<dim3((unsigned)ceil_div(G64, 256)), b, 0, ctx->stream>>>(
      best->as<uint32_t>(), perm->as<uint32_t>(), G, is_min, row->as<uint32_t>(), has->as<uint64_t>());
  SQ_HIP(hipGetLastError());
  DCol out = gather_column(ctx, all, row->p, false, has->as<uint64_t>(), G);
  out.dtype = SQLRS_UTF8;
  return out;
}

static void update_from_rows(sqlrs_hash_agg *a, const BufP &rgbuf, const std::vector<ArgView> &args,
                             int64_t n, int64_t nnew) {
  Ctx *ctx = a->ctx;
  int64_t G = a->st.ngroups;
  const uint32_t *rg = rgbuf->as<uint32_t>();
  for (AggSpec &s : a->aggs) {
    const ArgView &v = args[(size_t)s.argcol];
    if (v.dtype == SQLRS_UTF8 && (s.func == SQLRS_AGG_MIN || s.func == SQLRS_AGG_MAX)) {
      if (s.return_dtype != SQLRS_UTF8 || !v.col) fail(SQLRS_ERR_INTERNAL, "unsupported min_max scalar type");
      AggSpec::StrBatch sb;
      bool borrowed = (v.col->values && !v.col->own_values) || (v.col->validity && !v.col->own_validity) ||
                      (v.col->offsets && !v.col->own_offsets);
      sb.col = borrowed ? copy_column(ctx, *v.col) : *v.col; // the caller's batch may go away
      sb.gid = rgbuf;
      sb.n = n;
      s.strs.push_back(std::move(sb));
      continue;
    }
    if (s.func == SQLRS_AGG_COUNT) {
      s.nn.ensure(ctx, G, 0);
      agg_update_count(ctx, s.nn, rg, v.validity, nullptr, n);
      continue;
    }
    if (v.validity) start_tracking(ctx, s, G, nnew);
    if (s.track_nn) {
      s.nn.ensure(ctx, G, 0);
      agg_update_count(ctx, s.nn, rg, v.validity, nullptr, n);
    }
    if (s.func == SQLRS_AGG_SUM) {
      s.acc_dtype = s.return_dtype;
      s.acc.ensure(ctx, G, 0);
      agg_update_sum(ctx, s.acc, s.acc_dtype, rg, v.values, v.validity, n);
    } else {
      if (v.dtype != s.return_dtype) fail(SQLRS_ERR_INTERNAL, "unsupported min_max scalar type");
      s.acc_dtype = v.dtype;
      s.acc.ensure(ctx, G, acc_identity(s));
      agg_update_minmax(ctx, s.acc, s.acc_dtype, s.func == SQLRS_AGG_MIN, rg, v.values, v.validity, n);
    }
  }
}

// merge pre-aggregated groups into the table state: keys + first rows resolve like rows, the
// cells accumulate with weights (O(groups) global atomics)
static void merge_groups(sqlrs_hash_agg *a, PendingGroups &pg) {
  Ctx *ctx = a->ctx;
  PartAggOutput &po = pg.po;
  NKeys gk;
  gk.rows = po.groups;
  gk.keys = po.gkey;
  gk.exact = pg.exact;
  gk.dtype = pg.key_dtype;
  if (po.gvalid_bits) {
    gk.validity = po.gvalid_bits->as<uint64_t>();
    gk.own_validity = po.gvalid_bits;
  }
  int64_t nnew = 0;
  BufP rgid = resolve_groups(a, gk, part_row_ids(ctx, po), nullptr, pg.keyvals, &nnew);
  const uint32_t *rg = rgid->as<uint32_t>();
  int64_t G = a->st.ngroups, g = po.groups;
  auto cell = [&](int acc) { return po.gacc->as<uint64_t>() + (size_t)acc * (size_t)po.gcap; };
  for (size_t i = 0; i < a->aggs.size(); i++) {
    AggSpec &s = a->aggs[i];
    int col = pg.col_of_agg[i];
    int cc = pg.cnt_of_col[(size_t)col];
    if (s.func == SQLRS_AGG_COUNT) {
      s.nn.ensure(ctx, G, 0);
      agg_update_count(ctx, s.nn, rg, nullptr, (const int64_t *)cell(cc), g);
      continue;
    }
    if (pg.col_nullable[(size_t)col]) start_tracking(ctx, s, G, nnew);
    if (s.track_nn) {
      s.nn.ensure(ctx, G, 0);
      agg_update_count(ctx, s.nn, rg, nullptr, (const int64_t *)cell(cc), g);
    }
    int ac = pg.acc_of_agg[i];
    if (s.func == SQLRS_AGG_SUM) {
      s.acc_dtype = s.return_dtype;
      s.acc.ensure(ctx, G, 0);
      agg_update_sum(ctx, s.acc, s.acc_dtype, rg, cell(ac), nullptr, g);
    } else {
      s.acc_dtype = pg.col_dtype[(size_t)col];
      s.acc.ensure(ctx, G, acc_identity(s));
      agg_update_minmax(ctx, s.acc, SQLRS_UINT64 /* cells are ordered images */, s.func == SQLRS_AGG_MIN, rg,
                        cell(ac), nullptr, g);
    }
  }
}

static void flush_pending(sqlrs_hash_agg *a) {
  if (!a->pending.active) return;
  PendingGroups pg = std::move(a->pending);
  a->pending = PendingGroups();
  merge_groups(a, pg);
}

// finish() when one pre-aggregated batch is the whole input: emit its cells directly
static DBatch emit_pending(sqlrs_hash_agg *a) {
  Ctx *ctx = a->ctx;
  PendingGroups &pg = a->pending;
  PartAggOutput &po = pg.po;
  int64_t G = po.groups;
  DBatch o;
  o.rows = G;
  for (const DCol &k : pg.keyvals) o.cols.push_back(k);
  auto cell = [&](int acc) { return po.gacc->as<uint64_t>() + (size_t)acc * (size_t)po.gcap; };
  for (size_t i = 0; i < a->aggs.size(); i++) {
    AggSpec &s = a->aggs[i];
    int col = pg.col_of_agg[i];
    int cc = pg.cnt_of_col[(size_t)col];
    if (s.func == SQLRS_AGG_COUNT) {
      o.cols.push_back(agg_finalize_raw(ctx, s.func, SQLRS_INT64, cell(cc), nullptr, G, po.gacc));
      continue;
    }
    const uint64_t *nn = pg.col_nullable[(size_t)col] ? cell(cc) : nullptr;
    int32_t dt = s.func == SQLRS_AGG_SUM ? s.return_dtype : pg.col_dtype[(size_t)col];
    DCol c = agg_finalize_raw(ctx, s.func, dt, cell(pg.acc_of_agg[i]), nn, G, po.gacc);
    c.dtype = s.return_dtype;
    o.cols.push_back(c);
  }
  if (G > 1 && !a->any_order) { // bucket order -> first-seen order (hash_agg.rs:98,132)
    ProfScope ps(ctx, "agg_order_groups");
    // (the groups' first rows as they are — local u32 row numbers of the one batch, all below rows_seen < 2^bits)
    BufP perm = ctx->alloc(4 * (size_t)G);
    int bits = 1;
    while (bits < 32 && (1ull << bits) <= (uint64_t)std::max<int64_t>(a->rows_seen, 1)) bits++;
    radix_sort_index_u32(ctx, po.gfirst->as<uint32_t>(), G, bits, perm->as<uint32_t>());
    if (!gather_columns_packed(ctx, o.cols, G, perm->as<uint32_t>(), G))
      for (DCol &c : o.cols) c = gather_column(ctx, c, perm->p, false, nullptr, G);
  }
  return o;
}

// DISTINCT aggregates: distinct (keys..., arg) rows -> COUNT(*) / SUM(arg) per key group.  Both
// stages emit groups in first-seen order, so the column lines up with the main aggregation.
// DistinctCountAccumulator keeps NULL as one of the distinct values (count.rs:44-52); a distinct
// SUM skips it (sum.rs:64-85 via sum_result).
static DCol distinct_column(sqlrs_hash_agg *a, sqlrs_hash_agg::DistinctAgg &d, int64_t G) {
  Ctx *ctx = a->ctx;
  sqlrs_batch_t *dd = nullptr, *res = nullptr;
  int st = sqlrs_hash_agg_finish(d.dedup, SQLRS_MEM_DEVICE, &dd);
  if (st != SQLRS_OK) fail(st, ctx->last_error);
  const int nk = (int)a->group_by.size();
  std::vector<sqlrs_expr_node_t> nodes((size_t)nk + 1);
  std::vector<sqlrs_expr_t> gb((size_t)nk);
  std::memset(nodes.data(), 0, sizeof(sqlrs_expr_node_t) * nodes.size());
  for (int i = 0; i < nk; i++) {
    nodes[(size_t)i].op = SQLRS_EXPR_INPUT_REF;
    nodes[(size_t)i].index = i;
    gb[(size_t)i].nodes = &nodes[(size_t)i];
    gb[(size_t)i].num_nodes = 1;
    gb[(size_t)i].reserved = 0;
  }
  sqlrs_agg_func_t af;
  std::memset(&af, 0, sizeof(af));
  if (d.func == SQLRS_AGG_COUNT) {
    nodes[(size_t)nk].op = SQLRS_EXPR_CONSTANT; // COUNT of a never-NULL constant = number of distinct values
    nodes[(size_t)nk].dtype = SQLRS_INT64;
    nodes[(size_t)nk].i = 1;
    af.func = SQLRS_AGG_COUNT;
    af.return_dtype = SQLRS_INT64;
  } else {
    nodes[(size_t)nk].op = SQLRS_EXPR_INPUT_REF;
    nodes[(size_t)nk].index = nk;
    af.func = SQLRS_AGG_SUM;
    af.return_dtype = d.return_dtype;
  }
  af.arg.nodes = &nodes[(size_t)nk];
  af.arg.num_nodes = 1;
  sqlrs_hash_agg_t *b = nullptr;
  st = sqlrs_hash_agg_create((sqlrs_ctx_t *)ctx, nk, gb.data(), 1, &af, &b);
  if (st == SQLRS_OK) st = sqlrs_hash_agg_push(b, dd);
  if (st == SQLRS_OK) st = sqlrs_hash_agg_finish(b, SQLRS_MEM_DEVICE, &res);
  std::string err = ctx->last_error;
  if (b) sqlrs_hash_agg_destroy(b);
  DCol out;
  if (st == SQLRS_OK) {
    if (res->num_rows != G) {
      st = SQLRS_ERR_INTERNAL;
      err = "distinct aggregate produced a different number of groups";
    } else {
      const sqlrs_column_t &c = res->columns[nk];
      out.dtype = c.dtype;
      out.length = G;
      size_t w = width_of(c.dtype);
      out.own_values = ctx->alloc(w * (size_t)std::max<int64_t>(G, 1));
      out.values = out.own_values->p;
      if (G) SQ_HIP(hipMemcpyAsync(out.own_values->p, c.values, w * (size_t)G, hipMemcpyDeviceToDevice, ctx->stream));
      if (c.validity && c.null_count != 0) {
        out.own_validity = ctx->alloc(bitmap_bytes(std::max<int64_t>(G, 1)));
        SQ_HIP(hipMemcpyAsync(out.own_validity->p, c.validity, bitmap_bytes(G), hipMemcpyDeviceToDevice, ctx->stream));
        out.validity = out.own_validity->as<uint64_t>();
        out.null_count = c.null_count;
      }
      ctx->sync();
    }
  }
  if (dd) sqlrs_batch_release(dd);
  if (res) sqlrs_batch_release(res);
  if (st != SQLRS_OK) fail(st, err);
  return out;
}

// [keys..., plain aggregates...] -> [keys..., aggregates in declaration order]
static void place_aggregate_columns(sqlrs_hash_agg *a, DBatch &o) {
  if (a->distinct_aggs.empty()) return;
  const size_t nk = a->group_by.size();
  std::vector<DCol> cols(o.cols.begin(), o.cols.begin() + (long)nk);
  for (auto &e : a->out_order) {
    if (e.first == 0)
      cols.push_back(o.cols[nk + (size_t)e.second]);
    else
      cols.push_back(distinct_column(a, a->distinct_aggs[(size_t)e.second], o.rows));
  }
  o.cols = std::move(cols);
}

// evaluates the distinct (expression, cast) argument columns of the aggregates; `shift` is
// subtracted from every InputRef (fused join: join-output index -> probe batch index)
static std::vector<DCol> eval_arg_columns(sqlrs_hash_agg *a, const std::function<const DCol &(int)> &colfn,
                                          int64_t n, int shift) {
  Ctx *ctx = a->ctx;
  std::vector<DCol> acols;
  for (size_t k = 0; k < a->arg_exprs.size(); k++) {
    Expr e = a->arg_exprs[k];
    if (shift)
      for (auto &nd : e.nodes)
        if (nd.op == SQLRS_EXPR_INPUT_REF) nd.index -= shift;
    DCol c = eval_expr(ctx, e, colfn, n, true);
    if (a->arg_cast[k] && c.dtype != a->arg_cast[k]) { // SumAccumulator casts first (sum.rs:54)
      sqlrs_expr_node_t cn[2];
      std::memset(cn, 0, sizeof(cn));
      cn[0].op = SQLRS_EXPR_INPUT_REF;
      cn[1].op = SQLRS_EXPR_TYPE_CAST;
      cn[1].dtype = a->arg_cast[k];
      Expr ce;
      ce.nodes.assign(cn, cn + 2);
      ce.strings.resize(2);
      auto one = [&](int) -> const DCol & { return c; };
      DCol casted = eval_expr(ctx, ce, one, n, true);
      c = casted;
    }
    acols.push_back(c);
  }
  return acols;
}

struct JoinSide {
  const uint64_t *keys;     // normalised build keys of an Inner join with unique build keys
  const uint64_t *validity;
  int64_t n;
  PartitionedRows *cache;
  bool unique_known = true; // false: uniqueness of the build keys not established yet (agg_partition.hpp)
  bool range_known = false; // omin / omax: signed-order images of the smallest / largest valid key
  uint64_t omin = 0, omax = 0;
  const uint64_t *bits = nullptr; // existence bitmap over [omin, omax] (PartAggInput::join_bits)
  const uint32_t *mult = nullptr; // duplicate build keys: build rows per key of [omin, omax] (PartAggInput::join_mult)
};

// Consumes one batch given its evaluated key / argument columns: partition route when it
// applies, else the row route.  With `js` (fused join) only rows whose key has a build partner
// count; returns false if the fused route is not applicable (nothing has been consumed then).
static bool agg_consume(sqlrs_hash_agg *a, int64_t n, const std::vector<DCol> &kcols, const NKeys &nk,
                        const std::vector<DCol> &acols, const JoinSide *js, const RowFilter *rf = nullptr) {
  Ctx *ctx = a->ctx;
    auto views_of = [&](const std::vector<DCol> &cols) {
      std::vector<ArgView> v;
      for (const DCol &c : cols) {
        ArgView x;
        x.values = c.values;
        x.validity = (c.validity && c.null_count != 0) ? c.validity : nullptr;
        x.dtype = c.dtype;
        x.col = &c;
        v.push_back(x);
      }
      return v;
    };

    // identical argument expressions that evaluate to the same dtype share one column
    // (COUNT(val) and SUM(val): one read of val)
    std::vector<DCol> pcols;
    std::vector<int> pidx(acols.size(), -1);
    for (size_t k = 0; k < acols.size(); k++) {
      for (size_t j = 0; j < k && pidx[k] < 0; j++)
        if (same_expr(a->arg_exprs[j], a->arg_exprs[k]) && acols[j].dtype == acols[k].dtype) pidx[k] = pidx[j];
      if (pidx[k] < 0) {
        pidx[k] = (int)pcols.size();
        pcols.push_back(acols[k]);
      }
    }
    // ---- partition route ----------------------------------------------------------
    bool done = false;
    static const int64_t PART_MIN_ROWS = [] {
      const char *e = hook("SQLRS_PART_MIN_ROWS"); // test hook: force the partition route
      return e ? std::atoll(e) : (1ll << 21);
    }();
    if ((n >= PART_MIN_ROWS || js) && pcols.size() <= 2) {
      PartAggSpec spec;
      bool ok = true;
      std::vector<int> acc_of_agg(a->aggs.size(), -1), cnt_of_col(pcols.size(), -1);
      for (const DCol &c : pcols)
        ok &= (c.dtype == SQLRS_INT64 || c.dtype == SQLRS_FLOAT64); // 8-byte values only
      // a COUNT cell per column whose has-value state must be known
      for (size_t k = 0; ok && k < pcols.size(); k++) {
        bool need = pcols[k].validity && pcols[k].null_count != 0;
        for (const AggSpec &s : a->aggs)
          if (pidx[(size_t)s.argcol] == (int)k && (s.func == SQLRS_AGG_COUNT || s.track_nn)) need = true;
        if (need) {
          if (spec.n_acc >= PART_MAX_ACC) { ok = false; break; }
          cnt_of_col[k] = spec.n_acc;
          spec.op[spec.n_acc] = PART_COUNT;
          spec.src[spec.n_acc++] = (int)k;
        }
      }
      for (size_t i = 0; ok && i < a->aggs.size(); i++) {
        const AggSpec &s = a->aggs[i];
        if (s.func == SQLRS_AGG_COUNT) {
          acc_of_agg[i] = cnt_of_col[(size_t)pidx[(size_t)s.argcol]];
          continue;
        }
        if (spec.n_acc >= PART_MAX_ACC) { ok = false; break; }
        const DCol &c = pcols[(size_t)pidx[(size_t)s.argcol]];
        if (s.func != SQLRS_AGG_SUM && c.dtype != s.return_dtype) { ok = false; break; }
        acc_of_agg[i] = spec.n_acc;
        spec.op[spec.n_acc] = s.func == SQLRS_AGG_SUM ? (c.dtype == SQLRS_FLOAT64 ? PART_SUM_F64 : PART_SUM_I64)
                              : s.func == SQLRS_AGG_MIN ? PART_MIN : PART_MAX;
        spec.kind[spec.n_acc] = c.dtype == SQLRS_FLOAT64 ? 1 : 0;
        spec.src[spec.n_acc++] = pidx[(size_t)s.argcol];
      }
      spec.nv = (int)pcols.size();
      if (ok) {
        PartAggInput pin;
        pin.keys = nk.keys->as<uint64_t>();
        pin.key_validity = nk.validity;
        pin.n = n;
        for (size_t k = 0; k < pcols.size(); k++) {
          pin.vals[k] = pcols[k].values;
          pin.val_validity[k] = (pcols[k].validity && pcols[k].null_count != 0) ? pcols[k].validity : nullptr;
        }
        if (js) {
          pin.join_keys = js->keys;
          pin.join_validity = js->validity;
          pin.join_n = js->n;
          pin.join_cache = js->cache;
          pin.join_range_known = js->range_known;
          pin.join_unique_known = js->unique_known;
          pin.join_omin = js->omin;
          pin.join_omax = js->omax;
          pin.join_bits = js->bits;
          pin.join_mult = js->mult;
        }
        if (rf) pin.filter = *rf;
        PartAggOutput po;
        flush_pending(a); // an older deferred batch must be in the table before this one
        bool part_ok = partitioned_preaggregate(ctx, spec, pin, (uint64_t)a->rows_seen, &po);
        if (!part_ok && po.retry_exact) { // optimistic key statistics (sampled range) did not hold: exact pass, once
          pin.exact_stats = true;
          po = PartAggOutput();
          part_ok = partitioned_preaggregate(ctx, spec, pin, (uint64_t)a->rows_seen, &po);
        }
        if (part_ok) {
          PendingGroups pg;
          pg.active = true;
          pg.exact = nk.exact;
          pg.key_dtype = nk.dtype;
          pg.cnt_of_col = cnt_of_col;
          pg.acc_of_agg = acc_of_agg;
          for (const AggSpec &s : a->aggs) pg.col_of_agg.push_back(pidx[(size_t)s.argcol]);
          for (size_t k = 0; k < pcols.size(); k++) {
            pg.col_nullable.push_back(pin.val_validity[k] != nullptr);
            pg.col_dtype.push_back(pcols[k].dtype);
          }
          // key values of every group of the batch (hash_agg.rs:90-96).  One exactly-compared 8-byte
          // key column: the normalised key IS the value (int64 / f64 bit pattern), no gather needed
          // (a random gather of 1e7 keys costs 0.25 ms)
          if (nk.exact && kcols.size() == 1 && (kcols[0].dtype == SQLRS_INT64 || kcols[0].dtype == SQLRS_FLOAT64) &&
              kcols[0].dtype == nk.dtype) {
            DCol kv;
            kv.dtype = kcols[0].dtype;
            kv.length = po.groups;
            kv.values = po.gkey->p;
            kv.own_values = po.gkey;
            if (po.gvalid_bits) {
              kv.validity = po.gvalid_bits->as<uint64_t>();
              kv.own_validity = po.gvalid_bits;
              kv.null_count = -1;
            }
            pg.keyvals.push_back(kv);
          } else {
            for (const DCol &kc : kcols)
              pg.keyvals.push_back(gather_column(ctx, kc, po.gfirst->p, false, nullptr, po.groups));
          }
          pg.po = po;
          if (a->st.ngroups == 0 && po.n_overflow == 0 && !po.may_dup)
            a->pending = std::move(pg); // nothing to merge with yet: defer building the table
          else
            merge_groups(a, pg);
          // rows whose bucket table was full go through the row route
          if (po.n_overflow) {
            int64_t m = po.n_overflow;
            DCol kc;
            kc.dtype = SQLRS_UINT64;
            kc.length = n;
            kc.values = nk.keys->p;
            kc.validity = nk.validity;
            kc.null_count = nk.validity ? -1 : 0;
            DCol sub = gather_column(ctx, kc, po.ov_rows->p, false, nullptr, m);
            NKeys ok_;
            ok_.rows = m;
            ok_.keys = sub.own_values;
            ok_.exact = nk.exact;
            ok_.dtype = nk.dtype;
            if (sub.validity) {
              ok_.validity = sub.validity;
              ok_.own_validity = sub.own_validity;
            }
            BufP rid = ctx->alloc(8 * (size_t)m);
            rowid_from_u32_kernel<<<dim3((unsigned)ceil_div(m, 256)), dim3(256), 0, ctx->stream>>>(
                po.ov_rows->as<uint32_t>(), m, (uint64_t)a->rows_seen, rid->as<uint64_t>());
            SQ_HIP(hipGetLastError());
            std::vector<DCol> sub_args;
            for (const DCol &c : acols) sub_args.push_back(gather_column(ctx, c, po.ov_rows->p, false, nullptr, m));
            int64_t nn2 = 0;
            BufP rg2 = resolve_groups(a, ok_, rid->as<uint64_t>(), po.ov_rows->as<uint32_t>(), kcols, &nn2);
            for (DCol &c : sub_args)
              if (c.validity && c.null_count < 0) c.null_count = count_nulls(ctx, c);
            update_from_rows(a, rg2, views_of(sub_args), m, nn2);
          }
          done = true;
        }
      }
    }
    // ---- row route ------------------------------------------------------------------
    if (!done && (js || rf)) return false; // fused join: the caller composes (filter +) join + aggregate instead
    if (!done) {
      flush_pending(a);
      int64_t nnew = 0;
      BufP row_gid = resolve_groups(a, nk, nullptr, nullptr, kcols, &nnew); // 3.2 (:85-110)
      update_from_rows(a, row_gid, views_of(acols), n, nnew); // 4. (:113-121)
    }
    return true;
}

// a first batch this large is aggregated in place; staged rows that trigger an aggregation before
// finish (SQLRS_STAGE_DIRECT_ROWS / SQLRS_STAGE_FLUSH_ROWS: test hooks, read once)
static const int64_t STAGE_DIRECT_ROWS = [] {
  const char *e = hook("SQLRS_STAGE_DIRECT_ROWS");
  return e ? std::atoll(e) : (1ll << 26);
}();
static const int64_t STAGE_FLUSH_ROWS = [] {
  const char *e = hook("SQLRS_STAGE_FLUSH_ROWS");
  return e ? std::atoll(e) : (1ll << 28);
}();

// a column the operator may keep after the call returns: scalars materialised, borrowed buffers copied
static DCol own_column(Ctx *ctx, const DCol &c, int64_t n) {
  DCol m = c.stride == 0 ? materialize_scalar(ctx, c, n) : c;
  bool borrowed = (m.values && !m.own_values) || (m.validity && !m.own_validity) || (m.offsets && !m.own_offsets);
  return borrowed ? copy_column(ctx, m) : m;
}

// aggregate everything that is staged as one batch (arrival order = row order)
static void flush_staged(sqlrs_hash_agg *a) {
  if (a->staged.empty()) return;
  Ctx *ctx = a->ctx;
  std::vector<sqlrs_hash_agg::Staged> st = std::move(a->staged);
  a->staged.clear();
  int64_t n = a->staged_rows;
  a->staged_rows = 0;
  std::vector<DCol> kcols, acols;
  if (st.size() == 1) {
    kcols = std::move(st[0].kcols);
    acols = std::move(st[0].acols);
  } else {
    for (size_t c = 0; c < st[0].kcols.size(); c++) {
      std::vector<const DCol *> parts;
      for (auto &b : st) parts.push_back(&b.kcols[c]);
      kcols.push_back(concat_columns(ctx, parts));
    }
    for (size_t c = 0; c < st[0].acols.size(); c++) {
      std::vector<const DCol *> parts;
      for (auto &b : st) parts.push_back(&b.acols[c]);
      acols.push_back(concat_columns(ctx, parts));
    }
  }
  NKeys nk = a->strong_keys ? normalize_keys_strong(ctx, kcols, n) : normalize_keys(ctx, kcols, n);
  agg_consume(a, n, kcols, nk, acols, nullptr);
  a->rows_seen += n;
}

} // namespace sq

extern "C" {

static int hash_agg_create_impl(sqlrs_ctx_t *ctx, int num_group_by, const sqlrs_expr_t *group_by, int num_aggs,
                                const sqlrs_agg_func_t *aggs, bool allow_split, sqlrs_hash_agg_t **out);
int sqlrs_hash_agg_create(sqlrs_ctx_t *ctx, int num_group_by, const sqlrs_expr_t *group_by,
                          int num_aggs, const sqlrs_agg_func_t *aggs, sqlrs_hash_agg_t **out) {
  return hash_agg_create_impl(ctx, num_group_by, group_by, num_aggs, aggs, true, out);
}

// Parts of a wide aggregate list (sqlrs_hash_agg::parts): the aggregates of ONE argument expression per part (first
// fit, at most PART_MAX_ACC cells: a COUNT cell — the column may turn out nullable — plus one per SUM / MIN / MAX).
// One column, not two: a part with a single 8-byte argument travels as 16-byte packed records through a one-level
// claimed partition when its keys are dense integers (C4 with three argument columns: parts of 2 + 1 columns 10.0 ms,
// the two-column part alone 7 ms of two-level column-form passes).
static void hash_agg_plan_parts(sqlrs_hash_agg *a, sqlrs_ctx_t *ctx, int num_group_by, const sqlrs_expr_t *group_by,
                                const sqlrs_agg_func_t *aggs) {
  const char *env = hook("SQLRS_AGG_SPLIT"); // test / tuning hook, read per create: 0 = never
  if (env && std::atoi(env) == 0) return;
  if (!a->distinct_aggs.empty() || a->aggs.size() < 3) return;
  std::vector<int> col_of((size_t)a->aggs.size(), -1); // distinct argument expression of every aggregate (casts aside)
  int ncols = 0;
  for (size_t i = 0; i < a->aggs.size(); i++) {
    if (a->aggs[i].return_dtype == SQLRS_UTF8) return; // (string MIN / MAX keeps its batches: row route anyway)
    for (size_t k = 0; k < i && col_of[i] < 0; k++)
      if (same_expr(a->aggs[k].arg, a->aggs[i].arg)) col_of[i] = col_of[k];
    if (col_of[i] < 0) col_of[i] = ncols++;
  }
  struct Part {
    std::vector<int> cols, aggs;
    int cells = 0;
  };
  std::vector<Part> plan;
  for (size_t i = 0; i < a->aggs.size(); i++) {
    const int c = col_of[i], own = a->aggs[i].func != SQLRS_AGG_COUNT ? 1 : 0;
    size_t at = plan.size();
    for (size_t q = 0; q < plan.size() && at == plan.size(); q++) {
      const bool has = std::find(plan[q].cols.begin(), plan[q].cols.end(), c) != plan[q].cols.end();
      if (has && plan[q].cells + own <= PART_MAX_ACC) at = q;
    }
    if (at == plan.size()) plan.emplace_back();
    Part &pt = plan[at];
    if (std::find(pt.cols.begin(), pt.cols.end(), c) == pt.cols.end()) {
      pt.cols.push_back(c);
      pt.cells++;
    }
    pt.cells += own;
    pt.aggs.push_back((int)i);
  }
  if (plan.size() < 2) return;
  if (ncols <= 2) { // two argument columns within PART_MAX_ACC cells fit one partition-route operator as they are
    int cells = ncols;
    for (const AggSpec &sp : a->aggs) cells += sp.func != SQLRS_AGG_COUNT ? 1 : 0;
    if (cells <= PART_MAX_ACC) return;
  }
  a->part_col.assign(a->aggs.size(), {0, 0});
  for (size_t q = 0; q < plan.size(); q++) {
    std::vector<sqlrs_agg_func_t> sub;
    for (size_t k = 0; k < plan[q].aggs.size(); k++) {
      sub.push_back(aggs[plan[q].aggs[k]]); // (no DISTINCT aggregate here: aggregate i of `a->aggs` is aggs[i])
      a->part_col[(size_t)plan[q].aggs[k]] = {(int)q, (int)k};
    }
    sqlrs_hash_agg_t *p = nullptr;
    int st = hash_agg_create_impl(ctx, num_group_by, group_by, (int)sub.size(), sub.data(), false, &p);
    if (st != SQLRS_OK) fail(st, ctx->last_error);
    a->parts.push_back(p);
  }
}

static int hash_agg_create_impl(sqlrs_ctx_t *ctx, int num_group_by, const sqlrs_expr_t *group_by, int num_aggs,
                                const sqlrs_agg_func_t *aggs, bool allow_split, sqlrs_hash_agg_t **out) {
  return guard(ctx, [&] {
    if (num_group_by < 1) // PhysicalRewriter only builds HashAgg with keys (physical_rewriter.rs:49-62)
      fail(SQLRS_ERR_INTERNAL, "HashAgg needs at least one group-by expression");
    auto a = std::unique_ptr<sqlrs_hash_agg>(new sqlrs_hash_agg());
    a->ctx = ctx;
    for (int i = 0; i < num_group_by; i++) a->group_by.push_back(expr_from_abi(&group_by[i]));
    for (int i = 0; i < num_aggs; i++) {
      AggSpec s;
      s.func = aggs[i].func;
      s.distinct = aggs[i].distinct;
      s.return_dtype = aggs[i].return_dtype;
      s.arg = expr_from_abi(&aggs[i].arg);
      if (s.func < SQLRS_AGG_COUNT || s.func > SQLRS_AGG_MAX)
        fail(SQLRS_ERR_INTERNAL, "unknown aggregate function");
      if (s.distinct && (s.func == SQLRS_AGG_COUNT || s.func == SQLRS_AGG_SUM)) {
        if (s.func == SQLRS_AGG_SUM && s.return_dtype != SQLRS_INT64 && s.return_dtype != SQLRS_FLOAT64)
          fail(SQLRS_ERR_INTERNAL, "not expected types for sum");
        sqlrs_hash_agg::DistinctAgg d;
        d.func = s.func;
        d.return_dtype = s.return_dtype;
        d.dedup = new sqlrs_hash_agg();
        d.dedup->ctx = ctx;
        d.dedup->strong_keys = true;
        d.dedup->group_by = a->group_by;
        d.dedup->group_by.push_back(s.arg);
        d.dedup->key_parts.resize(d.dedup->group_by.size());
        a->out_order.emplace_back(1, (int)a->distinct_aggs.size());
        a->distinct_aggs.push_back(d);
        continue;
      }
      a->out_order.emplace_back(0, (int)a->aggs.size());
      // SumAccumulator casts its input to the return type (sum.rs:54); the reference's
      // sum_result has no (Int32, Int32) arm (sum.rs:64-85)
      if (s.func == SQLRS_AGG_SUM && s.return_dtype != SQLRS_INT64 && s.return_dtype != SQLRS_FLOAT64)
        fail(SQLRS_ERR_INTERNAL, "not expected types for sum");
      int32_t cast = s.func == SQLRS_AGG_SUM ? s.return_dtype : 0;
      int found = -1;
      for (size_t k = 0; k < a->arg_exprs.size(); k++)
        if (a->arg_cast[k] == cast && same_expr(a->arg_exprs[k], s.arg)) found = (int)k;
      if (found < 0) {
        a->arg_exprs.push_back(s.arg);
        a->arg_cast.push_back(cast);
        found = (int)a->arg_exprs.size() - 1;
      }
      s.argcol = found;
      a->aggs.push_back(std::move(s));
    }
    a->key_parts.resize((size_t)num_group_by);
    if (allow_split) hash_agg_plan_parts(a.get(), ctx, num_group_by, group_by, aggs);
    *out = a.release();
  });
}

static int hash_agg_push_device(sqlrs_hash_agg_t *a, const sqlrs_batch_t *in, bool filtered = false);
// what is staged on the host -> one device batch -> the operator
static int hash_agg_flush_host(sqlrs_hash_agg_t *a) {
  if (!a->hstage.has_schema) return SQLRS_OK;
  sqlrs_batch_t *dev = nullptr;
  int st = guard(a->ctx, [&] {
    SQ_HIP(hipSetDevice(a->ctx->device));
    dev = a->hstage.take();
  });
  if (st != SQLRS_OK) return st;
  st = hash_agg_push_device(a, dev);
  sqlrs_batch_release(dev);
  return st;
}

// one iteration of the for_await loop  [ref: hash_agg.rs:44-122]
int sqlrs_hash_agg_push(sqlrs_hash_agg_t *a, const sqlrs_batch_t *in) {
  if (!a->parts.empty()) a->saw_batch = true; // (wide aggregate list: staged here, handed to the parts by hash_agg_push_device)
  a->hstage.ctx = a->ctx;
  if (a->hstage.accepts(in)) {
    int st = guard(a->ctx, [&] { a->hstage.append(in); });
    if (st != SQLRS_OK || a->hstage.rows < HOST_STAGE_FLUSH_ROWS) return st;
    return hash_agg_flush_host(a);
  }
  int st = hash_agg_flush_host(a); // keeps the arrival order of the rows
  return st != SQLRS_OK ? st : hash_agg_push_device(a, in);
}

// A filter that the first partition pass can evaluate (col OP constant, radix_part.hpp) on a batch that is aggregated
// in place: keys and arguments are evaluated on the UNFILTERED batch, so nothing among them may raise for a row the
// filter would have dropped (a division: ADVICE r2) — then the Filter operator runs first.
static bool hash_agg_try_fused_filter(sqlrs_hash_agg_t *a, const sqlrs_batch_t *in) {
  Ctx *ctx = a->ctx;
  const int64_t n = in->num_rows;
  if (!a->staged.empty() || n < STAGE_DIRECT_ROWS || !a->distinct_aggs.empty()) return false;
  for (const Expr &e : a->group_by)
    for (const sqlrs_expr_node_t &nd : e.nodes)
      if (nd.op == SQLRS_EXPR_DIVIDE) return false;
  for (const Expr &e : a->arg_exprs)
    for (const sqlrs_expr_node_t &nd : e.nodes)
      if (nd.op == SQLRS_EXPR_DIVIDE) return false;
  InBatch ib(ctx, in);
  RowFilter rf;
  if (!fusable_row_filter(a->filter, ib, &rf)) return false;
  auto colfn = [&](int i) -> const DCol & { return ib.col(i); };
  std::vector<DCol> kcols;
  for (const Expr &e : a->group_by) kcols.push_back(eval_expr(ctx, e, colfn, n, true));
  std::vector<DCol> acols = eval_arg_columns(a, colfn, n, 0);
  NKeys nk = a->strong_keys ? normalize_keys_strong(ctx, kcols, n) : normalize_keys(ctx, kcols, n);
  if (!agg_consume(a, n, kcols, nk, acols, nullptr, &rf)) return false; // (nothing consumed)
  if (!a->saw_batch) {
    a->saw_batch = true;
    for (const DCol &k : kcols) a->key_dtypes.push_back(k.dtype);
  }
  a->rows_seen += n;
  a->filter_fused_batches++;
  return true;
}

// Wide aggregate list: every part sees every batch — ONE device copy of it.  A HOST batch is uploaded here once (the parts
// would each upload it again), and a filter the parts cannot evaluate inside their first partition pass runs here once,
// the parts taking the kept rows as they are; a fusable filter stays with the parts (each reads the filter column in the
// pass that reads its keys anyway: nothing is compacted at all).
static bool any_host_column(const sqlrs_batch_t *b) {
  for (int c = 0; c < b->num_columns; c++)
    if (b->columns[c].mem == SQLRS_MEM_HOST && b->columns[c].length > 0) return true;
  return false;
}
static int hash_agg_push_parts(sqlrs_hash_agg_t *a, const sqlrs_batch_t *in0) {
  const sqlrs_batch_t *in = in0;
  sqlrs_batch_t *dev = nullptr, *kept = nullptr;
  bool filtered = false;
  int st = SQLRS_OK;
  if (a->has_filter) {
    bool fusable = false;
    st = guard(a->ctx, [&] {
      SQ_HIP(hipSetDevice(a->ctx->device));
      if (any_host_column(in0) || in0->num_rows < STAGE_DIRECT_ROWS) return;
      InBatch ib(a->ctx, in0);
      RowFilter rf;
      fusable = fusable_row_filter(a->filter, ib, &rf);
    });
    if (st != SQLRS_OK) return st;
    if (!fusable) {
      std::vector<sqlrs_expr_node_t> nodes = a->filter.nodes;
      for (size_t i = 0; i < nodes.size(); i++) nodes[i].s = a->filter.strings[i].empty() ? nullptr : a->filter.strings[i].c_str();
      sqlrs_expr_t fe{nodes.data(), (int32_t)nodes.size(), 0};
      sqlrs_filter_t *f = nullptr;
      st = sqlrs_filter_create((sqlrs_ctx_t *)a->ctx, &fe, &f);
      if (st != SQLRS_OK) return st;
      st = sqlrs_filter_push(f, in0, SQLRS_MEM_DEVICE, &kept);
      sqlrs_filter_destroy(f);
      if (st != SQLRS_OK) return st;
      in = kept;
      filtered = true;
    }
  }
  if (any_host_column(in)) {
    st = guard(a->ctx, [&] {
      SQ_HIP(hipSetDevice(a->ctx->device));
      InBatch ib(a->ctx, in);
      dev = emit_batch(a->ctx, ib.materialize(true), SQLRS_MEM_DEVICE);
    });
    if (st != SQLRS_OK) return st;
    in = dev;
  }
  for (sqlrs_hash_agg *p : a->parts) {
    st = hash_agg_push_device(p, in, filtered);
    if (st != SQLRS_OK) {
      a->ctx->last_error = p->ctx->last_error;
      break;
    }
  }
  if (kept) sqlrs_batch_release(kept);
  if (dev) sqlrs_batch_release(dev);
  return st;
}

static int hash_agg_push_device(sqlrs_hash_agg_t *a, const sqlrs_batch_t *in0, bool filtered) {
  if (!a->parts.empty()) return hash_agg_push_parts(a, in0);
  const sqlrs_batch_t *in = in0;
  sqlrs_batch_t *kept = nullptr;
  if (a->has_filter && !filtered) { // [ref: filter.rs:13-25 feeding hash_agg.rs:44]
    bool fused = false;
    int st = guard(a->ctx, [&] {
      SQ_HIP(hipSetDevice(a->ctx->device));
      fused = hash_agg_try_fused_filter(a, in0);
    });
    if (st != SQLRS_OK || fused) return st;
    std::vector<sqlrs_expr_node_t> nodes = a->filter.nodes;
    for (size_t i = 0; i < nodes.size(); i++) nodes[i].s = a->filter.strings[i].empty() ? nullptr : a->filter.strings[i].c_str();
    sqlrs_expr_t fe{nodes.data(), (int32_t)nodes.size(), 0};
    sqlrs_filter_t *f = nullptr;
    st = sqlrs_filter_create((sqlrs_ctx_t *)a->ctx, &fe, &f);
    if (st != SQLRS_OK) return st;
    st = sqlrs_filter_push(f, in0, SQLRS_MEM_DEVICE, &kept);
    sqlrs_filter_destroy(f);
    if (st != SQLRS_OK) return st;
    in = kept;
  }
  int rc = guard(a->ctx, [&] {
    Ctx *ctx = a->ctx;
    SQ_HIP(hipSetDevice(ctx->device));
    InBatch ib(ctx, in);
    auto colfn = [&](int i) -> const DCol & { return ib.col(i); };
    int64_t n = ib.rows();
    // 2.2 group key columns (:69-73) and 3.1 their per-row key (:76-77)
    std::vector<DCol> kcols;
    for (const Expr &e : a->group_by) kcols.push_back(eval_expr(ctx, e, colfn, n, true));
    if (!a->saw_batch) {
      a->saw_batch = true;
      for (const DCol &k : kcols) a->key_dtypes.push_back(k.dtype);
    }
    // 2.1 argument columns (:63-66), evaluated once per distinct (expression, cast)
    std::vector<DCol> acols = eval_arg_columns(a, colfn, n, 0);
    // A first batch that is large by itself is aggregated in place (no copy of 16 B/row); everything
    // else is staged and aggregated together at finish (or every STAGE_FLUSH_ROWS rows).
    if (a->staged.empty() && n >= STAGE_DIRECT_ROWS) {
      NKeys nk = a->strong_keys ? normalize_keys_strong(ctx, kcols, n) : normalize_keys(ctx, kcols, n);
      agg_consume(a, n, kcols, nk, acols, nullptr);
      a->rows_seen += n;
    } else if (n > 0) {
      sqlrs_hash_agg::Staged sb;
      sb.n = n;
      for (const DCol &c : kcols) sb.kcols.push_back(own_column(ctx, c, n));
      for (const DCol &c : acols) sb.acols.push_back(own_column(ctx, c, n));
      a->staged.push_back(std::move(sb));
      a->staged_rows += n;
      if (a->staged_rows >= STAGE_FLUSH_ROWS) flush_staged(a);
    }
    for (auto &d : a->distinct_aggs) {
      int st = sqlrs_hash_agg_push(d.dedup, in);
      if (st != SQLRS_OK) fail(st, ctx->last_error);
    }
  });
  if (kept) sqlrs_batch_release(kept);
  return rc;
}

// [ref: hash_agg.rs:124-149]
static DBatch hash_agg_finish_device(sqlrs_hash_agg_t *a);
int sqlrs_hash_agg_finish(sqlrs_hash_agg_t *a, int out_mem, sqlrs_batch_t **out) {
  if (!a->parts.empty()) { // the parts' group columns are identical (same keys, same first-seen order): side by side
    int stp = hash_agg_flush_host(a); // (small host batches are staged in the parent)
    if (stp != SQLRS_OK) return stp;
    return guard(a->ctx, [&] {
      Ctx *ctx = a->ctx;
      SQ_HIP(hipSetDevice(ctx->device));
      std::vector<DBatch> outs;
      for (sqlrs_hash_agg *p : a->parts) {
        outs.push_back(hash_agg_finish_device(p));
        if (outs.back().rows != outs[0].rows) fail(SQLRS_ERR_INTERNAL, "parts of a wide aggregate list disagree on the groups");
      }
      const size_t nk = a->group_by.size();
      DBatch o;
      o.rows = outs[0].rows;
      for (size_t c = 0; c < nk; c++) o.cols.push_back(outs[0].cols[c]);
      for (const auto &pc : a->part_col) o.cols.push_back(outs[(size_t)pc.first].cols[nk + (size_t)pc.second]);
      *out = emit_batch(ctx, std::move(o), out_mem);
    });
  }
  int stf = hash_agg_flush_host(a);
  if (stf != SQLRS_OK) return stf;
  return guard(a->ctx, [&] {
    SQ_HIP(hipSetDevice(a->ctx->device));
    *out = emit_batch(a->ctx, hash_agg_finish_device(a), out_mem);
  });
}
// (host-staged batches are flushed by the caller; throws)
static DBatch hash_agg_finish_device(sqlrs_hash_agg_t *a) {
  {
    Ctx *ctx = a->ctx;
    if (!a->saw_batch) // group_and_agg_fields.unwrap() panics on None (:125)
      fail(SQLRS_ERR_INTERNAL, "hash agg finished without any input batch");
    flush_staged(a);
    if (a->pending.active && a->st.ngroups == 0) {
      DBatch pb = emit_pending(a);
      place_aggregate_columns(a, pb);
      return pb;
    }
    flush_pending(a);
    int64_t G = a->st.ngroups;
    DBatch o;
    o.rows = G;
    for (size_t c = 0; c < a->key_parts.size(); c++) {
      if (a->key_parts[c].empty()) {
        DCol e = make_null_column(ctx, a->key_dtypes[c], 0);
        e.null_count = 0;
        o.cols.push_back(e);
        continue;
      }
      std::vector<const DCol *> parts;
      for (const DCol &p : a->key_parts[c]) parts.push_back(&p);
      o.cols.push_back(concat_columns(ctx, parts));
    }
    for (AggSpec &s : a->aggs) {
      if (s.func == SQLRS_AGG_COUNT) {
        s.nn.ensure(ctx, std::max<int64_t>(G, 1), 0);
        o.cols.push_back(agg_finalize_values(ctx, s.func, SQLRS_INT64, s.nn, nullptr, G));
        continue;
      }
      if (!s.strs.empty() || s.return_dtype == SQLRS_UTF8) {
        o.cols.push_back(finalize_utf8_minmax(ctx, s, G));
        continue;
      }
      int32_t dt = s.acc_dtype ? s.acc_dtype : s.return_dtype;
      s.acc.ensure(ctx, std::max<int64_t>(G, 1), acc_identity(s));
      DCol c = agg_finalize_values(ctx, s.func, dt, s.acc, s.track_nn ? &s.nn : nullptr, G);
      c.dtype = s.return_dtype;
      o.cols.push_back(c);
    }
    if (!a->in_order && G > 1 && !a->any_order) {
      // groups were discovered out of row order (partition route): order by first row
      ProfScope ps(ctx, "agg_order_groups");
      agg_refresh_gfirst(ctx, a->st);
      BufP keys = ctx->alloc(8 * (size_t)G), perm = ctx->alloc(4 * (size_t)G);
      SQ_HIP(hipMemcpyAsync(keys->p, a->st.gfirst.buf->p, 8 * (size_t)G, hipMemcpyDeviceToDevice, ctx->stream));
      iota_u32(ctx, perm->as<uint32_t>(), G);
      int bits = 1;
      while (bits < 64 && (1ull << bits) <= (uint64_t)std::max<int64_t>(a->rows_seen, 1)) bits++;
      radix_sort_pairs(ctx, keys->as<uint64_t>(), perm->as<uint32_t>(), G, 0, bits, true); // (first rows < rows_seen < 2^bits)
      if (!gather_columns_packed(ctx, o.cols, G, perm->as<uint32_t>(), G))
        for (DCol &c : o.cols) c = gather_column(ctx, c, perm->p, false, nullptr, G);
    }
    place_aggregate_columns(a, o);
    return o;
  }
}

int sqlrs_hash_agg_set_group_order(sqlrs_hash_agg_t *a, int group_order) {
  return guard(a->ctx, [&] {
    if (group_order != SQLRS_GROUP_ORDER_FIRST_SEEN && group_order != SQLRS_GROUP_ORDER_ANY)
      fail(SQLRS_ERR_INTERNAL, "unknown group order");
    if (!a->distinct_aggs.empty() && group_order == SQLRS_GROUP_ORDER_ANY)
      fail(SQLRS_ERR_INTERNAL, "DISTINCT aggregates line up by first-seen order: SQLRS_GROUP_ORDER_ANY not supported");
    a->any_order = group_order == SQLRS_GROUP_ORDER_ANY; // (parts stay in first-seen order: their columns line up by it)
  });
}

int sqlrs_hash_agg_set_filter(sqlrs_hash_agg_t *a, const sqlrs_expr_t *filter) {
  return guard(a->ctx, [&] {
    if (a->saw_batch || !a->staged.empty() || a->hstage.has_schema)
      fail(SQLRS_ERR_INTERNAL, "set_filter after the first batch");
    a->has_filter = filter && filter->num_nodes > 0;
    if (a->has_filter) a->filter = expr_from_abi(filter);
    for (sqlrs_hash_agg *p : a->parts) {
      int st = sqlrs_hash_agg_set_filter(p, filter);
      if (st != SQLRS_OK) fail(st, a->ctx->last_error);
    }
  });
}
int64_t sqlrs_hash_agg_filter_fused_batches(const sqlrs_hash_agg_t *a) {
  return a->parts.empty() ? a->filter_fused_batches : a->parts[0]->filter_fused_batches;
}

void sqlrs_hash_agg_destroy(sqlrs_hash_agg_t *a) { delete a; }

} // extern "C"

// ============================================================ HashJoin + HashAgg fused ==
#include "join_state.hpp"

extern "C" {
int sqlrs_hash_join_create(sqlrs_ctx_t *, int, int, const sqlrs_expr_t *, const sqlrs_expr_t *, const sqlrs_expr_t *,
                           int, const int32_t *, sqlrs_hash_join_t **);
int sqlrs_hash_join_build_push(sqlrs_hash_join_t *, const sqlrs_batch_t *);
int sqlrs_hash_join_build_finish(sqlrs_hash_join_t *);
int sqlrs_hash_join_probe_push(sqlrs_hash_join_t *, const sqlrs_batch_t *, int, sqlrs_batch_t **);
int sqlrs_hash_join_probe_indices(sqlrs_hash_join_t *, const sqlrs_batch_t *, int, sqlrs_batch_t **);
void sqlrs_hash_join_destroy(sqlrs_hash_join_t *);
void sqlrs_batch_release(sqlrs_batch_t *);
int sqlrs_filter_create(sqlrs_ctx_t *, const sqlrs_expr_t *, sqlrs_filter_t **);
int sqlrs_filter_push(sqlrs_filter_t *, const sqlrs_batch_t *, int, sqlrs_batch_t **);
void sqlrs_filter_destroy(sqlrs_filter_t *);
}

namespace sq {
// `col OP constant` over an int64 / float64 column without NULLs -> RowFilter (the shape the first
// partition level can evaluate itself); anything else is run by the Filter operator
bool fusable_row_filter(const Expr &e, InBatch &ib, RowFilter *rf) {
  if (e.nodes.size() != 3) return false;
  const sqlrs_expr_node_t &a = e.nodes[0], &b = e.nodes[1], &o = e.nodes[2];
  if (a.op != SQLRS_EXPR_INPUT_REF || b.op != SQLRS_EXPR_CONSTANT || b.is_null) return false;
  if (o.op < SQLRS_EXPR_GT || o.op > SQLRS_EXPR_NOTEQ) return false;
  if (a.index < 0 || a.index >= ib.num_columns()) return false;
  const DCol &c = ib.col(a.index);
  if (c.dtype != b.dtype || c.stride == 0) return false;
  if (c.dtype != SQLRS_INT64 && c.dtype != SQLRS_FLOAT64) return false;
  if (c.validity && c.null_count != 0) return false;
  rf->col = c.v<uint64_t>();
  rf->is_f64 = c.dtype == SQLRS_FLOAT64;
  if (rf->is_f64) {
    uint64_t bits;
    std::memcpy(&bits, &b.f, 8);
    rf->kord = (bits >> 63) ? ~bits : (bits | (1ull << 63)); // f64_to_ordered
  } else {
    rf->kord = (uint64_t)b.i ^ (1ull << 63);
  }
  static const uint32_t masks[6] = {4, 1, 6, 3, 2, 5}; // GT, LT, GTEQ, LTEQ, EQ, NOTEQ: keep if {<, ==, >}
  rf->keep_mask = masks[o.op - SQLRS_EXPR_GT];
  return true;
}
} // namespace sq

struct sqlrs_join_agg {
  Ctx *ctx = nullptr;
  sqlrs_hash_join *join = nullptr;
  sqlrs_hash_agg *agg = nullptr;
  int nleft = 0;
  PartitionedRows build_parts; // build keys in bucket order (cache for the fused route)
  int64_t fused_batches = 0, composed_batches = 0;
  // probe batches wait here (private copies) and are processed as one batch, for the same reason
  // HashAgg stages its input; a first batch of >= 2^26 rows is processed in place
  std::vector<DBatch> staged;
  int64_t staged_rows = 0;
  bool processed_any = false;
  // FilterExecutor directly below the probe side (sqlrs_join_agg_set_probe_filter)
  bool has_filter = false;
  Expr probe_filter;
  int64_t filter_fused_batches = 0;
  HostStage hstage; // small HOST probe batches (raw, unfiltered) until enough rows for one upload
  // Eager aggregation (GROUP BY columns of the BUILD side, e.g. `GROUP BY d.region` over fact JOIN dim): when the build
  // keys are unique every build column is a function of the join key, so the rows are first grouped by the probe-side
  // JOIN KEY (`inner`: the fused route of join_agg_process, groups in first-seen order), each of those groups finds
  // its build row once (one probe of the join per distinct key instead of one per row), and `outer` re-aggregates
  // the partial rows by the requested columns: COUNT -> SUM of the counts, SUM / MIN / MAX of the partials.  `outer`
  // sees the partial groups in first-seen order, so its own first-seen order is the operator's (hash_agg.rs:98).
  bool eager_possible = false, eager_decided = false, eager = false;
  bool remapped_key = false; // a GROUP BY on the probe-side join key was redirected to the build-side key column
  int lkey_col = -1, rkey_col = -1;
  sqlrs_hash_agg *inner = nullptr, *outer = nullptr;
  std::vector<int> group_cols; // build-side column of each GROUP BY expression
  int64_t eager_groups = 0;    // partial groups (distinct join keys) the last finish() re-aggregated
  ~sqlrs_join_agg() {
    if (join) sqlrs_hash_join_destroy(join);
    delete agg;
    delete inner;
    delete outer;
  }
};

// GROUP BY over build-side columns only (and not just the join key itself, which the fused route takes directly), one
// INPUT_REF key pair, plain aggregates whose arguments read the probe side: sets up `inner` and `outer`
static void join_agg_plan_eager(sqlrs_join_agg *ja, int num_keys, const sqlrs_expr_t *left_keys, const sqlrs_expr_t *right_keys,
                                int num_group_by, const sqlrs_expr_t *group_by, int num_aggs, const sqlrs_agg_func_t *aggs) {
  const char *env = hook("SQLRS_EAGER_AGG"); // test / tuning hook: 0 = never
  if (env && std::atoi(env) == 0) return;
  if (num_keys != 1 || num_group_by < 1 || left_keys[0].num_nodes != 1 || right_keys[0].num_nodes != 1 ||
      left_keys[0].nodes[0].op != SQLRS_EXPR_INPUT_REF || right_keys[0].nodes[0].op != SQLRS_EXPR_INPUT_REF)
    return;
  const int lc = left_keys[0].nodes[0].index, rc = right_keys[0].nodes[0].index;
  ja->lkey_col = lc;
  ja->rkey_col = rc;
  for (int g = 0; g < num_group_by; g++) {
    if (group_by[g].num_nodes != 1 || group_by[g].nodes[0].op != SQLRS_EXPR_INPUT_REF) return;
    int idx = group_by[g].nodes[0].index;
    if (idx == ja->nleft + rc) { // the probe-side join key: equal to the build side's in every joined row (exactly compared) —
      idx = lc;                  // bit for bit only for integer keys of ONE type (checked against the build column at the
      ja->remapped_key = true;   // first batch; -0.0 / 0.0 and int32 / int64 pairs keep the composed route)
    }
    if (idx < 0 || idx >= ja->nleft) return;
    ja->group_cols.push_back(idx);
  }
  if (num_group_by == 1 && ja->group_cols[0] == lc) return; // the fused route's own shape
  for (int i = 0; i < num_aggs; i++) {
    if (aggs[i].distinct || aggs[i].return_dtype == SQLRS_UTF8) return;
    if (aggs[i].func < SQLRS_AGG_COUNT || aggs[i].func > SQLRS_AGG_MAX) return;
    for (int k = 0; k < aggs[i].arg.num_nodes; k++)
      if (aggs[i].arg.nodes[k].op == SQLRS_EXPR_INPUT_REF && aggs[i].arg.nodes[k].index < ja->nleft) return;
  }
  // inner: GROUP BY the probe-side join key (joined-schema index), the caller's aggregates as they are
  sqlrs_expr_node_t kn = right_keys[0].nodes[0];
  kn.index = ja->nleft + rc;
  sqlrs_expr_t ke{&kn, 1, 0};
  int st = sqlrs_hash_agg_create((sqlrs_ctx_t *)ja->ctx, 1, &ke, num_aggs, aggs, &ja->inner);
  if (st != SQLRS_OK) fail(st, ja->ctx->last_error);
  // outer: input = [group columns..., partial aggregates...]
  std::vector<sqlrs_expr_node_t> nodes((size_t)(num_group_by + num_aggs));
  std::vector<sqlrs_expr_t> gb((size_t)num_group_by);
  std::vector<sqlrs_agg_func_t> fin((size_t)std::max(num_aggs, 1));
  for (int c = 0; c < num_group_by + num_aggs; c++) {
    nodes[(size_t)c] = sqlrs_expr_node_t{};
    nodes[(size_t)c].op = SQLRS_EXPR_INPUT_REF;
    nodes[(size_t)c].index = c;
  }
  for (int g = 0; g < num_group_by; g++) gb[(size_t)g] = sqlrs_expr_t{&nodes[(size_t)g], 1, 0};
  for (int i = 0; i < num_aggs; i++) {
    sqlrs_agg_func_t f{};
    f.func = aggs[i].func == SQLRS_AGG_COUNT ? SQLRS_AGG_SUM : aggs[i].func; // counts add up (count.rs:17-23 per batch)
    f.return_dtype = aggs[i].func == SQLRS_AGG_COUNT ? SQLRS_INT64 : aggs[i].return_dtype;
    f.arg = sqlrs_expr_t{&nodes[(size_t)(num_group_by + i)], 1, 0};
    fin[(size_t)i] = f;
  }
  st = sqlrs_hash_agg_create((sqlrs_ctx_t *)ja->ctx, num_group_by, gb.data(), num_aggs, fin.data(), &ja->outer);
  if (st != SQLRS_OK) fail(st, ja->ctx->last_error);
  ja->eager_possible = true;
}

extern "C" {

int sqlrs_join_agg_create(sqlrs_ctx_t *ctx, int num_keys, const sqlrs_expr_t *left_keys,
                          const sqlrs_expr_t *right_keys, int num_left_columns, int num_right_columns,
                          const int32_t *right_dtypes, int num_group_by, const sqlrs_expr_t *group_by,
                          int num_aggs, const sqlrs_agg_func_t *aggs, sqlrs_join_agg_t **out) {
  return guard(ctx, [&] {
    auto ja = std::unique_ptr<sqlrs_join_agg>(new sqlrs_join_agg());
    ja->ctx = ctx;
    ja->nleft = num_left_columns;
    int st = sqlrs_hash_join_create(ctx, SQLRS_JOIN_INNER, num_keys, left_keys, right_keys, nullptr,
                                    num_right_columns, right_dtypes, &ja->join);
    if (st != SQLRS_OK) fail(st, ctx->last_error);
    ja->join->lazy_table = true; // (join_state.hpp: built when the composed route first probes it)
    st = sqlrs_hash_agg_create(ctx, num_group_by, group_by, num_aggs, aggs, &ja->agg);
    if (st != SQLRS_OK) fail(st, ctx->last_error);
    join_agg_plan_eager(ja.get(), num_keys, left_keys, right_keys, num_group_by, group_by, num_aggs, aggs);
    *out = ja.release();
  });
}

int sqlrs_join_agg_build_push(sqlrs_join_agg_t *ja, const sqlrs_batch_t *left) {
  return sqlrs_hash_join_build_push(ja->join, left);
}
int sqlrs_join_agg_build_finish(sqlrs_join_agg_t *ja) { return sqlrs_hash_join_build_finish(ja->join); }

// one probe batch: (Filter, filter.rs:13-25, when `with_filter`) -> HashJoin probe (hash_join.rs:207-292)
// feeding HashAgg push (hash_agg.rs:44-122)
static int join_agg_process(sqlrs_join_agg_t *ja, const sqlrs_batch_t *right, bool with_filter) {
  if (with_filter && !ja->has_filter) with_filter = false;
  bool filter_pending = with_filter; // the filter still has to be applied by the composed path below
  int st0 = guard(ja->ctx, [&] {
    ja->processed_any = true;
    Ctx *ctx = ja->ctx;
    SQ_HIP(hipSetDevice(ctx->device));
    sqlrs_hash_join *j = ja->join;
    if (!j->finished) fail(SQLRS_ERR_INTERNAL, "probe before build_finish");
    if (j->empty_build) return; // the join emits nothing (hash_join.rs:183-185)
    if (!ja->eager_decided) { // once, at the first batch: eager aggregation needs unique, exactly compared build keys
      ja->eager_decided = true;
      bool same_int_key = true; // (advisor, round 3: the output column would take the build column's type and bits)
      if (ja->remapped_key) {
        const int lc = ja->lkey_col, rc = ja->rkey_col;
        const int32_t ld = (lc >= 0 && lc < (int)j->left.cols.size()) ? j->left.cols[(size_t)lc].dtype : SQLRS_NULLTYPE;
        const int32_t rd = (rc >= 0 && rc < right->num_columns) ? right->columns[rc].dtype : SQLRS_NULLTYPE;
        same_int_key = ld == rd && (ld == SQLRS_INT64 || ld == SQLRS_INT32 || ld == SQLRS_UINT64 || ld == SQLRS_UINT32);
      }
      if (ja->eager_possible && same_int_key && j->exact && right->num_rows >= (1ll << 16)) {
        hash_join_ensure_table(j); // (establishes `unique` when the direct-address table did not)
        ja->eager = j->unique && j->unique_known;
      }
    }
    sqlrs_hash_agg *a = ja->eager ? ja->inner : ja->agg;
    // Fused route: Inner join on ONE exactly-compared key with unique build keys, grouped by
    // that key, aggregate arguments taken from the probe side only.  Then every probe row
    // yields at most one joined row and Agg(Join(build, probe)) = Agg(probe rows whose key
    // has a build partner): the joined batch is never materialised.
    // (a join whose hash table is still deferred has not established uniqueness: the fused bucket pass inserts the
    //  build keys itself and reports duplicates — the attempt then fails and the table is built after all)
    // (duplicate build keys over a dense range: the direct-addressed route takes a multiplicity per key)
    const bool dup_dense = j->unique_known && !j->unique && j->dup_range && !j->bkeys_validity;
    bool eligible = (j->unique || !j->unique_known || dup_dense) && j->exact && j->lkeys.size() == 1 && j->lkeys[0].nodes.size() == 1 &&
                    j->rkeys[0].nodes.size() == 1 && j->lkeys[0].nodes[0].op == SQLRS_EXPR_INPUT_REF &&
                    j->rkeys[0].nodes[0].op == SQLRS_EXPR_INPUT_REF && a->group_by.size() == 1 &&
                    a->group_by[0].nodes.size() == 1 && a->group_by[0].nodes[0].op == SQLRS_EXPR_INPUT_REF &&
                    a->distinct_aggs.empty() && a->parts.empty() && !a->strong_keys && right->num_rows >= (1ll << 16);
    if (eligible) {
      int lc = j->lkeys[0].nodes[0].index, rc = j->rkeys[0].nodes[0].index, g = a->group_by[0].nodes[0].index;
      eligible = (g == lc || g == ja->nleft + rc);
      for (const Expr &e : a->arg_exprs)
        for (const auto &nd : e.nodes)
          if (nd.op == SQLRS_EXPR_INPUT_REF && nd.index < ja->nleft) eligible = false;
    }
    if (eligible) {
      InBatch ib(ctx, right);
      auto colfn = [&](int i) -> const DCol & { return ib.col(i); };
      int64_t n = ib.rows();
      RowFilter rf;
      // The fused filter drops rows INSIDE the first partition pass, i.e. after the aggregate arguments were
      // evaluated over all n rows: an argument that can raise (integer DIVIDE: "Divide by zero error") would
      // fail on a row the predicate removes — `SUM(a / b) ... WHERE b > 0` succeeds in the reference
      // (filter.rs:13-25 runs first).  Such plans take the Filter operator first.
      bool args_cannot_raise = true;
      for (const Expr &e : a->arg_exprs)
        for (const auto &nd : e.nodes)
          if (nd.op == SQLRS_EXPR_DIVIDE) args_cannot_raise = false;
      const bool fuse_filter = with_filter && args_cannot_raise && fusable_row_filter(ja->probe_filter, ib, &rf);
      if (with_filter && !fuse_filter) eligible = false; // Filter operator first, then this function again
      std::vector<DCol> kcols;
      NKeys nk;
      if (eligible) {
        kcols.push_back(eval_expr(ctx, j->rkeys[0], colfn, n, true));
        nk = normalize_keys(ctx, kcols, n);
      }
      if (eligible && nk.exact && nk.dtype == j->key_dtype) {
        if (!a->saw_batch) {
          a->saw_batch = true;
          a->key_dtypes.push_back(kcols[0].dtype);
        }
        std::vector<DCol> acols = eval_arg_columns(a, colfn, n, ja->nleft);
        JoinSide js;
        js.keys = j->bkeys->as<uint64_t>();
        js.validity = j->bkeys_validity ? j->bkeys_validity->as<uint64_t>() : nullptr;
        js.n = j->nB;
        js.cache = &ja->build_parts;
        js.unique_known = j->unique_known;
        if (dup_dense) {
          js.mult = hash_join_dup_mult(j);
          js.unique_known = true; // (what the route needs of uniqueness — one slot per key — the multiplicities restore)
          js.range_known = js.mult != nullptr;
          js.omin = j->dup_min ^ (1ull << 63);
          js.omax = js.omin + (j->dup_range - 1);
          js.cache = nullptr; // (no build-side partition on this route)
        } else if (j->dense && j->dense_range) { // the direct-address table's key range
          js.range_known = true;
          js.omin = j->dense_min ^ (1ull << 63);
          js.omax = js.omin + (j->dense_range - 1);
          if (j->dense_range != (uint64_t)j->nB && !js.validity) js.bits = hash_join_dense_bits(j); // keys with gaps
        }
        flush_staged(a); // batches staged by the composed route come first in row order
        if (agg_consume(a, n, kcols, nk, acols, &js, fuse_filter ? &rf : nullptr)) {
          a->rows_seen += n;
          ja->fused_batches++;
          if (fuse_filter) ja->filter_fused_batches++;
          filter_pending = false;
          return;
        }
      }
      hash_join_ensure_table(j); // (the attempt may have failed on duplicate build keys: `unique` is a fact from here on)
    }
    if (filter_pending) return; // (handled below: Filter operator, then the unfiltered path)
    // composed route: materialise the joined batch on the device and aggregate it
    sqlrs_batch_t *joined = nullptr;
    int st = sqlrs_hash_join_probe_push(j, right, SQLRS_MEM_DEVICE, &joined);
    if (st != SQLRS_OK) fail(st, ctx->last_error);
    if (joined) {
      st = sqlrs_hash_agg_push(a, joined);
      std::string err = ctx->last_error;
      sqlrs_batch_release(joined);
      if (st != SQLRS_OK) fail(st, err);
    }
    ja->composed_batches++;
  });
  if (st0 != SQLRS_OK || !filter_pending) return st0;
  // the filter was not fused: run the Filter operator on the batch, then process its output
  Ctx *ctx = ja->ctx;
  std::vector<sqlrs_expr_node_t> nodes = ja->probe_filter.nodes;
  for (size_t i = 0; i < nodes.size(); i++) nodes[i].s = ja->probe_filter.strings[i].empty() ? nullptr : ja->probe_filter.strings[i].c_str();
  sqlrs_expr_t fe{nodes.data(), (int32_t)nodes.size(), 0};
  sqlrs_filter_t *f = nullptr;
  int st = sqlrs_filter_create((sqlrs_ctx_t *)ctx, &fe, &f);
  if (st != SQLRS_OK) return st;
  sqlrs_batch_t *kept = nullptr;
  st = sqlrs_filter_push(f, right, SQLRS_MEM_DEVICE, &kept);
  sqlrs_filter_destroy(f);
  if (st != SQLRS_OK) return st;
  st = join_agg_process(ja, kept, false);
  sqlrs_batch_release(kept);
  return st;
}

static int join_agg_flush(sqlrs_join_agg_t *ja) {
  if (ja->staged.empty()) return SQLRS_OK;
  Ctx *ctx = ja->ctx;
  sqlrs_batch_t *view = nullptr;
  int st = guard(ctx, [&] {
    SQ_HIP(hipSetDevice(ctx->device));
    std::vector<DBatch> bs = std::move(ja->staged);
    ja->staged.clear();
    ja->staged_rows = 0;
    DBatch all;
    size_t nc = bs[0].cols.size();
    for (size_t c = 0; c < nc; c++) {
      std::vector<const DCol *> parts;
      for (DBatch &b : bs) {
        if (b.cols.size() != nc) fail(SQLRS_ERR_ARROW, "concat_batches: schema mismatch");
        parts.push_back(&b.cols[c]);
      }
      all.cols.push_back(bs.size() == 1 ? bs[0].cols[c] : concat_columns(ctx, parts));
    }
    for (DBatch &b : bs) all.rows += b.rows;
    view = emit_batch(ctx, std::move(all), SQLRS_MEM_DEVICE);
  });
  if (st != SQLRS_OK) return st;
  st = join_agg_process(ja, view, false); // staged batches are already filtered
  sqlrs_batch_release(view);
  return st;
}

static int join_agg_probe_push_device(sqlrs_join_agg_t *ja, const sqlrs_batch_t *right);
static int join_agg_flush_host(sqlrs_join_agg_t *ja) {
  if (!ja->hstage.has_schema) return SQLRS_OK;
  sqlrs_batch_t *dev = nullptr;
  int st = guard(ja->ctx, [&] {
    SQ_HIP(hipSetDevice(ja->ctx->device));
    dev = ja->hstage.take();
  });
  if (st != SQLRS_OK) return st;
  st = join_agg_probe_push_device(ja, dev);
  sqlrs_batch_release(dev);
  return st;
}

int sqlrs_join_agg_probe_push(sqlrs_join_agg_t *ja, const sqlrs_batch_t *right) {
  ja->hstage.ctx = ja->ctx;
  if (ja->hstage.accepts(right)) {
    int st = guard(ja->ctx, [&] {
      if (!ja->join->finished) fail(SQLRS_ERR_INTERNAL, "probe before build_finish");
      ja->hstage.append(right);
    });
    // (flushed in units large enough for the in-place route: the Filter is then evaluated by the first partition pass)
    if (st != SQLRS_OK || ja->hstage.rows < std::max(HOST_STAGE_FLUSH_ROWS, STAGE_DIRECT_ROWS)) return st;
    return join_agg_flush_host(ja);
  }
  int st = join_agg_flush_host(ja);
  return st != SQLRS_OK ? st : join_agg_probe_push_device(ja, right);
}

static int join_agg_probe_push_device(sqlrs_join_agg_t *ja, const sqlrs_batch_t *right) {
  if (ja->staged.empty() && right->num_rows >= STAGE_DIRECT_ROWS) return join_agg_process(ja, right, true);
  sqlrs_batch_t *kept = nullptr;
  if (ja->has_filter) { // small batches are filtered on arrival; what is staged is the Filter's output
    Ctx *ctx = ja->ctx;
    std::vector<sqlrs_expr_node_t> nodes = ja->probe_filter.nodes;
    for (size_t i = 0; i < nodes.size(); i++) nodes[i].s = ja->probe_filter.strings[i].empty() ? nullptr : ja->probe_filter.strings[i].c_str();
    sqlrs_expr_t fe{nodes.data(), (int32_t)nodes.size(), 0};
    sqlrs_filter_t *f = nullptr;
    int stf = sqlrs_filter_create((sqlrs_ctx_t *)ctx, &fe, &f);
    if (stf != SQLRS_OK) return stf;
    stf = sqlrs_filter_push(f, right, SQLRS_MEM_DEVICE, &kept);
    sqlrs_filter_destroy(f);
    if (stf != SQLRS_OK) return stf;
    right = kept;
  }
  int st = guard(ja->ctx, [&] {
    SQ_HIP(hipSetDevice(ja->ctx->device));
    if (!ja->join->finished) fail(SQLRS_ERR_INTERNAL, "probe before build_finish");
    InBatch ib(ja->ctx, right);
    ja->staged.push_back(ib.materialize(true)); // library-owned device batches are shared, not copied
    ja->staged_rows += right->num_rows;
  });
  if (kept) sqlrs_batch_release(kept);
  if (st != SQLRS_OK) return st;
  return ja->staged_rows >= STAGE_FLUSH_ROWS ? join_agg_flush(ja) : SQLRS_OK;
}

int sqlrs_join_agg_finish(sqlrs_join_agg_t *ja, int out_mem, sqlrs_batch_t **out) {
  int st = join_agg_flush_host(ja);
  if (st != SQLRS_OK) return st;
  st = join_agg_flush(ja);
  if (st != SQLRS_OK) return st;
  if (!ja->eager || !ja->inner->saw_batch) return sqlrs_hash_agg_finish(ja->agg, out_mem, out);
  // eager aggregation: partial groups by join key -> their build rows -> re-aggregation by the GROUP BY columns
  sqlrs_batch_t *part = nullptr, *fake = nullptr, *pairs = nullptr, *fin = nullptr;
  st = sqlrs_hash_agg_finish(ja->inner, SQLRS_MEM_DEVICE, &part);
  if (st == SQLRS_OK)
    st = guard(ja->ctx, [&] {
      Ctx *ctx = ja->ctx;
      SQ_HIP(hipSetDevice(ctx->device));
      sqlrs_hash_join *j = ja->join;
      InBatch pb(ctx, part);
      const int64_t G = pb.rows();
      ja->eager_groups = G;
      const int rc = j->rkeys[0].nodes[0].index;
      DBatch f;
      f.rows = G;
      if (G > 0) {
        // the partial keys as a probe batch of the join (every column is the key column: only column rc is read)
        DBatch fk;
        fk.rows = G;
        for (int c = 0; c <= rc; c++) fk.cols.push_back(pb.col(0));
        fake = emit_batch(ctx, std::move(fk), SQLRS_MEM_DEVICE);
        int s2 = sqlrs_hash_join_probe_indices(j, fake, SQLRS_MEM_DEVICE, &pairs);
        if (s2 != SQLRS_OK) fail(s2, ctx->last_error);
        // unique build keys, and a partial group exists only for a key with a partner: exactly one pair per partial
        // group, in probe-row order (hash_join.rs:207-253) — pair i belongs to partial group i
        if (!pairs || pairs->num_rows != G) fail(SQLRS_ERR_INTERNAL, "eager aggregation: a partial group without exactly one build row");
        InBatch pr(ctx, pairs);
        for (int gc : ja->group_cols) f.cols.push_back(gather_column(ctx, j->left.cols[(size_t)gc], pr.col(0).values, true, nullptr, G));
      } else {
        for (int gc : ja->group_cols) f.cols.push_back(make_null_column(ctx, j->left.cols[(size_t)gc].dtype, 0));
      }
      for (int c = 1; c < pb.num_columns(); c++) f.cols.push_back(pb.col(c));
      fin = emit_batch(ctx, std::move(f), SQLRS_MEM_DEVICE);
    });
  if (st == SQLRS_OK) st = sqlrs_hash_agg_push(ja->outer, fin);
  if (st == SQLRS_OK) st = sqlrs_hash_agg_finish(ja->outer, out_mem, out);
  for (sqlrs_batch_t *b : {part, fake, pairs, fin})
    if (b) sqlrs_batch_release(b);
  return st;
}
int sqlrs_join_agg_set_group_order(sqlrs_join_agg_t *ja, int group_order) {
  int st = sqlrs_hash_agg_set_group_order(ja->agg, group_order);
  if (st == SQLRS_OK && ja->outer) st = sqlrs_hash_agg_set_group_order(ja->outer, group_order); // (`inner` stays in first-seen order)
  return st;
}
// partial groups (distinct join keys) that the last finish() re-aggregated by build-side GROUP BY columns; 0 = the
// eager-aggregation route did not run (diagnostics / tests)
int64_t sqlrs_join_agg_eager_groups(const sqlrs_join_agg_t *ja) { return ja->eager_groups; }
// number of probe batches that took the fused route (diagnostics / tests)
int64_t sqlrs_join_agg_fused_batches(const sqlrs_join_agg_t *ja) { return ja->fused_batches; }
int sqlrs_join_agg_set_probe_filter(sqlrs_join_agg_t *ja, const sqlrs_expr_t *filter) {
  return guard(ja->ctx, [&] {
    if (ja->processed_any || !ja->staged.empty() || ja->hstage.has_schema)
      fail(SQLRS_ERR_INTERNAL, "set_probe_filter after the first probe batch");
    ja->has_filter = filter && filter->num_nodes > 0;
    if (ja->has_filter) ja->probe_filter = expr_from_abi(filter);
  });
}
int64_t sqlrs_join_agg_filter_fused_batches(const sqlrs_join_agg_t *ja) { return ja->filter_fused_batches; }
void sqlrs_join_agg_destroy(sqlrs_join_agg_t *ja) { delete ja; }

} // extern "C"
