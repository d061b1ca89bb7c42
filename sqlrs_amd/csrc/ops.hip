// ops.hip — C entry points of FilterExecutor, eval_column, HashAggExecutor and OrderExecutor
// (the join lives in join.hip).  Each entry point cites the reference lines it replaces.
#include "agg_state.hpp"
#include "common.hpp"
#include "device_utils.hpp"
#include "prims.hpp"

using namespace sq;

// =========================================================================== Filter ==
struct sqlrs_filter {
  Ctx *ctx;
  Expr expr;
};

extern "C" {

int sqlrs_filter_create(sqlrs_ctx_t *ctx, const sqlrs_expr_t *expr, sqlrs_filter_t **out) {
  return guard(ctx, [&] {
    auto *f = new sqlrs_filter();
    f->ctx = ctx;
    try {
      f->expr = expr_from_abi(expr);
    } catch (...) {
      delete f;
      throw;
    }
    *out = f;
  });
}

// one iteration of the for_await loop  [ref: filter.rs:16-24]
int sqlrs_filter_push(sqlrs_filter_t *f, const sqlrs_batch_t *in, int out_mem, sqlrs_batch_t **out) {
  return guard(f->ctx, [&] {
    Ctx *ctx = f->ctx;
    SQ_HIP(hipSetDevice(ctx->device));
    InBatch ib(ctx, in);
    auto colfn = [&](int i) -> const DCol & { return ib.col(i); };
    int64_t rows = ib.rows();
    Selection sel;
    DCol fast_col;
    int fast_idx = -1;
    if (!filter_fast_path(ctx, f->expr, colfn, rows, &fast_idx, &sel, &fast_col)) {
      DCol mask = eval_expr(ctx, f->expr, colfn, rows, false);
      mask.length = rows;
      sel = selection_from_mask(ctx, mask);
    }
    DBatch o;
    o.rows = sel.count;
    for (int i = 0; i < ib.num_columns(); i++) {
      if (i == fast_idx)
        o.cols.push_back(fast_col);
      else
        o.cols.push_back(compact_column(ctx, ib.col(i), sel));
    }
    *out = emit_batch(ctx, std::move(o), out_mem);
  });
}

void sqlrs_filter_destroy(sqlrs_filter_t *f) { delete f; }

// [ref: evaluator.rs:13-28]
int sqlrs_eval_expr(sqlrs_ctx_t *ctx, const sqlrs_expr_t *expr, const sqlrs_batch_t *in, int out_mem,
                    sqlrs_batch_t **out) {
  return guard(ctx, [&] {
    SQ_HIP(hipSetDevice(ctx->device));
    Expr e = expr_from_abi(expr);
    InBatch ib(ctx, in);
    auto colfn = [&](int i) -> const DCol & { return ib.col(i); };
    DBatch o;
    o.rows = ib.rows();
    o.cols.push_back(eval_expr(ctx, e, colfn, ib.rows(), true));
    o.cols[0].length = ib.rows();
    *out = emit_batch(ctx, std::move(o), out_mem);
  });
}

} // extern "C"

// ========================================================================== HashAgg ==
namespace {

struct AggSpec {
  int func = 0, distinct = 0;
  int32_t return_dtype = 0;
  Expr arg;
  GrowBuf acc, nn;
  bool track_nn = false; // nn maintained (COUNT always; others once a NULL input was seen)
  int32_t acc_dtype = 0; // dtype the accumulator works in
};

uint64_t acc_identity(const AggSpec &a) {
  if (a.func == SQLRS_AGG_MIN) return ~0ull;
  return 0; // SUM (0 / +0.0), MAX (smallest ordered image), COUNT
}

} // namespace

struct sqlrs_hash_agg {
  Ctx *ctx = nullptr;
  std::vector<Expr> group_by;
  std::vector<AggSpec> aggs;
  AggState st;
  bool saw_batch = false;
  int64_t rows_seen = 0;
  std::vector<int32_t> key_dtypes;
  std::vector<std::vector<DCol>> key_parts; // per key column: values of new groups, per batch
};

extern "C" {

int sqlrs_hash_agg_create(sqlrs_ctx_t *ctx, int num_group_by, const sqlrs_expr_t *group_by,
                          int num_aggs, const sqlrs_agg_func_t *aggs, sqlrs_hash_agg_t **out) {
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
      if (s.distinct && (s.func == SQLRS_AGG_COUNT || s.func == SQLRS_AGG_SUM))
        fail(SQLRS_ERR_INTERNAL, "DISTINCT aggregates are not yet supported on the device path");
      a->aggs.push_back(std::move(s));
    }
    a->key_parts.resize((size_t)num_group_by);
    *out = a.release();
  });
}

// one iteration of the for_await loop  [ref: hash_agg.rs:44-122]
int sqlrs_hash_agg_push(sqlrs_hash_agg_t *a, const sqlrs_batch_t *in) {
  return guard(a->ctx, [&] {
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
    NKeys nk = normalize_keys(ctx, kcols, n);
    // 3.2 groups (:85-110)
    BufP new_rows;
    int64_t nnew = 0;
    BufP row_gid = agg_resolve_rows(ctx, a->st, nk, nullptr, (uint64_t)a->rows_seen, &new_rows, &nnew);
    if (nnew) // group key values of first sight (:90-96)
      for (size_t c = 0; c < kcols.size(); c++)
        a->key_parts[c].push_back(gather_column(ctx, kcols[c], new_rows->p, false, nullptr, nnew));
    // 2.1 / 4. accumulate (:63-66, :113-121)
    int64_t G = a->st.ngroups;
    for (AggSpec &s : a->aggs) {
      DCol col = eval_expr(ctx, s.arg, colfn, n, true);
      const uint64_t *valid = (col.validity && col.null_count != 0) ? col.validity : nullptr;
      const uint32_t *rg = row_gid->as<uint32_t>();
      if (s.func == SQLRS_AGG_COUNT) {
        s.nn.ensure(ctx, G, 0);
        agg_update_count(ctx, s.nn, rg, valid, nullptr, n);
        continue;
      }
      // has-value tracking starts with the first NULL-bearing batch: until then every
      // existing group has at least one valid value, so its counter is back-filled with 1
      if (valid && !s.track_nn) {
        s.track_nn = true;
        s.nn.ensure(ctx, std::max<int64_t>(G, 1), 0);
        if (G - nnew > 0) fill_u64(ctx, s.nn.buf->as<uint64_t>(), G - nnew, 1);
      }
      if (s.track_nn) {
        s.nn.ensure(ctx, G, 0);
        agg_update_count(ctx, s.nn, rg, valid, nullptr, n);
      }
      if (s.func == SQLRS_AGG_SUM) {
        // SumAccumulator: cast to the return type, then sum (sum.rs:54-60); the reference's
        // sum_result has no (Int32, Int32) arm (sum.rs:64-85)
        if (s.return_dtype != SQLRS_INT64 && s.return_dtype != SQLRS_FLOAT64)
          fail(SQLRS_ERR_INTERNAL, "not expected types for sum");
        if (col.dtype != s.return_dtype) {
          sqlrs_expr_node_t cn[2];
          std::memset(cn, 0, sizeof(cn));
          cn[0].op = SQLRS_EXPR_INPUT_REF;
          cn[0].index = 0;
          cn[1].op = SQLRS_EXPR_TYPE_CAST;
          cn[1].dtype = s.return_dtype;
          Expr ce;
          ce.nodes.assign(cn, cn + 2);
          ce.strings.resize(2);
          auto one = [&](int) -> const DCol & { return col; };
          DCol casted = eval_expr(ctx, ce, one, n, true);
          col = casted;
          valid = (col.validity && col.null_count != 0) ? col.validity : nullptr;
        }
        s.acc_dtype = s.return_dtype;
        s.acc.ensure(ctx, G, 0);
        agg_update_sum(ctx, s.acc, s.acc_dtype, rg, col.values, valid, n);
      } else {
        if (col.dtype != s.return_dtype) fail(SQLRS_ERR_INTERNAL, "unsupported min_max scalar type");
        s.acc_dtype = col.dtype;
        s.acc.ensure(ctx, G, acc_identity(s));
        agg_update_minmax(ctx, s.acc, s.acc_dtype, s.func == SQLRS_AGG_MIN, rg, col.values, valid, n);
      }
    }
    a->rows_seen += n;
  });
}

// [ref: hash_agg.rs:124-149]
int sqlrs_hash_agg_finish(sqlrs_hash_agg_t *a, int out_mem, sqlrs_batch_t **out) {
  return guard(a->ctx, [&] {
    Ctx *ctx = a->ctx;
    SQ_HIP(hipSetDevice(ctx->device));
    if (!a->saw_batch) // group_and_agg_fields.unwrap() panics on None (:125)
      fail(SQLRS_ERR_INTERNAL, "hash agg finished without any input batch");
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
      if (s.func == SQLRS_AGG_SUM && s.return_dtype != SQLRS_INT64 && s.return_dtype != SQLRS_FLOAT64)
        fail(SQLRS_ERR_INTERNAL, "not expected types for sum");
      int32_t dt = s.acc_dtype ? s.acc_dtype : s.return_dtype;
      s.acc.ensure(ctx, std::max<int64_t>(G, 1), acc_identity(s));
      DCol c = agg_finalize_values(ctx, s.func, dt, s.acc, s.track_nn ? &s.nn : nullptr, G);
      c.dtype = s.return_dtype;
      o.cols.push_back(c);
    }
    *out = emit_batch(ctx, std::move(o), out_mem);
  });
}

void sqlrs_hash_agg_destroy(sqlrs_hash_agg_t *a) { delete a; }

} // extern "C"

// ============================================================================ Order ==
namespace sq {

// kind: 0 i64, 1 f64, 2 i32, 3 bool.  NULL rows get key 0 so that they tie (stable).
template <int KIND>
__global__ void sort_key_kernel(const void *__restrict__ vals, const uint64_t *__restrict__ validity,
                                const uint32_t *__restrict__ perm, int64_t n, int desc,
                                uint64_t *__restrict__ keys) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t r = perm[i];
  if (validity && !((validity[r >> 6] >> (r & 63)) & 1)) {
    keys[i] = 0;
    return;
  }
  uint64_t u;
  if (KIND == 0) u = i64_to_ordered(((const int64_t *)vals)[r]);
  else if (KIND == 1) u = f64_to_ordered(((const double *)vals)[r]);
  else if (KIND == 2) u = i64_to_ordered((int64_t)((const int32_t *)vals)[r]);
  else u = (((const uint64_t *)vals)[r >> 6] >> (r & 63)) & 1;
  keys[i] = desc ? ~u : u;
}
__global__ void sort_valid_key_kernel(const uint64_t *__restrict__ validity,
                                      const uint32_t *__restrict__ perm, int64_t n,
                                      uint64_t *__restrict__ keys) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t r = perm[i];
  keys[i] = (validity[r >> 6] >> (r & 63)) & 1; // NULL (0) first
}

} // namespace sq

struct sqlrs_order {
  Ctx *ctx = nullptr;
  std::vector<Expr> exprs;
  std::vector<int> asc;
  std::vector<DBatch> batches;
};

extern "C" {

int sqlrs_order_create(sqlrs_ctx_t *ctx, int num_keys, const sqlrs_order_by_t *order_by,
                       sqlrs_order_t **out) {
  return guard(ctx, [&] {
    auto o = std::unique_ptr<sqlrs_order>(new sqlrs_order());
    o->ctx = ctx;
    for (int i = 0; i < num_keys; i++) {
      o->exprs.push_back(expr_from_abi(&order_by[i].expr));
      o->asc.push_back(order_by[i].asc);
    }
    *out = o.release();
  });
}

// [ref: order.rs:19-26]
int sqlrs_order_push(sqlrs_order_t *o, const sqlrs_batch_t *in) {
  return guard(o->ctx, [&] {
    SQ_HIP(hipSetDevice(o->ctx->device));
    InBatch ib(o->ctx, in);
    o->batches.push_back(ib.materialize(true));
  });
}

// [ref: order.rs:27-66] concat -> lexsort_to_indices (nulls first, stable) -> take
int sqlrs_order_finish(sqlrs_order_t *o, int out_mem, sqlrs_batch_t **out) {
  return guard(o->ctx, [&] {
    Ctx *ctx = o->ctx;
    SQ_HIP(hipSetDevice(ctx->device));
    if (o->batches.empty()) fail(SQLRS_ERR_INTERNAL, "order finished without any input batch"); // :27
    DBatch all;
    size_t nc = o->batches[0].cols.size();
    for (size_t c = 0; c < nc; c++) {
      std::vector<const DCol *> parts;
      for (DBatch &b : o->batches) {
        if (b.cols.size() != nc) fail(SQLRS_ERR_ARROW, "concat_batches: schema mismatch");
        parts.push_back(&b.cols[c]);
      }
      all.cols.push_back(concat_columns(ctx, parts));
    }
    for (DBatch &b : o->batches) all.rows += b.rows;
    o->batches.clear();
    int64_t n = all.rows;
    if (n > 0xffffffffll) fail(SQLRS_ERR_INTERNAL, "order: more than 2^32 rows");
    auto colfn = [&](int i) -> const DCol & {
      if (i < 0 || (size_t)i >= all.cols.size()) fail(SQLRS_ERR_INTERNAL, "input ref out of range");
      return all.cols[(size_t)i];
    };
    int64_t n1 = std::max<int64_t>(n, 1);
    BufP perm = ctx->alloc(4 * (size_t)n1), keys = ctx->alloc(8 * (size_t)n1);
    iota_u32(ctx, perm->as<uint32_t>(), n);
    dim3 g((unsigned)ceil_div(n1, 256)), b(256);
    // LSD over the sort columns: last key first, each step a stable sort
    for (int k = (int)o->exprs.size() - 1; k >= 0 && n > 1; k--) {
      DCol c = eval_expr(ctx, o->exprs[(size_t)k], colfn, n, true);
      const uint64_t *valid = (c.validity && c.null_count != 0) ? c.validity : nullptr;
      int desc = o->asc[(size_t)k] ? 0 : 1;
      int bits = 64;
      ProfScope ps(ctx, "order_keys");
      switch (c.dtype) {
      case SQLRS_INT64:
        sort_key_kernel<0><<<g, b, 0, ctx->stream>>>(c.values, valid, perm->as<uint32_t>(), n, desc, keys->as<uint64_t>());
        break;
      case SQLRS_FLOAT64:
        sort_key_kernel<1><<<g, b, 0, ctx->stream>>>(c.values, valid, perm->as<uint32_t>(), n, desc, keys->as<uint64_t>());
        break;
      case SQLRS_INT32:
        sort_key_kernel<2><<<g, b, 0, ctx->stream>>>(c.values, valid, perm->as<uint32_t>(), n, desc, keys->as<uint64_t>());
        break;
      case SQLRS_BOOLEAN:
        sort_key_kernel<3><<<g, b, 0, ctx->stream>>>(c.values, valid, perm->as<uint32_t>(), n, desc, keys->as<uint64_t>());
        break;
      case SQLRS_UTF8:
        fail(SQLRS_ERR_INTERNAL, "utf8 sort keys are not supported on the device path");
      default:
        fail(SQLRS_ERR_INTERNAL, "unsupported sort key type");
      }
      SQ_HIP(hipGetLastError());
      radix_sort_pairs(ctx, keys->as<uint64_t>(), perm->as<uint32_t>(), n, 0, bits);
      if (valid) { // nulls_first = true regardless of direction (order.rs:37-40)
        sort_valid_key_kernel<<<g, b, 0, ctx->stream>>>(valid, perm->as<uint32_t>(), n, keys->as<uint64_t>());
        SQ_HIP(hipGetLastError());
        radix_sort_pairs(ctx, keys->as<uint64_t>(), perm->as<uint32_t>(), n, 0, 8);
      }
    }
    DBatch r;
    r.rows = n;
    for (const DCol &c : all.cols) r.cols.push_back(gather_column(ctx, c, perm->p, false, nullptr, n));
    *out = emit_batch(ctx, std::move(r), out_mem);
  });
}

void sqlrs_order_destroy(sqlrs_order_t *o) { delete o; }

} // extern "C"
