// ops.hip — C entry points of FilterExecutor, eval_column, HashAggExecutor and OrderExecutor
// (the join lives in join.hip).  Each entry point cites the reference lines it replaces.
#include <cstdlib>

#include "agg_state.hpp"
#include "common.hpp"
#include "device_utils.hpp"
#include "host_stage.hpp"
#include "prims.hpp"
#include "radix_part.hpp"
#include "small_async.hpp"

using namespace sq;

extern "C" void sqlrs_batch_release(sqlrs_batch_t *batch);

// =========================================================================== Filter ==
struct sqlrs_filter {
  Ctx *ctx;
  Expr expr;
  std::unique_ptr<HostStage> stage; // sqlrs_filter_push_many: the calls' small HOST batches, uploaded together
  void *pin_out = nullptr;          // ... and where their kept rows land (pinned: the copy back runs at the PCIe rate)
  size_t pin_cap = 0;
  ~sqlrs_filter() {
    if (pin_out) (void)hipHostFree(pin_out);
  }
};

// ---- conjunctions of `column OP constant` ------------------------------------------------------------------------
// `a > x AND b < y [AND ...]` (TPC-H Q6 has three range terms) used to go term by term through the expression
// evaluator: a comparison kernel and a BOOLEAN array per term, an AND kernel per pair, the mask conversion, then tile
// offsets and compaction.  For up to four terms over int64 / float64 columns without NULLs one kernel reads the predicate
// columns once and writes the selection's mask words directly (ballot = one word per wave trip); tile offsets and the
// compaction of the columns follow as for any mask.  2e7 rows, three carried columns, two terms: 0.38 -> 0.30 ms.
namespace sq {
constexpr int CONJ_MAX = 4;
struct ConjTerms {
  RowFilter t[CONJ_MAX];
  int n = 0;
};
__global__ __launch_bounds__(256) void filter_conj_mask_kernel(ConjTerms ct, int64_t rows, uint64_t *__restrict__ bits) {
  const int lane = lane_id();
  const int64_t nwords = (rows + 63) / 64;
  constexpr int KU = 4; // words per wave and trip: independent loads in flight
  for (int64_t w0 = ((int64_t)blockIdx.x * 4 + wave_id()) * KU; w0 < nwords; w0 += (int64_t)gridDim.x * 4 * KU) {
    uint64_t v[CONJ_MAX][KU];
#pragma unroll
    for (int q = 0; q < CONJ_MAX; q++) {
      if (q >= ct.n) break;
#pragma unroll
      for (int u = 0; u < KU; u++) v[q][u] = __builtin_nontemporal_load(ct.t[q].col + min((w0 + u) * 64 + lane, rows - 1));
    }
#pragma unroll
    for (int u = 0; u < KU; u++) {
      const int64_t r = (w0 + u) * 64 + lane;
      bool keep = r < rows;
#pragma unroll
      for (int q = 0; q < CONJ_MAX; q++) {
        if (q >= ct.n) break;
        keep = keep && row_passes(ct.t[q], v[q][u]);
      }
      const uint64_t m = __ballot(keep);
      if (lane == 0 && w0 + u < nwords) bits[w0 + u] = m;
    }
  }
}
// true = `sel` is the selection of the conjunction `e` over `ib`
static bool filter_conjunction_fast_path(Ctx *ctx, const Expr &e, InBatch &ib, int64_t rows, Selection *sel) {
  // postfix of t1 AND t2 AND ... : t1 t2 AND t3 AND ...   (every term = INPUT_REF CONSTANT CMP)
  const size_t nn = e.nodes.size();
  if (rows < (1 << 16) || nn < 7 || (nn - 3) % 4 != 0) return false;
  const size_t nterms = 1 + (nn - 3) / 4;
  if (nterms > (size_t)CONJ_MAX) return false;
  ConjTerms ct;
  for (size_t k = 0; k < nterms; k++) {
    const size_t at = k == 0 ? 0 : 3 + (k - 1) * 4;
    if (k > 0 && e.nodes[at + 3].op != SQLRS_EXPR_AND) return false;
    Expr term;
    term.nodes.assign(e.nodes.begin() + (long)at, e.nodes.begin() + (long)at + 3);
    term.strings.assign(e.strings.begin() + (long)at, e.strings.begin() + (long)at + 3);
    if (!fusable_row_filter(term, ib, &ct.t[k])) return false;
  }
  ct.n = (int)nterms;
  const int64_t nwords = ceil_div(rows, 64);
  sel->rows = rows;
  sel->own_bits = ctx->alloc(8 * (size_t)nwords + 8);
  sel->bits = sel->own_bits->as<uint64_t>();
  {
    ProfScope ps(ctx, "filter_conj_mask");
    const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(nwords, 16), 8 * (int64_t)ctx->num_cus));
    filter_conj_mask_kernel<<<dim3(blocks), dim3(256), 0, ctx->stream>>>(ct, rows, sel->own_bits->as<uint64_t>());
    SQ_HIP(hipGetLastError());
  }
  selection_finish(ctx, *sel);
  return true;
}
} // namespace sq

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

} // extern "C"

namespace sq {
// mask = expr.eval_column(batch); filter_record_batch(batch, mask)  [ref: filter.rs:16-24] -> the kept rows; *sel_out
// (optional) = the selection itself (bits + kept rows before every 4096-row tile)
static DBatch filter_batch(sqlrs_filter *f, InBatch &ib, Selection *sel_out) {
  Ctx *ctx = f->ctx;
  auto colfn = [&](int i) -> const DCol & { return ib.col(i); };
  int64_t rows = ib.rows();
  Selection sel;
  DCol fast_col;
  int fast_idx = -1;
  if (!filter_fast_path(ctx, f->expr, colfn, rows, &fast_idx, &sel, &fast_col) &&
      !filter_conjunction_fast_path(ctx, f->expr, ib, rows, &sel)) {
    DCol mask = eval_expr(ctx, f->expr, colfn, rows, false);
    mask.length = rows;
    sel = selection_from_mask(ctx, mask);
  }
  DBatch o;
  o.rows = sel.count;
  // every row passed: the output is the input (filter_record_batch of an all-true mask) — its columns are shared
  // when they are this library's own buffers and copied once when the caller only lent them, not compacted
  DBatch whole;
  const bool keep_all = sel.count == rows && rows > 0;
  if (keep_all) whole = ib.materialize(true);
  for (int i = 0; i < ib.num_columns(); i++) {
    if (i == fast_idx)
      o.cols.push_back(fast_col);
    else if (keep_all)
      o.cols.push_back(whole.cols[(size_t)i]);
    else
      o.cols.push_back(compact_column(ctx, ib.col(i), sel));
  }
  if (sel_out) *sel_out = sel;
  return o;
}

// kept[i] = selected rows before row bounds[i] (bounds ascending, <= rows): tile prefix + the bits of the tile up to it
__global__ void kept_before_kernel(const uint64_t *__restrict__ bits, const uint64_t *__restrict__ tile_off, int64_t rows,
                                   const int64_t *__restrict__ bounds, int64_t n, int64_t total, int64_t *__restrict__ kept) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t b = bounds[i];
  if (b >= rows) {
    kept[i] = total;
    return;
  }
  const int64_t t = b / TILE_ROWS;
  int64_t k = (int64_t)tile_off[t];
  for (int64_t w = t * (TILE_ROWS / 64); w < (b >> 6); w++) k += __popcll(bits[w]);
  if (b & 63) k += __popcll(bits[b >> 6] & ((1ull << (b & 63)) - 1));
  kept[i] = k;
}

} // namespace sq

extern "C" {

// one iteration of the for_await loop  [ref: filter.rs:16-24]
int sqlrs_filter_push(sqlrs_filter_t *f, const sqlrs_batch_t *in, int out_mem, sqlrs_batch_t **out) {
  return guard(f->ctx, [&] {
    Ctx *ctx = f->ctx;
    SQ_HIP(hipSetDevice(ctx->device));
    InBatch ib(ctx, in);
    *out = emit_batch(ctx, filter_batch(f, ib, nullptr), out_mem);
  });
}

} // extern "C"

namespace sq {
// ---- one small HOST batch, one launch, no copy call (small_async.hpp) ------------------------------------------------
// mask = `column OP constant [AND column OP constant ...]` (up to four terms; RowFilter: the stand-alone filter's comparison;
// a row is kept when EVERY term is true — a NULL term is not, filter.rs:16-24 over Kleene AND); every column of the batch
// compacted through the same positions; validity bitmaps re-packed in output order.
constexpr int SA_TERMS = 4;
struct SaFilterConj {
  int pred_col[SA_TERMS];
  int pred_is32[SA_TERMS]; // an int32 column: compared through its sign-extended value
  RowFilter rf[SA_TERMS];  // (rf.col unused: the predicate columns are read from the slot)
};
struct SaFilterParams {
  SaLayout lay;
  int nterms; // > 0: the conjunction `conj`; 0: the postfix program `prog` (any other predicate over fixed-width columns)
  union {
    SaFilterConj conj;
    SaProgram prog;
  };
  SaFilterParams() : nterms(0), prog() {}
  const uint8_t *in;
  uint8_t *out;
  unsigned long long seq;
};
__global__ __launch_bounds__(1024) void sa_filter_kernel(SaGroup<SaFilterParams> grp) {
  const SaFilterParams &p = grp.p[blockIdx.x]; // (one workgroup per batch of the group)
  __shared__ uint32_t s_w[17], s_nulls[SA_MAX_COLS];
  __shared__ uint8_t s_v[SA_MAX_ROWS];
  __shared__ uint32_t s_len[SA_MAX_ROWS + 4]; // Utf8: lengths of the kept rows in output order, then their exclusive prefix
  __shared__ uint32_t s_div0;
  if (threadIdx.x < SA_MAX_COLS) s_nulls[threadIdx.x] = 0;
  if (threadIdx.x == 0) s_div0 = 0;
  __syncthreads();
  uint32_t pos[4], total;
  const uint32_t bits = sa_positions(
      p.lay.rows,
      [&](uint32_t r, int) {
        if (p.nterms == 0) { // the general predicate: kept = the program's value is TRUE (NULL -> dropped, filter.rs:16-24)
          bool valid, div0 = false;
          const unsigned long long v = sa_eval_row(p.prog, p.lay, p.in, r, &valid, &div0);
          if (div0) s_div0 = 1u;
          return valid && v != 0;
        }
        bool keep = true;
        for (int q = 0; q < p.nterms; q++) { // (uniform trip count)
          const SaCol &pc = p.lay.c[p.conj.pred_col[q]];
          const uint8_t *pvalid = pc.in_voff != SA_NONE ? p.in + pc.in_voff : nullptr;
          const uint64_t v = p.conj.pred_is32[q] ? (uint64_t)(int64_t)((const int32_t *)(p.in + pc.in_off))[r] : ((const uint64_t *)(p.in + pc.in_off))[r];
          keep = keep && (!pvalid || ((pvalid[r >> 3] >> (r & 7)) & 1)) && row_passes(p.conj.rf[q], v);
        }
        return keep;
      },
      pos, s_w, &total);
  for (int c = 0; c < p.lay.ncols; c++) { // (uniform loop: the layout is a kernel argument)
    const SaCol &col = p.lay.c[c];
    const uint8_t *valid = col.in_voff != SA_NONE ? p.in + col.in_voff : nullptr;
    if (col.dtype == SQLRS_UTF8) { // lengths -> exclusive prefix (the output offsets) -> the bytes, row by row
      const int32_t *ioff = (const int32_t *)(p.in + col.in_off);
      int32_t *ooff = (int32_t *)(p.out + col.out_off);
#pragma unroll
      for (int t = 0; t < 4; t++) {
        if (!((bits >> t) & 1)) continue;
        const uint32_t r = (uint32_t)t * 1024u + threadIdx.x;
        s_len[pos[t]] = (uint32_t)(ioff[r + 1] - ioff[r]);
        if (valid) s_v[pos[t]] = (valid[r >> 3] >> (r & 7)) & 1;
      }
      __syncthreads();
      { // thread i owns entries 4 i .. 4 i + 3 of the (<= 4096) lengths
        uint32_t a[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint32_t i = threadIdx.x * 4 + q;
          a[q] = i < total ? s_len[i] : 0u;
          sum += a[q];
        }
        const uint32_t inc = wave_iscan_u32(sum);
        if (lane_id() == 63) s_w[wave_id()] = inc;
        __syncthreads();
        uint32_t base = inc - sum;
        for (int q = 0; q < wave_id(); q++) base += s_w[q];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint32_t i = threadIdx.x * 4 + q;
          if (i <= total) { // (entry `total` = the end of the last string)
            s_len[i] = base;
            ooff[i] = (int32_t)base;
          }
          base += a[q];
        }
        if (total == SA_MAX_ROWS && threadIdx.x == 1023) ooff[SA_MAX_ROWS] = (int32_t)base; // (a full batch kept whole: its end offset has no owner above)
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < 4; t++) {
        if (!((bits >> t) & 1)) continue;
        const uint32_t r = (uint32_t)t * 1024u + threadIdx.x;
        const uint8_t *src = p.in + col.in_data + ((uint32_t)ioff[r] - col.data_base);
        uint8_t *dst = p.out + col.out_data + s_len[pos[t]];
        const uint32_t len = (uint32_t)(ioff[r + 1] - ioff[r]);
        for (uint32_t b = 0; b < len; b++) dst[b] = src[b];
      }
      if (valid) sa_pack_validity(s_v, total, p.out + col.out_voff, &s_nulls[c]);
      else __syncthreads(); // (s_len is reused by the next Utf8 column)
      continue;
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
      if (!((bits >> t) & 1)) continue;
      const uint32_t r = (uint32_t)t * 1024u + threadIdx.x;
      if (col.width == 8) ((uint64_t *)(p.out + col.out_off))[pos[t]] = ((const uint64_t *)(p.in + col.in_off))[r];
      else ((uint32_t *)(p.out + col.out_off))[pos[t]] = ((const uint32_t *)(p.in + col.in_off))[r];
      if (valid) s_v[pos[t]] = (valid[r >> 3] >> (r & 7)) & 1;
    }
    if (valid) sa_pack_validity(s_v, total, p.out + col.out_voff, &s_nulls[c]);
  }
  sa_publish((SaHeader *)p.out, p.seq, total, s_nulls, p.lay.ncols, s_div0);
}
static void sa_filter_launch(SaRing *r, Ctx *ctx) {
  SaGroup<SaFilterParams> g;
  for (int i = 0; i < r->pend_n; i++) std::memcpy(&g.p[i], r->pend_buf + (size_t)i * SA_PARAM_MAX, sizeof(SaFilterParams));
  sa_filter_kernel<<<dim3((unsigned)r->pend_n), dim3(1024), 0, r->stream_of(r->pend_first_slot)>>>(g); // (reads nothing the ctx stream produces)
  SQ_HIP(hipGetLastError());
}
// the shapes the fast path evaluates: t1 [t2 AND [t3 AND [t4 AND]]] in postfix, every term INPUT_REF CONSTANT CMP over an
// int32 / int64 / float64 column with a constant of the column's type
static bool sa_filter_term(const sqlrs_expr_node_t *nd, const sqlrs_batch_t *in, int *pred_col, int *is32, RowFilter *rf) {
  const sqlrs_expr_node_t &a = nd[0], &b = nd[1], &o = nd[2];
  if (a.op != SQLRS_EXPR_INPUT_REF || b.op != SQLRS_EXPR_CONSTANT || b.is_null) return false;
  if (o.op < SQLRS_EXPR_GT || o.op > SQLRS_EXPR_NOTEQ) return false;
  if (a.index < 0 || a.index >= in->num_columns) return false;
  const int32_t dt = in->columns[a.index].dtype;
  if (dt != b.dtype || (dt != SQLRS_INT64 && dt != SQLRS_FLOAT64 && dt != SQLRS_INT32)) return false;
  *pred_col = a.index;
  *is32 = dt == SQLRS_INT32;
  rf->col = nullptr;
  rf->is_f64 = dt == SQLRS_FLOAT64;
  if (rf->is_f64) {
    uint64_t bits;
    std::memcpy(&bits, &b.f, 8);
    rf->kord = (bits >> 63) ? ~bits : (bits | (1ull << 63)); // f64_to_ordered (hashagg_op.hip, fusable_row_filter)
  } else
    rf->kord = (uint64_t)(dt == SQLRS_INT32 ? (int64_t)(int32_t)b.i : b.i) ^ (1ull << 63);
  static const uint32_t masks[6] = {4, 1, 6, 3, 2, 5}; // GT, LT, GTEQ, LTEQ, EQ, NOTEQ: keep if {<, ==, >}
  rf->keep_mask = masks[o.op - SQLRS_EXPR_GT];
  return true;
}
static bool sa_filter_shape(const Expr &e, const sqlrs_batch_t *in, SaFilterParams *p) {
  const size_t nn = e.nodes.size();
  if (!in || nn < 3 || (nn - 3) % 4 != 0) return false;
  const size_t nterms = 1 + (nn - 3) / 4;
  if (nterms > (size_t)SA_TERMS) return false;
  for (size_t k = 0; k < nterms; k++) {
    const size_t at = k == 0 ? 0 : 3 + (k - 1) * 4;
    if (k > 0 && e.nodes[at + 3].op != SQLRS_EXPR_AND) return false;
    if (!sa_filter_term(&e.nodes[at], in, &p->conj.pred_col[k], &p->conj.pred_is32[k], &p->conj.rf[k])) return false;
  }
  p->nterms = (int)nterms;
  return true;
}
// ... and any other predicate over the batch's int32 / int64 / float64 columns, as a postfix program (small_async.hpp)
static bool sa_filter_program(const Expr &e, const sqlrs_batch_t *in, SaFilterParams *p) {
  p->nterms = 0;
  return sa_compile(e, in, &p->prog) && p->prog.result_dtype == SQLRS_BOOLEAN;
}
} // namespace sq

extern "C" {

// sqlrs_filter_push without the wait: *ticket stands for the HOST batch sqlrs_filter_push(f, in, SQLRS_MEM_HOST, ..) would
// return (small_async.hpp; sqlrs_batch_wait hands it out).  `in` is read completely before the call returns, as for push.
int sqlrs_filter_push_async(sqlrs_filter_t *f, const sqlrs_batch_t *in, sqlrs_ticket_t **ticket) {
  if (ticket) *ticket = nullptr;
  return guard(f->ctx, [&] {
    Ctx *ctx = f->ctx;
    if (!ticket) fail(SQLRS_ERR_INTERNAL, "push_async: null ticket");
    SQ_HIP(hipSetDevice(ctx->device));
    auto t = std::unique_ptr<sqlrs_ticket>(new sqlrs_ticket());
    t->ctx = ctx;
    SaFilterParams p;
    const char *off_e = hook("SQLRS_ASYNC_FAST"); // test hook, read per call: 0 = every batch through the synchronous operator
    if (!(off_e && off_e[0] == '0') && (sa_filter_shape(f->expr, in, &p) || sa_filter_program(f->expr, in, &p))) {
      SaRing *r = sa_ring(ctx);
      const int slot = r ? sa_take_slot(r) : -1;
      if (slot >= 0) {
        if (sa_stage_input(in, r->in_area(slot), &p.lay, 0, nullptr, true)) { // (Utf8 payload columns travel too)
          p.in = r->in_area(slot);
          p.out = r->out_area(slot);
          p.seq = ++r->seq;
          sa_enqueue(ctx, r, f, sa_filter_launch, p, slot);
          t->slot = slot;
          t->seq = p.seq;
          t->lay = p.lay;
          *ticket = t.release();
          return;
        }
        r->busy[slot] = false;
      }
    }
    sa_flush(ctx); // (tickets complete in issue order: what waits for a launch goes first)
    InBatch ib(ctx, in); // the synchronous operator, its batch parked in the ticket
    t->done = emit_batch(ctx, filter_batch(f, ib, nullptr), SQLRS_MEM_HOST);
    *ticket = t.release();
  });
}

// n iterations of that loop in one call — out[i] is exactly what sqlrs_filter_push(in[i]) returns (one output batch per
// input batch, empty ones included, filter.rs:15-24) — for the reference's batch shape: 1024-row HOST batches
// (storage/csv.rs:105).  One upload + one synchronisation per BATCH and column is ~70 us a call (14.5 Mrows/s); here the
// batches' fixed-width columns are appended to a pinned staging area, uploaded with one copy per column, filtered by
// ONE launch sequence, the kept rows come back with one copy per column, and every input batch's share of them (the
// kept rows before each batch boundary are counted on the device from the selection bits) is handed out as a batch of
// its own.  Batches the staging area does not take (DEVICE memory, Utf8 / Boolean columns, other out_mem) run through
// sqlrs_filter_push one by one.
int sqlrs_filter_push_many(sqlrs_filter_t *f, int n, const sqlrs_batch_t *const *in, int out_mem, sqlrs_batch_t **out) {
  return guard(f->ctx, [&] {
    Ctx *ctx = f->ctx;
    SQ_HIP(hipSetDevice(ctx->device));
    for (int i = 0; i < n; i++) out[i] = nullptr;
    if (n <= 0) return;
    if (!f->stage) {
      f->stage.reset(new HostStage());
      f->stage->ctx = ctx;
    }
    HostStage &st = *f->stage;
    bool stageable = out_mem == SQLRS_MEM_HOST && !st.has_schema && n > 1;
    int64_t total_rows = 0;
    for (int i = 0; i < n && stageable; i++) {
      stageable = st.accepts(in[i]) && in[i]->num_columns == in[0]->num_columns;
      for (int c = 0; c < in[i]->num_columns && stageable; c++) stageable = in[i]->columns[c].dtype == in[0]->columns[c].dtype;
      total_rows += in[i] ? in[i]->num_rows : 0;
    }
    if (!stageable || total_rows == 0 || total_rows > (1ll << 30)) {
      int i = 0;
      try {
        for (; i < n; i++) {
          InBatch ib(ctx, in[i]);
          out[i] = emit_batch(ctx, filter_batch(f, ib, nullptr), out_mem);
        }
      } catch (...) {
        for (int k = 0; k < i; k++) {
          sqlrs_batch_release(out[k]);
          out[k] = nullptr;
        }
        throw;
      }
      return;
    }
    std::vector<int64_t> bounds((size_t)n + 1, 0);
    try {
      for (int i = 0; i < n; i++) {
        st.append(in[i]);
        bounds[(size_t)i + 1] = bounds[(size_t)i] + in[i]->num_rows;
      }
    } catch (...) {
      f->stage.reset(); // (a half-staged call must not leave its schema and rows behind: the staged path would stay off)
      throw;
    }
    sqlrs_batch_t *dev = st.take(); // one upload per column
    struct Rel {
      sqlrs_batch_t *b;
      ~Rel() { if (b) sqlrs_batch_release(b); }
    } rel{dev};
    Selection sel;
    DBatch o;
    {
      InBatch ib(ctx, dev);
      o = filter_batch(f, ib, &sel);
    }
    // kept rows before every batch boundary
    BufP dbounds = ctx->alloc(8 * ((size_t)n + 1)), dkept = ctx->alloc(8 * ((size_t)n + 1));
    SQ_HIP(hipMemcpyAsync(dbounds->p, bounds.data(), 8 * ((size_t)n + 1), hipMemcpyHostToDevice, ctx->stream));
    kept_before_kernel<<<dim3((unsigned)ceil_div(n + 1, 256)), dim3(256), 0, ctx->stream>>>(
        sel.bits, sel.tile_off->as<uint64_t>(), total_rows, dbounds->as<int64_t>(), n + 1, sel.count, dkept->as<int64_t>());
    SQ_HIP(hipGetLastError());
    std::vector<int64_t> kept((size_t)n + 1);
    SQ_HIP(hipMemcpyAsync(kept.data(), dkept->p, 8 * ((size_t)n + 1), hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
    if (!all_fixed_width(o)) fail(SQLRS_ERR_INTERNAL, "filter_push_many: staged columns are fixed width");
    split_rows_to_host(ctx, o, kept, &f->pin_out, &f->pin_cap, n, out); // one copy per column, one slice per input batch
  });
}

void sqlrs_filter_destroy(sqlrs_filter_t *f) {
  if (f) sa_flush(f->ctx); // (a group of its batches may still wait for its launch; their tickets stay valid)
  delete f;
}

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
// inverse of sort_key_kernel for 8-byte kinds: the sorted keys ARE the sorted column
template <int KIND>
__global__ void unsort_key_kernel(const uint64_t *__restrict__ keys, int64_t n, int desc, uint64_t *__restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t u = desc ? ~keys[i] : keys[i];
  if (KIND == 0) out[i] = (uint64_t)ordered_to_i64(u);
  else out[i] = (uint64_t)__double_as_longlong(ordered_to_f64(u));
}
// Utf8 sort keys: LSD over 8-byte big-endian chunks of the strings (zero padded), preceded by a
// pass on the length so that a string sorts before its zero-extended twin (byte-wise
// lexicographic order, like arrow's).  NULL rows get key 0 (they tie; the validity pass places them).
__global__ void utf8_maxlen_kernel(const int32_t *__restrict__ off, int64_t n, unsigned int *mx) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  unsigned int l = i < n ? (unsigned int)(off[i + 1] - off[i]) : 0u;
  for (int m = 32; m >= 1; m >>= 1) {
    unsigned int o = (unsigned int)__shfl_xor((int)l, m, 64);
    l = o > l ? o : l;
  }
  if (lane_id() == 0 && l) atomicMax(mx, l);
}
// chunk < 0: key = string length
__global__ void utf8_chunk_key_kernel(const uint8_t *__restrict__ data, const int32_t *__restrict__ off,
                                      const uint64_t *__restrict__ validity,
                                      const uint32_t *__restrict__ perm, int64_t n, int chunk, int desc,
                                      uint64_t *__restrict__ keys) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  uint64_t u = 0;
  if (i < n) {
    uint32_t r = perm[i];
    if (!validity || ((validity[r >> 6] >> (r & 63)) & 1)) {
      int32_t a = off[r], len = off[r + 1] - a;
      if (chunk < 0) {
        u = (uint64_t)(uint32_t)len;
      } else {
        for (int k = 0; k < 8; k++) {
          int32_t p = chunk * 8 + k;
          u = (u << 8) | (p < len ? (uint64_t)data[a + p] : 0ull);
        }
      }
      if (desc) u = ~u;
    }
    keys[i] = u;
  }
}
// OR of (key ^ key[0]) over all rows: the bit positions on which the keys differ at all
__global__ void keys_diff_kernel(const uint64_t *__restrict__ keys, int64_t n, unsigned long long *diff_or) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  uint64_t d = (i < n) ? (keys[i] ^ keys[0]) : 0;
  for (int m = 32; m >= 1; m >>= 1) d |= shfl_xor_u64(d, m);
  if (lane_id() == 0 && d) atomicOr(diff_or, (unsigned long long)d);
}

__global__ void sort_valid_key_kernel(const uint64_t *__restrict__ validity,
                                      const uint32_t *__restrict__ perm, int64_t n,
                                      uint64_t *__restrict__ keys) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t r = perm[i];
  keys[i] = (validity[r >> 6] >> (r & 63)) & 1; // NULL (0) first
}


// Stable sort of the rows `perm` refers to by the Utf8 column `c` (byte-wise lexicographic, like
// arrow): LSD over the length and the strings' 8-byte big-endian chunks, chunks on which no two
// keys differ skipped.  NULL rows tie (key 0).  `keys` is n-element scratch.
void sort_perm_by_utf8(Ctx *ctx, const DCol &c, const uint64_t *valid, int desc, uint32_t *perm, uint64_t *keys,
                       int64_t n) {
  if (n <= 1) return;
  dim3 b(256);
  BufP mx = ctx->alloc_zero(8);
  int64_t n64 = (int64_t)round_up((size_t)n, 64);
  dim3 g64((unsigned)ceil_div(n64, 256));
  utf8_maxlen_kernel<<<g64, b, 0, ctx->stream>>>(c.offsets, n, mx->as<unsigned int>());
  SQ_HIP(hipGetLastError());
  int max_len = (int)ctx->fetch_value(mx->as<unsigned int>());
  int chunks = (max_len + 7) / 8;
  BufP diff = ctx->alloc(16);
  for (int ch = -1; ch < chunks; ch++) { // length first (least significant), then chunks
    int chunk = ch < 0 ? -1 : chunks - 1 - ch; // last chunk -> first chunk
    SQ_HIP(hipMemsetAsync(diff->p, 0, 16, ctx->stream));
    utf8_chunk_key_kernel<<<g64, b, 0, ctx->stream>>>(c.v<uint8_t>(), c.offsets, valid, perm, n, chunk, desc, keys);
    keys_diff_kernel<<<g64, b, 0, ctx->stream>>>(keys, n, diff->as<unsigned long long>());
    SQ_HIP(hipGetLastError());
    uint64_t d = ctx->fetch_value(diff->as<uint64_t>());
    if (!d) continue; // every key equal in this chunk: nothing to sort
    int lo = __builtin_ctzll(d) & ~7, hi = 64 - __builtin_clzll(d);
    radix_sort_pairs(ctx, keys, perm, n, lo, hi);
  }
}

} // namespace sq

struct sqlrs_order {
  Ctx *ctx = nullptr;
  HostStage hstage; // small HOST batches until one upload (host_stage.hpp)
  std::vector<Expr> exprs;
  std::vector<int> asc;
  std::vector<DBatch> batches;
  int64_t limit_hint = 0;      // sqlrs_order_set_limit: only the first `limit_hint` rows of the result will be read
  int64_t topk_candidates = 0; // rows the last finish() sorted because of the hint (0 = everything)
};

namespace sq {
// ---- ORDER BY ... LIMIT k: sort the candidates only -------------------------------------------------------------
// PhysicalLimit(PhysicalOrder(child)) reads the first offset + limit rows of the sorted result; the reference sorts
// everything (order.rs:27-66) and slices (limit.rs:12-80).  With the hint the operator picks a threshold T from a sample
// of the keys (65 536 evenly spaced rows, sorted; the sample's quantile at twice the wanted fraction plus a margin), keeps
// the rows with key <= T with the Filter operator's own one-pass compaction (ties at T included: every row that can be
// among the first k is kept), and sorts those — the sorted candidates ARE a prefix of the full result, stable ties and
// all.  Fewer candidates than k (the sample misjudged the column): the hint is ignored and everything is sorted.
template <int KIND>
__global__ void topk_sample_kernel(const void *__restrict__ vals, int64_t n, int64_t step, int64_t S, int desc,
                                   uint64_t *__restrict__ out) {
  const int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (s >= S) return;
  const int64_t i = min(s * step, n - 1);
  uint64_t u;
  if (KIND == 0) u = i64_to_ordered(((const int64_t *)vals)[i]);
  else if (KIND == 1) u = f64_to_ordered(((const double *)vals)[i]);
  else u = i64_to_ordered((int64_t)((const int32_t *)vals)[i]);
  out[s] = desc ? ~u : u;
}
} // namespace sq

extern "C" {
int sqlrs_order_create(sqlrs_ctx_t *ctx, int num_keys, const sqlrs_order_by_t *order_by, sqlrs_order_t **out);
int sqlrs_order_push(sqlrs_order_t *o, const sqlrs_batch_t *in);
int sqlrs_order_finish(sqlrs_order_t *o, int out_mem, sqlrs_batch_t **out);
void sqlrs_order_destroy(sqlrs_order_t *o);
}

// true = `out` holds the sorted candidates (a prefix of the full result with at least limit_hint rows)
static bool order_topk(sqlrs_order *o, DBatch &all, int kc, int out_mem, sqlrs_batch_t **out) {
  Ctx *ctx = o->ctx;
  const int64_t n = all.rows, L = o->limit_hint;
  const DCol &key = all.cols[(size_t)kc];
  const int desc = o->asc[0] ? 0 : 1;
  const int kind = key.dtype == SQLRS_INT64 ? 0 : key.dtype == SQLRS_FLOAT64 ? 1 : 2;
  // 1. threshold from a sorted sample
  const int64_t S = 65536, step = n / S;
  BufP sk = ctx->alloc(8 * (size_t)S), sv = ctx->alloc(4 * (size_t)S);
  {
    ProfScope ps(ctx, "order_topk_sample");
    dim3 g((unsigned)ceil_div(S, 256)), b(256);
    if (kind == 0) topk_sample_kernel<0><<<g, b, 0, ctx->stream>>>(key.values, n, step, S, desc, sk->as<uint64_t>());
    else if (kind == 1) topk_sample_kernel<1><<<g, b, 0, ctx->stream>>>(key.values, n, step, S, desc, sk->as<uint64_t>());
    else topk_sample_kernel<2><<<g, b, 0, ctx->stream>>>(key.values, n, step, S, desc, sk->as<uint64_t>());
    SQ_HIP(hipGetLastError());
    iota_u32(ctx, sv->as<uint32_t>(), S);
    radix_sort_pairs(ctx, sk->as<uint64_t>(), sv->as<uint32_t>(), S, 0, 64);
  }
  // (rows below the q-th of S sampled keys: q n / S on average with a deviation of sqrt(q) n / S — twice the wanted
  //  fraction plus 64 sample ranks is >= 8 deviations above k for every k)
  const int64_t q = std::min<int64_t>(S - 1, (int64_t)std::ceil((double)L * (double)S / (double)n * 2.0) + 64);
  uint64_t T = ctx->fetch_value(sk->as<uint64_t>() + q);
  if (desc) T = ~T; // back to the ascending image: the candidates of a descending order are the rows with image >= T
  // 2. the candidates: Filter(key <= value(T))  (>= for DESC), the operator's own fast path
  sqlrs_expr_node_t nodes[3];
  std::memset(nodes, 0, sizeof(nodes));
  nodes[0].op = SQLRS_EXPR_INPUT_REF;
  nodes[0].index = kc;
  nodes[1].op = SQLRS_EXPR_CONSTANT;
  nodes[1].dtype = key.dtype;
  if (kind == 1) { // (host forms of device_utils.hpp's ordered_to_f64 / ordered_to_i64)
    const uint64_t bits = (T >> 63) ? (T & ~(1ull << 63)) : ~T;
    std::memcpy(&nodes[1].f, &bits, 8);
  } else
    nodes[1].i = (int64_t)(T ^ (1ull << 63));
  nodes[2].op = desc ? SQLRS_EXPR_GTEQ : SQLRS_EXPR_LTEQ;
  sqlrs_expr_t fe{nodes, 3, 0};
  sqlrs_filter_t *f = nullptr;
  sqlrs_batch_t *view = nullptr, *kept = nullptr;
  sqlrs_order_t *tmp = nullptr;
  auto cleanup = [&] {
    if (tmp) sqlrs_order_destroy(tmp);
    if (f) sqlrs_filter_destroy(f);
    if (kept) sqlrs_batch_release(kept);
    if (view) sqlrs_batch_release(view);
  };
  bool done = false;
  try {
    DBatch copy = all; // (columns share their buffers; lent ones stay lent for the duration of this call)
    view = emit_batch(ctx, std::move(copy), SQLRS_MEM_DEVICE);
    int st = sqlrs_filter_create((sqlrs_ctx_t *)ctx, &fe, &f);
    if (st == SQLRS_OK) st = sqlrs_filter_push(f, view, SQLRS_MEM_DEVICE, &kept);
    if (st != SQLRS_OK) fail(st, ctx->last_error);
    if (kept && kept->num_rows >= L) {
      // 3. sort the candidates (same keys, same directions, no hint)
      std::vector<std::vector<sqlrs_expr_node_t>> kn;
      std::vector<sqlrs_order_by_t> ob;
      for (size_t k = 0; k < o->exprs.size(); k++) {
        kn.push_back(o->exprs[k].nodes);
        for (size_t q2 = 0; q2 < kn.back().size(); q2++)
          kn.back()[q2].s = o->exprs[k].strings[q2].empty() ? nullptr : o->exprs[k].strings[q2].c_str();
      }
      for (size_t k = 0; k < o->exprs.size(); k++) {
        sqlrs_order_by_t e;
        e.expr = sqlrs_expr_t{kn[k].data(), (int32_t)kn[k].size(), 0};
        e.asc = o->asc[k];
        e.reserved = 0;
        ob.push_back(e);
      }
      st = sqlrs_order_create((sqlrs_ctx_t *)ctx, (int)ob.size(), ob.data(), &tmp);
      if (st == SQLRS_OK) st = sqlrs_order_push(tmp, kept);
      if (st == SQLRS_OK) st = sqlrs_order_finish(tmp, out_mem, out);
      if (st != SQLRS_OK) fail(st, ctx->last_error);
      o->topk_candidates = kept->num_rows;
      done = true;
    }
  } catch (...) {
    cleanup();
    throw;
  }
  cleanup();
  return done;
}

// the columns of `all` listed in `which`, gathered by the permutation `perm` of a fast Order route: plain 8-byte columns
// without NULLs travel together, 2..4 at a time as packed rows (one random line fetch per row for all of them instead of
// one per column: gather.hip), everything else column by column.  out[ci] is set for every ci in `which`.
static void gather_by_perm(Ctx *ctx, const DBatch &all, const std::vector<size_t> &which, const BufP &perm, int64_t n,
                           std::vector<DCol> *out) {
  std::vector<size_t> plain, rest;
  for (size_t ci : which) {
    const DCol &c = all.cols[ci];
    if (width_of(c.dtype) == 8 && c.stride != 0 && !(c.validity && c.null_count != 0) && c.length == all.rows) plain.push_back(ci);
    else rest.push_back(ci);
  }
  size_t at = 0;
  while (plain.size() - at >= 2) {
    const size_t left = plain.size() - at, k = left == 5 ? 3 : std::min<size_t>(4, left); // (5 = 3 + 2, not 4 + a single one)
    std::vector<DCol> grp;
    for (size_t q = 0; q < k; q++) grp.push_back(all.cols[plain[at + q]]);
    if (!gather_columns_packed(ctx, grp, all.rows, perm->as<uint32_t>(), n)) break;
    for (size_t q = 0; q < k; q++) (*out)[plain[at + q]] = grp[q];
    at += k;
  }
  for (; at < plain.size(); at++) rest.push_back(plain[at]);
  for (size_t ci : rest) (*out)[ci] = gather_column(ctx, all.cols[ci], perm->p, false, nullptr, n);
}

// ---- ONE key with NULLs that the composite key cannot take (a double; an int64 whose range needs all 64 bits): the NULL rows
// first, in input order (order.rs:33-41: nulls_first whatever the direction), then the rest ordered by the routes for keys
// without NULLs.  Both parts are order-preserving compactions of every column (the Filter operator's), the valid part goes
// through an inner Order on the same key, and the two are concatenated.
static bool order_null_split(sqlrs_order *o, DBatch &all, int kc, int out_mem, sqlrs_batch_t **out) {
  Ctx *ctx = o->ctx;
  const int64_t n = all.rows;
  const DCol &key = all.cols[(size_t)kc];
  const int64_t nulls = count_nulls(ctx, key);
  if (nulls == 0 || n - nulls < (1 << 20)) return false; // (nothing to split off / the rest is small: general path)
  Selection sv = selection_from_set_bits(ctx, key.validity, n); // (a masked copy: the caller's padding bits are unspecified)
  Selection sn = selection_from_clear_bits(ctx, key.validity, n);
  if (sv.count + sn.count != n) fail(SQLRS_ERR_INTERNAL, "order: NULL / valid rows do not add up");
  DBatch vb, nb;
  vb.rows = sv.count;
  nb.rows = sn.count;
  for (DCol &c : all.cols) {
    if (c.stride == 0) c = materialize_scalar(ctx, c, n);
    vb.cols.push_back(compact_column(ctx, c, sv));
    nb.cols.push_back(compact_column(ctx, c, sn));
  }
  DCol &vk = vb.cols[(size_t)kc];
  vk.validity = nullptr;
  vk.own_validity = nullptr;
  vk.null_count = 0;
  sqlrs_order_t *tmp = nullptr;
  sqlrs_batch_t *sorted = nullptr;
  auto cleanup = [&] {
    if (sorted) sqlrs_batch_release(sorted);
    if (tmp) sqlrs_order_destroy(tmp);
  };
  try {
    std::vector<sqlrs_expr_node_t> kn = o->exprs[0].nodes;
    for (size_t q = 0; q < kn.size(); q++) kn[q].s = o->exprs[0].strings[q].empty() ? nullptr : o->exprs[0].strings[q].c_str();
    sqlrs_order_by_t ob{sqlrs_expr_t{kn.data(), (int32_t)kn.size(), 0}, o->asc[0], 0};
    int st = sqlrs_order_create((sqlrs_ctx_t *)ctx, 1, &ob, &tmp);
    if (st != SQLRS_OK) fail(st, ctx->last_error);
    tmp->batches.push_back(std::move(vb));
    st = sqlrs_order_finish(tmp, SQLRS_MEM_DEVICE, &sorted);
    if (st != SQLRS_OK) fail(st, ctx->last_error);
    InBatch ib(ctx, sorted);
    DBatch r;
    r.rows = n;
    for (size_t c = 0; c < all.cols.size(); c++) {
      std::vector<const DCol *> parts{&nb.cols[c], &ib.col((int)c)};
      r.cols.push_back(concat_columns(ctx, parts));
    }
    *out = emit_batch(ctx, std::move(r), out_mem);
  } catch (...) {
    cleanup();
    throw;
  }
  cleanup();
  return true;
}

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

static int order_push_device(sqlrs_order_t *o, const sqlrs_batch_t *in, bool retained = false) {
  return guard(o->ctx, [&] {
    SQ_HIP(hipSetDevice(o->ctx->device));
    InBatch ib(o->ctx, in);
    o->batches.push_back(ib.materialize(!retained)); // retained: borrowed device columns stay where they are
  });
}
static int order_flush_host(sqlrs_order_t *o) {
  if (!o->hstage.has_schema) return SQLRS_OK;
  sqlrs_batch_t *dev = nullptr;
  int st = guard(o->ctx, [&] {
    SQ_HIP(hipSetDevice(o->ctx->device));
    dev = o->hstage.take();
  });
  if (st != SQLRS_OK) return st;
  st = order_push_device(o, dev);
  sqlrs_batch_release(dev);
  return st;
}
// [ref: order.rs:19-26]
int sqlrs_order_push(sqlrs_order_t *o, const sqlrs_batch_t *in) {
  o->hstage.ctx = o->ctx;
  if (o->hstage.accepts(in)) {
    int st = guard(o->ctx, [&] { o->hstage.append(in); });
    if (st != SQLRS_OK || o->hstage.rows < HOST_STAGE_FLUSH_ROWS) return st;
    return order_flush_host(o);
  }
  int st = order_flush_host(o);
  return st != SQLRS_OK ? st : order_push_device(o, in);
}

// [ref: order.rs:27-66] concat -> lexsort_to_indices (nulls first, stable) -> take
// [ref: order.rs:19-26 pushes the child's batches — Arc'd arrays — into a Vec: nothing is copied]
int sqlrs_order_push_retained(sqlrs_order_t *o, const sqlrs_batch_t *in) {
  o->hstage.ctx = o->ctx;
  int st = order_flush_host(o);
  return st != SQLRS_OK ? st : order_push_device(o, in, true);
}

int sqlrs_order_finish(sqlrs_order_t *o, int out_mem, sqlrs_batch_t **out) {
  int stf = order_flush_host(o);
  if (stf != SQLRS_OK) return stf;
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
    // ---- ORDER BY ... LIMIT k (sqlrs_order_set_limit): the first key a plain fixed-width column without NULLs, k a small part of the rows
    o->topk_candidates = 0;
    {
      const char *tk_e = hook("SQLRS_ORDER_TOPK"); // test / tuning hook, read per call: 0 = ignore the hint, 1 = whatever the sizes
      const int tk = tk_e ? std::atoi(tk_e) : -1;
      // (several keys: the threshold is taken on the FIRST one — a row whose first key lies beyond k rows' first keys
      //  cannot be among the first k whatever the other keys say; the candidates are sorted on all of them)
      if (o->limit_hint > 0 && tk != 0 && !o->exprs.empty() && o->exprs[0].nodes.size() == 1 && o->exprs[0].nodes[0].op == SQLRS_EXPR_INPUT_REF &&
          (tk == 1 ? n >= (1 << 17) : (n >= (1 << 20) && o->limit_hint <= n / 16))) {
        const int kc0 = o->exprs[0].nodes[0].index;
        if (kc0 >= 0 && (size_t)kc0 < all.cols.size()) {
          const DCol &kcol = all.cols[(size_t)kc0];
          const bool plain = (kcol.dtype == SQLRS_INT64 || kcol.dtype == SQLRS_FLOAT64 || kcol.dtype == SQLRS_INT32) && kcol.stride != 0 &&
                             !(kcol.validity && kcol.null_count != 0);
          if (plain && order_topk(o, all, kc0, out_mem, out)) return;
        }
      }
    }
    // ---- fast route: one plain key column without NULLs (order_fast.hip)
    const char *fast_e = hook("SQLRS_ORDER_FAST"); // (test / A-B hook, read per call: 0 = the general path)
    const bool fast_on = !(fast_e && fast_e[0] == '0');
    if (fast_on && o->exprs.size() == 1 && o->exprs[0].nodes.size() == 1 && o->exprs[0].nodes[0].op == SQLRS_EXPR_INPUT_REF) {
      const int kc = o->exprs[0].nodes[0].index;
      if (kc >= 0 && (size_t)kc < all.cols.size()) {
        int cc = -1; // one other 8-byte column without NULLs travels with the rows; the rest is gathered
        for (size_t ci = 0; ci < all.cols.size() && cc < 0; ci++) {
          const DCol &c = all.cols[ci];
          if ((int)ci != kc && width_of(c.dtype) == 8 && c.stride != 0 && !(c.validity && c.null_count != 0)) cc = (int)ci;
        }
        const bool want_perm = all.cols.size() > (size_t)(cc >= 0 ? 2 : 1);
        DCol ko, co;
        BufP fperm;
        // (the fast route holds ~48 B of scratch per row, the general path ~24: when the pool cannot give the former,
        //  the ORDER still runs — an allocation failure of the attempt is "route not taken", nothing was produced)
        bool fast = false;
        try {
          bool in_order = false;
          fast = order_fast(ctx, all.cols[(size_t)kc], o->asc[0] ? 0 : 1, cc >= 0 ? &all.cols[(size_t)cc] : nullptr, n, &ko, &co, &fperm, want_perm,
                            &in_order);
          if (in_order) { // the rows arrived in the requested order (ties included: a stable sort is the identity)
            for (DCol &c : all.cols) { // (columns of a retained push are only lent until this call returns: one copy)
              if (c.stride == 0) c = materialize_scalar(ctx, c, n);
              const bool borrowed = (c.values && !c.own_values) || (c.validity && !c.own_validity) || (c.offsets && !c.own_offsets);
              if (borrowed) c = copy_column(ctx, c);
            }
            *out = emit_batch(ctx, std::move(all), out_mem);
            return;
          }
        } catch (const Error &e) {
          if (e.status != SQLRS_ERR_DEVICE || e.msg.rfind("hipMalloc(", 0) != 0) throw;
          (void)hipGetLastError(); // (clears the sticky out-of-memory error code)
          ko = DCol();
          co = DCol();
          fperm = nullptr;
        }
        if (fast) {
          DBatch r;
          r.rows = n;
          r.cols.resize(all.cols.size());
          std::vector<size_t> togather;
          for (size_t ci = 0; ci < all.cols.size(); ci++) {
            if ((int)ci == kc) r.cols[ci] = ko;
            else if ((int)ci == cc && co.values) r.cols[ci] = co; // (wide keys + more columns: the row ids travelled instead)
            else togather.push_back(ci);
          }
          if (!togather.empty()) gather_by_perm(ctx, all, togather, fperm, n, &r.cols);
          *out = emit_batch(ctx, std::move(r), out_mem);
          return;
        }
      }
    }
    // ---- several plain integer keys whose ranges together fit 64 bits: one composite key through the same routes
    // (also ONE integer key with NULLs, which the route above leaves alone: a valid bit in front of the value, NULLs first)
    if (fast_on && o->exprs.size() >= 1 && o->exprs.size() <= 4 && n >= (1 << 20)) {
      std::vector<int> kcs;
      for (const Expr &e : o->exprs) {
        if (e.nodes.size() != 1 || e.nodes[0].op != SQLRS_EXPR_INPUT_REF || e.nodes[0].index < 0 || (size_t)e.nodes[0].index >= all.cols.size()) break;
        kcs.push_back(e.nodes[0].index);
      }
      if (kcs.size() == o->exprs.size()) {
        auto is_key = [&](size_t ci) { return std::find(kcs.begin(), kcs.end(), (int)ci) != kcs.end(); };
        int cc = -1, others = 0;
        for (size_t ci = 0; ci < all.cols.size(); ci++) {
          if (is_key(ci)) continue;
          others++;
          const DCol &c = all.cols[ci];
          if (cc < 0 && width_of(c.dtype) == 8 && c.stride != 0 && !(c.validity && c.null_count != 0)) cc = (int)ci;
        }
        const bool want_perm = others > (cc >= 0 ? 1 : 0);
        std::vector<const DCol *> kp;
        std::vector<int> kd;
        for (size_t k = 0; k < kcs.size(); k++) {
          kp.push_back(&all.cols[(size_t)kcs[k]]);
          kd.push_back(o->asc[k] ? 0 : 1);
        }
        std::vector<DCol> kout;
        DCol co;
        BufP fperm;
        bool done = false, in_order = false;
        try {
          done = order_composite(ctx, kp, kd, cc >= 0 ? &all.cols[(size_t)cc] : nullptr, n, &kout, &co, &fperm, want_perm, &in_order);
        } catch (const Error &e) { // (an allocation failure of the attempt is "route not taken", as above)
          if (e.status != SQLRS_ERR_DEVICE || e.msg.rfind("hipMalloc(", 0) != 0) throw;
          (void)hipGetLastError();
          kout.clear();
          co = DCol();
          fperm = nullptr;
        }
        if (in_order) { // the rows arrived in the requested order
          for (DCol &c : all.cols) {
            if (c.stride == 0) c = materialize_scalar(ctx, c, n);
            const bool borrowed = (c.values && !c.own_values) || (c.validity && !c.own_validity) || (c.offsets && !c.own_offsets);
            if (borrowed) c = copy_column(ctx, c);
          }
          *out = emit_batch(ctx, std::move(all), out_mem);
          return;
        }
        if (done) {
          DBatch r;
          r.rows = n;
          r.cols.resize(all.cols.size());
          std::vector<size_t> togather;
          for (size_t ci = 0; ci < all.cols.size(); ci++) {
            const auto at = std::find(kcs.begin(), kcs.end(), (int)ci);
            if (at != kcs.end()) r.cols[ci] = kout[(size_t)(at - kcs.begin())];
            else if ((int)ci == cc && co.values) r.cols[ci] = co;
            else togather.push_back(ci);
          }
          if (!togather.empty()) gather_by_perm(ctx, all, togather, fperm, n, &r.cols);
          *out = emit_batch(ctx, std::move(r), out_mem);
          return;
        }
      }
    }
    // ---- one key with NULLs (what the composite key above did not take): NULL rows first, the rest through an inner Order
    if (fast_on && o->exprs.size() == 1 && o->exprs[0].nodes.size() == 1 && o->exprs[0].nodes[0].op == SQLRS_EXPR_INPUT_REF && n >= (1 << 21)) {
      const int kc = o->exprs[0].nodes[0].index;
      if (kc >= 0 && (size_t)kc < all.cols.size()) {
        const DCol &kcol = all.cols[(size_t)kc];
        const bool plain = (kcol.dtype == SQLRS_INT64 || kcol.dtype == SQLRS_FLOAT64 || kcol.dtype == SQLRS_INT32) && kcol.stride != 0;
        if (plain && kcol.validity && kcol.null_count != 0 && order_null_split(o, all, kc, out_mem, out)) return;
      }
    }
    BufP perm = ctx->alloc(4 * (size_t)n1), keys = ctx->alloc(8 * (size_t)n1);
    iota_u32(ctx, perm->as<uint32_t>(), n);
    dim3 g((unsigned)ceil_div(n1, 256)), b(256);
    // When the most significant sort key is a plain non-NULL 8-byte column, its sorted values are
    // the inverse image of the sorted keys: that column needs no random gather (2 ms per 1e8 rows).
    int key_col = -1, key_kind = 0, key_desc = 0;
    // LSD over the sort columns: last key first, each step a stable sort
    for (int k = (int)o->exprs.size() - 1; k >= 0 && n > 1; k--) {
      DCol c = eval_expr(ctx, o->exprs[(size_t)k], colfn, n, true);
      const uint64_t *valid = (c.validity && c.null_count != 0) ? c.validity : nullptr;
      int desc = o->asc[(size_t)k] ? 0 : 1;
      int bits = 64;
      if (k == 0 && !valid && (c.dtype == SQLRS_INT64 || c.dtype == SQLRS_FLOAT64) && c.stride != 0 &&
          o->exprs[0].nodes.size() == 1 && o->exprs[0].nodes[0].op == SQLRS_EXPR_INPUT_REF) {
        key_col = o->exprs[0].nodes[0].index;
        key_kind = c.dtype == SQLRS_FLOAT64 ? 1 : 0;
        key_desc = desc;
      }
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
      case SQLRS_UTF8: {
        sort_perm_by_utf8(ctx, c, valid, desc, perm->as<uint32_t>(), keys->as<uint64_t>(), n);
        bits = 0; // sorted above
        break;
      }
      default:
        fail(SQLRS_ERR_INTERNAL, "unsupported sort key type");
      }
      SQ_HIP(hipGetLastError());
      if (bits) radix_sort_pairs(ctx, keys->as<uint64_t>(), perm->as<uint32_t>(), n, 0, bits);
      if (valid) { // nulls_first = true regardless of direction (order.rs:37-40)
        sort_valid_key_kernel<<<g, b, 0, ctx->stream>>>(valid, perm->as<uint32_t>(), n, keys->as<uint64_t>());
        SQ_HIP(hipGetLastError());
        radix_sort_pairs(ctx, keys->as<uint64_t>(), perm->as<uint32_t>(), n, 0, 8);
      }
    }
    DBatch r;
    r.rows = n;
    for (size_t ci = 0; ci < all.cols.size(); ci++) {
      const DCol &c = all.cols[ci];
      if ((int)ci == key_col) {
        DCol oc;
        oc.dtype = c.dtype;
        oc.length = n;
        oc.null_count = 0;
        oc.own_values = ctx->alloc(8 * (size_t)n1);
        oc.values = oc.own_values->p;
        if (key_kind == 0) unsort_key_kernel<0><<<g, b, 0, ctx->stream>>>(keys->as<uint64_t>(), n, key_desc, oc.own_values->as<uint64_t>());
        else unsort_key_kernel<1><<<g, b, 0, ctx->stream>>>(keys->as<uint64_t>(), n, key_desc, oc.own_values->as<uint64_t>());
        SQ_HIP(hipGetLastError());
        r.cols.push_back(std::move(oc));
      } else {
        r.cols.push_back(gather_column(ctx, c, perm->p, false, nullptr, n));
      }
    }
    *out = emit_batch(ctx, std::move(r), out_mem);
  });
}

int sqlrs_order_set_limit(sqlrs_order_t *o, int64_t rows) {
  return guard(o->ctx, [&] {
    if (rows < 0) fail(SQLRS_ERR_INTERNAL, "order limit hint must be >= 0");
    o->limit_hint = rows;
  });
}
int64_t sqlrs_order_topk_candidates(const sqlrs_order_t *o) { return o->topk_candidates; }
void sqlrs_order_destroy(sqlrs_order_t *o) { delete o; }

} // extern "C"
