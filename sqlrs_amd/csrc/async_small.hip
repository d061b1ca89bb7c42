// async_small.hip — ring, tickets and sqlrs_batch_wait of the single-batch async path (small_async.hpp).
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "small_async.hpp"

extern "C" void sqlrs_batch_release(sqlrs_batch_t *batch);

namespace sq {

SaRing *sa_ring(Ctx *ctx) {
  if (!ctx->small_ring) {
    static thread_local bool failed_once = false; // (a platform without coherent mapped host memory: the synchronous path, quietly)
    if (failed_once) return nullptr;
    auto r = std::make_shared<SaRing>();
    try {
    // coherent (fine-grained) + mapped: the kernel's stores are visible to a polling host thread while the stream runs on
    SQ_HIP(hipHostMalloc((void **)&r->pin, (size_t)SA_SLOTS * 2 * SA_AREA, hipHostMallocCoherent | hipHostMallocMapped));
    std::memset(r->pin, 0, (size_t)SA_SLOTS * 2 * SA_AREA);
    for (int i = 0; i < SA_STREAMS; i++) SQ_HIP(hipStreamCreateWithFlags(&r->side[i], hipStreamNonBlocking));
    SQ_HIP(hipEventCreateWithFlags(&r->order_ev, hipEventDisableTiming));
    if (const char *g = hook("SQLRS_ASYNC_GROUP")) r->group = std::min(SA_GROUP_MAX, std::max(1, std::atoi(g))); // test hook: batches per launch
    } catch (const Error &) {
      failed_once = true;
      return nullptr;
    }
    ctx->small_ring = r;
  }
  return (SaRing *)ctx->small_ring.get();
}
void sa_order_after_ctx(Ctx *ctx, SaRing *r) {
  SQ_HIP(hipEventRecord(r->order_ev, ctx->stream));
  for (int i = 0; i < SA_STREAMS; i++) SQ_HIP(hipStreamWaitEvent(r->side[i], r->order_ev, 0));
}
void sa_flush(Ctx *ctx) {
  SaRing *r = (SaRing *)ctx->small_ring.get();
  if (!r || !r->pend_n) return;
  r->pend_launch(r, ctx);
  r->pend_n = 0;
  r->launched_seq = r->pend_last_seq;
  r->dirty = true;
}
void sa_drain(Ctx *ctx) {
  SaRing *r = (SaRing *)ctx->small_ring.get();
  if (!r) return;
  sa_flush(ctx);
  if (!r->dirty) return;
  for (int i = 0; i < SA_STREAMS; i++) (void)hipStreamSynchronize(r->side[i]);
  r->dirty = false;
}
int sa_take_slot(SaRing *r) {
  for (int k = 0; k < SA_SLOTS; k++) {
    const int s = (r->next + k) % SA_SLOTS;
    if (!r->busy[s]) {
      r->busy[s] = true;
      r->next = (s + 1) % SA_SLOTS;
      return s;
    }
  }
  return -1;
}

static uint32_t sa_width(int32_t dtype) { return dtype == SQLRS_INT32 ? 4u : (dtype == SQLRS_INT64 || dtype == SQLRS_FLOAT64) ? 8u : 0u; }

bool sa_stage_input(const sqlrs_batch_t *in, uint8_t *area, SaLayout *lay, int first_out_col, const int32_t *front_dtypes, bool allow_utf8) {
  if (!in || in->num_rows < 0 || in->num_rows > (int64_t)SA_MAX_ROWS || in->num_columns <= 0 ||
      in->num_columns + first_out_col > SA_MAX_COLS)
    return false;
  const uint32_t rows = (uint32_t)in->num_rows, vbytes = (rows + 7) / 8;
  for (int c = 0; c < in->num_columns; c++) {
    const sqlrs_column_t &col = in->columns[c];
    if (col.mem != SQLRS_MEM_HOST || col.length != in->num_rows) return false;
    if (col.dtype == SQLRS_UTF8) {
      if (!allow_utf8 || !col.offsets || col.offsets[rows] < col.offsets[0] || (col.offsets[rows] > col.offsets[0] && !col.values)) return false;
    } else if (!sa_width(col.dtype) || (rows && !col.values))
      return false;
  }
  auto up64 = [](size_t x) { return (x + 63) & ~(size_t)63; };
  // output: header | per column values (`rows` bound both sides) + validity (+ the bytes of a Utf8 column)
  size_t in_at = 0, out_at = up64(sizeof(SaHeader));
  lay->ncols = first_out_col + in->num_columns;
  lay->rows = rows;
  for (int c = 0; c < lay->ncols; c++) {
    SaCol &d = lay->c[c];
    d.dtype = c < first_out_col ? front_dtypes[c] : in->columns[c - first_out_col].dtype;
    const bool utf8 = d.dtype == SQLRS_UTF8;
    d.width = utf8 ? 4u : sa_width(d.dtype);
    if (!d.width) return false;
    d.in_off = d.in_voff = d.in_data = d.out_data = SA_NONE;
    d.data_base = 0;
    const size_t nval = utf8 ? (size_t)rows + 1 : rows;
    d.out_off = (uint32_t)out_at;
    out_at = up64(out_at + (size_t)d.width * nval);
    d.out_voff = (uint32_t)out_at;
    out_at = up64(out_at + vbytes);
    if (c >= first_out_col) {
      const sqlrs_column_t &col = in->columns[c - first_out_col];
      d.in_off = (uint32_t)in_at;
      in_at = up64(in_at + (size_t)d.width * nval);
      if (col.validity && col.null_count != 0) {
        d.in_voff = (uint32_t)in_at;
        in_at = up64(in_at + vbytes + 8); // (+ 8: the kernel may read the bitmap in whole words)
      }
      if (utf8) {
        const size_t nbytes = (size_t)(col.offsets[rows] - col.offsets[0]);
        d.data_base = (uint32_t)col.offsets[0];
        d.in_data = (uint32_t)in_at;
        in_at = up64(in_at + nbytes);
        d.out_data = (uint32_t)out_at;
        out_at = up64(out_at + nbytes);
      }
    }
    if (in_at > SA_AREA || out_at > SA_AREA) return false;
  }
  for (int c = first_out_col; c < lay->ncols; c++) { // the only copies of the fast path: 4-32 KB per column, host to pinned host
    const sqlrs_column_t &col = in->columns[c - first_out_col];
    const SaCol &d = lay->c[c];
    if (d.dtype == SQLRS_UTF8) {
      std::memcpy(area + d.in_off, col.offsets, 4 * ((size_t)rows + 1));
      const size_t nbytes = (size_t)(col.offsets[rows] - col.offsets[0]);
      if (nbytes) std::memcpy(area + d.in_data, (const uint8_t *)col.values + col.offsets[0], nbytes);
    } else if (rows)
      std::memcpy(area + d.in_off, col.values, (size_t)d.width * rows);
    if (d.in_voff != SA_NONE) std::memcpy(area + d.in_voff, col.validity, vbytes);
  }
  return true;
}

bool sa_compile(const Expr &e, const sqlrs_batch_t *in, SaProgram *out) {
  if (!in || e.nodes.empty() || e.nodes.size() > (size_t)SA_PROG_MAX) return false;
  int32_t st[SA_STACK_MAX];
  int sp = 0;
  out->n = 0;
  auto numeric = [](int32_t d) { return d == SQLRS_INT32 || d == SQLRS_INT64 || d == SQLRS_FLOAT64; };
  for (size_t k = 0; k < e.nodes.size(); k++) {
    const sqlrs_expr_node_t &n = e.nodes[k];
    SaInstr I{};
    switch (n.op) {
    case SQLRS_EXPR_INPUT_REF: {
      if (n.index < 0 || n.index >= in->num_columns || sp >= SA_STACK_MAX) return false;
      const int32_t d = in->columns[n.index].dtype;
      if (!numeric(d)) return false; // (Boolean columns are bit-packed, Utf8 has no place on this stack)
      I.op = SAO_COL;
      I.dtype = (uint8_t)d;
      I.col = (uint32_t)n.index;
      st[sp++] = d;
      break;
    }
    case SQLRS_EXPR_CONSTANT: {
      if (sp >= SA_STACK_MAX || !(numeric(n.dtype) || n.dtype == SQLRS_BOOLEAN)) return false;
      I.op = SAO_CONST;
      I.dtype = (uint8_t)n.dtype;
      I.is_null = n.is_null ? 1 : 0;
      if (n.dtype == SQLRS_FLOAT64) std::memcpy(&I.imm, &n.f, 8);
      else if (n.dtype == SQLRS_INT32) I.imm = (unsigned long long)(long long)(int32_t)n.i;
      else if (n.dtype == SQLRS_BOOLEAN) I.imm = n.i ? 1ull : 0ull;
      else I.imm = (unsigned long long)n.i;
      st[sp++] = n.dtype;
      break;
    }
    case SQLRS_EXPR_TYPE_CAST: {
      if (sp < 1) return false;
      const int32_t from = st[sp - 1], to = n.dtype;
      if (from == to) continue; // (cast_col: the column as it is)
      if (!numeric(to) || !(numeric(from) || from == SQLRS_BOOLEAN)) return false;
      I.op = SAO_CAST;
      I.dtype = (uint8_t)to;
      I.from = (uint8_t)from;
      st[sp - 1] = to;
      break;
    }
    default: {
      if (sp < 2) return false;
      const int32_t r = st[sp - 1], l = st[sp - 2];
      if (n.op >= SQLRS_EXPR_PLUS && n.op <= SQLRS_EXPR_DIVIDE) {
        if (l != r || !numeric(l)) return false;
        I.op = (uint8_t)(SAO_ADD + (n.op - SQLRS_EXPR_PLUS));
        I.dtype = (uint8_t)l;
        st[sp - 2] = l;
      } else if (n.op >= SQLRS_EXPR_GT && n.op <= SQLRS_EXPR_NOTEQ) {
        if (l != r || !(numeric(l) || l == SQLRS_BOOLEAN)) return false;
        I.op = (uint8_t)(SAO_GT + (n.op - SQLRS_EXPR_GT));
        I.dtype = (uint8_t)l;
        st[sp - 2] = SQLRS_BOOLEAN;
      } else if (n.op == SQLRS_EXPR_AND || n.op == SQLRS_EXPR_OR) {
        if (l != SQLRS_BOOLEAN || r != SQLRS_BOOLEAN) return false;
        I.op = n.op == SQLRS_EXPR_AND ? SAO_AND : SAO_OR;
        I.dtype = SQLRS_BOOLEAN;
        st[sp - 2] = SQLRS_BOOLEAN;
      } else
        return false;
      sp--;
    }
    }
    out->ins[out->n++] = I;
  }
  if (sp != 1) return false;
  out->result_dtype = st[0];
  return true;
}

} // namespace sq

using namespace sq;

extern "C" {

// [ref: src/executor/mod.rs:34 — the consumer polls its child one batch at a time]  Blocks until the batch behind
// `ticket` exists, hands it out as a HOST batch (possibly NULL where the synchronous call would have set *out = NULL) and
// consumes the ticket.  Tickets of one ctx complete in the order they were issued.
int sqlrs_batch_wait(sqlrs_ticket_t *ticket, sqlrs_batch_t **out) {
  if (!ticket) return SQLRS_ERR_INTERNAL;
  Ctx *ctx = ticket->ctx;
  int st = guard(ctx, [&] {
    if (!out) fail(SQLRS_ERR_INTERNAL, "batch_wait: null argument");
    *out = nullptr;
    if (ticket->slot < 0) {
      *out = ticket->done;
      ticket->done = nullptr;
      return;
    }
    SaRing *r = sa_ring(ctx);
    if (!r) fail(SQLRS_ERR_INTERNAL, "batch_wait: a slot ticket without a ring");
    if (ticket->seq > r->launched_seq) { // its kernel is still waiting for its group to fill
      SQ_HIP(hipSetDevice(ctx->device));
      sa_flush(ctx);
    }
    const SaHeader *h = (const SaHeader *)r->out_area(ticket->slot);
    bool seen = false;
    for (int spin = 0; spin < 20000 && !seen; spin++) {
      seen = __atomic_load_n(&h->seq, __ATOMIC_ACQUIRE) == ticket->seq;
      if (!seen) __builtin_ia32_pause();
    }
    if (!seen) { // (not there yet — a long queue ahead, or stores that only a finished stream makes visible)
      SQ_HIP(hipSetDevice(ctx->device));
      for (int i = 0; i < SA_STREAMS; i++) SQ_HIP(hipStreamSynchronize(r->side[i]));
      if (__atomic_load_n(&h->seq, __ATOMIC_ACQUIRE) != ticket->seq) fail(SQLRS_ERR_DEVICE, "batch_wait: the batch's kernel left no result");
    }
    if (h->pad) fail(SQLRS_ERR_ARROW, "Divide by zero error"); // (what the synchronous evaluator raises at the push, expr.hip)
    const SaLayout &lay = ticket->lay;
    const uint8_t *oa = r->out_area(ticket->slot);
    const void *vals[SA_MAX_COLS];
    const uint8_t *valid[SA_MAX_COLS];
    const int32_t *offs[SA_MAX_COLS];
    int64_t nulls[SA_MAX_COLS];
    int32_t dts[SA_MAX_COLS];
    for (int c = 0; c < lay.ncols; c++) {
      const bool utf8 = lay.c[c].dtype == SQLRS_UTF8;
      vals[c] = oa + (utf8 ? lay.c[c].out_data : lay.c[c].out_off);
      offs[c] = utf8 ? (const int32_t *)(oa + lay.c[c].out_off) : nullptr;
      valid[c] = oa + lay.c[c].out_voff;
      nulls[c] = h->nulls[c];
      dts[c] = lay.c[c].dtype;
    }
    *out = emit_host_copy(ctx, lay.ncols, dts, (int64_t)h->count, vals, valid, nulls, offs);
  });
  if (ticket->slot >= 0 && ctx->small_ring) ((SaRing *)ctx->small_ring.get())->busy[ticket->slot] = false;
  if (ticket->done) sqlrs_batch_release(ticket->done); // (an error above: nothing leaks)
  delete ticket;
  return st;
}

} // extern "C"
