// plumbing.hip — the operators either side of the hot path: ProjectExecutor,
// LimitExecutor, SimpleAggExecutor, the sqllogictest text form of a batch, and CSV ingest.
// None of them is a hot loop in the reference; they exist here so that a whole query plan can
// stay device resident between the scan and the final rendering, and so that the reference's
// .slt expectations are compared in the reference's own text form.
#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <strings.h>

#include "common.hpp"
#include "host_stage.hpp"
#include "device_utils.hpp"
#include "prims.hpp"
#include "small_async.hpp"

using namespace sq;

extern "C" {
int sqlrs_hash_agg_create(sqlrs_ctx_t *, int, const sqlrs_expr_t *, int, const sqlrs_agg_func_t *, sqlrs_hash_agg_t **);
int sqlrs_hash_agg_push(sqlrs_hash_agg_t *, const sqlrs_batch_t *);
int sqlrs_hash_agg_finish(sqlrs_hash_agg_t *, int, sqlrs_batch_t **);
void sqlrs_hash_agg_destroy(sqlrs_hash_agg_t *);
void sqlrs_batch_release(sqlrs_batch_t *);
}

namespace sq {

__global__ void iota_offset_u32_kernel(uint32_t *out, int64_t n, uint32_t start) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = start + (uint32_t)i;
}
// cross join: output row k = (left row k / R, right row k % R)
__global__ void cross_index_kernel(uint32_t *left_idx, uint32_t *right_idx, int64_t n, uint32_t R, uint32_t left0) {
  int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (k < n) {
    left_idx[k] = left0 + (uint32_t)(k / R);
    right_idx[k] = (uint32_t)(k % R);
  }
}
__global__ void fill_zero_u64_kernel(uint64_t *out, int64_t n) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = 0;
}

} // namespace sq

// ========================================================================== Project ==
struct sqlrs_project {
  Ctx *ctx = nullptr;
  std::vector<Expr> exprs;
  std::unique_ptr<HostStage> stage; // sqlrs_project_push_many: the call's small HOST batches, uploaded together
  void *pin_out = nullptr;          // ... and where their projected rows land on the host (pinned)
  size_t pin_cap = 0;
  ~sqlrs_project() {
    if (pin_out) (void)hipHostFree(pin_out);
  }
};

// ======================================================================= CrossJoin ==
struct sqlrs_cross_join {
  Ctx *ctx = nullptr;
  std::vector<DBatch> left; // the left child's batches (private copies: the caller's buffers are only lent)
  DBatch left_all;          // ... concatenated once, at the first probe batch (cross_join.rs:36)
  bool concatenated = false;
};

// =========================================================================== Limit ==
struct sqlrs_limit {
  Ctx *ctx = nullptr;
  bool has_limit = false, done = false;
  uint64_t limit = 0, offset_val = 0, returned_count = 0;
};

// ======================================================================= SimpleAgg ==
struct sqlrs_simple_agg {
  Ctx *ctx = nullptr;
  sqlrs_hash_agg_t *inner = nullptr; // HashAgg over one constant key: same accumulators, one group
  std::vector<int32_t> funcs, return_dtypes;
  bool saw_batch = false;
  ~sqlrs_simple_agg() {
    if (inner) sqlrs_hash_agg_destroy(inner);
  }
};

// ============================================================================= CSV ==
namespace {

enum CsvKind { CSV_INT = 1, CSV_FLOAT = 2, CSV_BOOL = 4, CSV_TEXT = 8 };

bool all_digits(const char *p, const char *e) {
  if (p == e) return false;
  for (; p < e; p++)
    if (*p < '0' || *p > '9') return false;
  return true;
}
bool is_int(const std::string &s) {
  const char *p = s.data(), *e = p + s.size();
  if (p < e && *p == '-') p++;
  return all_digits(p, e);
}
bool is_float(const std::string &s) { // -?(\d*\.\d+|\d+\.\d*)([eE]-?\d+)? | -?\d+[eE]-?\d+
  const char *p = s.data(), *e = p + s.size();
  if (p < e && *p == '-') p++;
  const char *d0 = p;
  while (p < e && *p >= '0' && *p <= '9') p++;
  int int_digits = (int)(p - d0), frac_digits = 0;
  bool dot = false;
  if (p < e && *p == '.') {
    dot = true;
    p++;
    const char *f0 = p;
    while (p < e && *p >= '0' && *p <= '9') p++;
    frac_digits = (int)(p - f0);
  }
  if (int_digits + frac_digits == 0) return false;
  bool exp = false;
  if (p < e && (*p == 'e' || *p == 'E')) {
    p++;
    if (p < e && *p == '-') p++; // (arrow-csv 28's DECIMAL_RE has [eE]-?\d+: "1e+5" is text, not a float)
    if (!all_digits(p, e)) return false;
    p = e;
    exp = true;
  }
  return p == e && (dot || exp);
}
bool is_bool(const std::string &s) {
  if (s.size() == 4) return strncasecmp(s.c_str(), "true", 4) == 0;
  if (s.size() == 5) return strncasecmp(s.c_str(), "false", 5) == 0;
  return false;
}

// One CSV record -> fields (RFC 4180 quoting: "..." with "" as an escaped quote; the csv crate's
// defaults, which arrow-csv 28 uses).  Returns false at end of input.
bool read_record(std::istream &in, char delim, std::vector<std::string> &fields) {
  fields.clear();
  std::string cur;
  bool in_quotes = false, any = false, was_quoted = false;
  int ch;
  while ((ch = in.get()) != EOF) {
    any = true;
    char c = (char)ch;
    if (in_quotes) {
      if (c == '"') {
        if (in.peek() == '"') {
          cur.push_back('"');
          in.get();
        } else
          in_quotes = false;
      } else
        cur.push_back(c);
      continue;
    }
    if (c == '"' && cur.empty() && !was_quoted) {
      in_quotes = true;
      was_quoted = true;
    } else if (c == delim) {
      fields.push_back(cur);
      cur.clear();
      was_quoted = false;
    } else if (c == '\n') {
      if (!cur.empty() && cur.back() == '\r') cur.pop_back();
      if (fields.empty() && cur.empty() && !was_quoted) { // blank line: skipped by the csv crate
        any = false;
        continue;
      }
      fields.push_back(cur);
      return true;
    } else
      cur.push_back(c);
  }
  if (!any) return false;
  if (!cur.empty() && cur.back() == '\r') cur.pop_back();
  if (fields.empty() && cur.empty() && !was_quoted) return false;
  fields.push_back(cur);
  return true;
}

} // namespace

struct sqlrs_csv {
  Ctx *ctx = nullptr;
  std::ifstream file;
  char delimiter = ',';
  int64_t batch_size = 1024;
  std::vector<std::string> names;
  std::vector<int32_t> dtypes;
  std::vector<int> projection; // indices into the file's columns
  uint64_t remaining = ~0ull;  // records still allowed by the bounds
  uint64_t line = 0;           // for error messages
  bool has_header = true;
};

extern "C" {

// --------------------------------------------------------------------------- Project --
int sqlrs_project_create(sqlrs_ctx_t *ctx, int num_exprs, const sqlrs_expr_t *exprs, sqlrs_project_t **out) {
  return guard(ctx, [&] {
    auto p = std::unique_ptr<sqlrs_project>(new sqlrs_project());
    p->ctx = ctx;
    for (int i = 0; i < num_exprs; i++) p->exprs.push_back(expr_from_abi(&exprs[i]));
    *out = p.release();
  });
}
// [ref: project.rs:14-27] a bare InputRef shares the input column's buffers (the reference clones an Arc)
static DBatch project_batch(sqlrs_project *p, InBatch &ib, int out_mem);
int sqlrs_project_push(sqlrs_project_t *p, const sqlrs_batch_t *in, int out_mem, sqlrs_batch_t **out) {
  return guard(p->ctx, [&] {
    Ctx *ctx = p->ctx;
    SQ_HIP(hipSetDevice(ctx->device));
    InBatch ib(ctx, in);
    *out = emit_batch(ctx, project_batch(p, ib, out_mem), out_mem);
  });
}
// n input batches in one call (see sqlrs_filter_push_many): small HOST batches of fixed-width columns are staged, uploaded
// once, projected by ONE launch sequence — an expression's row i depends on row i alone (project.rs:15-27) — and cut
// back into one HOST batch per input batch; anything else (DEVICE memory, Utf8 / Boolean columns in or out, a constant
// projection) runs batch by batch.
static DBatch project_batch(sqlrs_project *p, InBatch &ib, int out_mem) {
  Ctx *ctx = p->ctx;
  auto colfn = [&](int i) -> const DCol & { return ib.col(i); };
  DBatch o;
  o.rows = ib.rows();
  for (const Expr &e : p->exprs) {
    DCol c = eval_expr(ctx, e, colfn, ib.rows(), true);
    c.length = ib.rows();
    // a column borrowed from a caller-built DEVICE batch must not outlive the call as a view
    const bool borrowed = (c.values && !c.own_values) || (c.validity && !c.own_validity) || (c.offsets && !c.own_offsets);
    if (borrowed && out_mem == SQLRS_MEM_DEVICE) c = copy_column(ctx, c);
    o.cols.push_back(std::move(c));
  }
  return o;
}
} // extern "C"

namespace sq {
// ---- one small HOST batch, one launch, no copy call (small_async.hpp): Project -------------------------------------------------
// An output column is either a bare InputRef — copied by the HOST into the slot's output area at the push (int32 / int64 /
// float64 / boolean / utf8: the rows do not move, project.rs:15-27) — or an expression over the batch's int32 / int64 /
// float64 columns, run per row as a postfix program (sa_eval_row: the evaluator's semantics; int32 / int64 / float64 / boolean
// results).  A wave owns 64 consecutive rows, so a validity bitmap (and a boolean result) leaves as one ballot word per wave.
constexpr int SP_PROGS = 6; // computed columns per projection (their programs are copied into LDS: 6 x 392 B)
struct SaOutCol {
  uint32_t out_off, out_voff, width;
  int32_t dtype;
  int32_t prog;       // -1: the host copied the column at the push
  uint32_t pre_nulls; // ... and counted its NULLs
};
struct SaProjectParams {
  SaLayout lay; // the INPUT columns the programs read (in_off = SA_NONE: not staged)
  int nout, nprog;
  uint32_t prog_off; // SaProgram[nprog] in the slot's input area
  SaOutCol oc[SA_MAX_COLS];
  const uint8_t *in;
  uint8_t *out;
  unsigned long long seq;
};
static_assert(sizeof(SaProjectParams) <= SA_PARAM_MAX, "parameter block too large");
__global__ __launch_bounds__(1024) void sa_project_kernel(SaGroup<SaProjectParams> grp) {
  const SaProjectParams &p = grp.p[blockIdx.x]; // (one workgroup per batch of the group)
  __shared__ __attribute__((aligned(8))) unsigned char s_prog_raw[SP_PROGS * sizeof(SaProgram)]; // (SaProgram has member initialisers)
  const SaProgram *s_prog = (const SaProgram *)s_prog_raw;
  __shared__ uint32_t s_nulls[SA_MAX_COLS], s_div0;
  {
    const uint32_t *src = (const uint32_t *)(p.in + p.prog_off);
    uint32_t *dst = (uint32_t *)s_prog_raw;
    const uint32_t nw = (uint32_t)p.nprog * (uint32_t)(sizeof(SaProgram) / 4);
    for (uint32_t i = threadIdx.x; i < nw; i += 1024) dst[i] = src[i];
  }
  if (threadIdx.x < SA_MAX_COLS) s_nulls[threadIdx.x] = (int)threadIdx.x < p.nout ? p.oc[threadIdx.x].pre_nulls : 0u;
  if (threadIdx.x == 0) s_div0 = 0;
  __syncthreads();
  const uint32_t rows = p.lay.rows;
  const int lane = lane_id();
  for (int c = 0; c < p.nout; c++) { // (uniform loop: the plan is a kernel argument)
    const SaOutCol &oc = p.oc[c];
    if (oc.prog < 0) continue;
    const SaProgram &pr = s_prog[oc.prog];
    for (int t = 0; t < 4; t++) {
      if ((uint32_t)t * 1024u >= rows) break; // (uniform)
      const uint32_t r = (uint32_t)t * 1024u + threadIdx.x, wbase = r & ~63u;
      bool valid = false, div0 = false;
      unsigned long long v = 0;
      if (r < rows) v = sa_eval_row(pr, p.lay, p.in, r, &valid, &div0);
      if (div0) s_div0 = 1u;
      if (!valid) v = 0; // (the slot of a NULL: zero)
      const uint64_t vm = __ballot(valid);
      if (oc.dtype == SQLRS_BOOLEAN) {
        const uint64_t bm = __ballot(valid && v != 0);
        if (lane == 0 && wbase < rows) ((uint64_t *)(p.out + oc.out_off))[r >> 6] = bm;
      } else if (r < rows) {
        if (oc.width == 8) ((unsigned long long *)(p.out + oc.out_off))[r] = v;
        else ((uint32_t *)(p.out + oc.out_off))[r] = (uint32_t)v;
      }
      if (lane == 0 && wbase < rows) {
        ((uint64_t *)(p.out + oc.out_voff))[r >> 6] = vm;
        const uint32_t nulls = min(64u, rows - wbase) - (uint32_t)__popcll(vm);
        if (nulls) atomicAdd(&s_nulls[c], nulls);
      }
    }
  }
  sa_publish((SaHeader *)p.out, p.seq, rows, s_nulls, p.nout, s_div0);
}
static void sa_project_launch(SaRing *r, Ctx *ctx) {
  SaGroup<SaProjectParams> g;
  for (int i = 0; i < r->pend_n; i++) std::memcpy(&g.p[i], r->pend_buf + (size_t)i * SA_PARAM_MAX, sizeof(SaProjectParams));
  sa_project_kernel<<<dim3((unsigned)r->pend_n), dim3(1024), 0, r->stream_of(r->pend_first_slot)>>>(g); // (reads nothing the ctx stream produces)
  SQ_HIP(hipGetLastError());
}
// plans the batch (which columns are copied, which computed), lays it out in the slot and copies what the host copies;
// false: not a batch for the fast path
static bool sa_project_stage(const sqlrs_project *pj, const sqlrs_batch_t *in, uint8_t *in_area, uint8_t *out_area, SaProjectParams *pp,
                             SaLayout *olay) {
  if (!in || in->num_rows < 0 || in->num_rows > (int64_t)SA_MAX_ROWS || in->num_columns <= 0 || in->num_columns > SA_MAX_COLS ||
      pj->exprs.empty() || pj->exprs.size() > (size_t)SA_MAX_COLS)
    return false;
  const uint32_t rows = (uint32_t)in->num_rows, vbytes = (rows + 7) / 8;
  auto fixed_w = [](int32_t d) { return d == SQLRS_INT32 ? 4u : (d == SQLRS_INT64 || d == SQLRS_FLOAT64) ? 8u : 0u; };
  for (int c = 0; c < in->num_columns; c++) {
    const sqlrs_column_t &col = in->columns[c];
    if (col.mem != SQLRS_MEM_HOST || col.length != in->num_rows) return false;
  }
  SaProgram progs[SP_PROGS];
  bool read[SA_MAX_COLS] = {};
  pp->nout = (int)pj->exprs.size();
  pp->nprog = 0;
  for (int e = 0; e < pp->nout; e++) {
    const Expr &ex = pj->exprs[(size_t)e];
    SaOutCol &oc = pp->oc[e];
    oc.pre_nulls = 0;
    if (ex.nodes.size() == 1 && ex.nodes[0].op == SQLRS_EXPR_INPUT_REF) {
      const int ci = ex.nodes[0].index;
      if (ci < 0 || ci >= in->num_columns) return false;
      const sqlrs_column_t &col = in->columns[ci];
      oc.prog = -1;
      oc.dtype = col.dtype;
      oc.width = fixed_w(col.dtype);
      if (col.dtype == SQLRS_UTF8) {
        if (!col.offsets || col.offsets[rows] < col.offsets[0] || (col.offsets[rows] > col.offsets[0] && !col.values)) return false;
      } else if (col.dtype == SQLRS_BOOLEAN) {
        if (rows && !col.values) return false;
      } else if (!oc.width || (rows && !col.values))
        return false;
      continue;
    }
    if (pp->nprog >= SP_PROGS || !sa_compile(ex, in, &progs[pp->nprog])) return false;
    const SaProgram &pr = progs[pp->nprog];
    oc.prog = pp->nprog++;
    oc.dtype = pr.result_dtype;
    oc.width = fixed_w(pr.result_dtype);
    if (!oc.width && pr.result_dtype != SQLRS_BOOLEAN) return false;
    for (int k = 0; k < pr.n; k++)
      if (pr.ins[k].op == SAO_COL) read[pr.ins[k].col] = true;
  }
  auto up64 = [](size_t x) { return (x + 63) & ~(size_t)63; };
  // input area: the columns a program reads | the programs
  size_t in_at = 0;
  SaLayout &lay = pp->lay;
  lay.ncols = in->num_columns;
  lay.rows = rows;
  for (int c = 0; c < in->num_columns; c++) {
    const sqlrs_column_t &col = in->columns[c];
    SaCol &d = lay.c[c];
    d.dtype = col.dtype;
    d.width = fixed_w(col.dtype);
    d.in_off = d.in_voff = d.out_off = d.out_voff = d.in_data = d.out_data = SA_NONE;
    d.data_base = 0;
    if (!read[c]) continue;
    if (!d.width || (rows && !col.values)) return false; // (sa_compile admits only the fixed-width types)
    d.in_off = (uint32_t)in_at;
    in_at = up64(in_at + (size_t)d.width * rows);
    if (col.validity && col.null_count != 0) {
      d.in_voff = (uint32_t)in_at;
      in_at = up64(in_at + vbytes + 8);
    }
  }
  pp->prog_off = (uint32_t)in_at;
  in_at = up64(in_at + sizeof(SaProgram) * (size_t)pp->nprog);
  // output area: header | per column values (+ the bytes of a Utf8 column) + validity, whole 64-bit words of bitmap
  size_t out_at = up64(sizeof(SaHeader));
  olay->ncols = pp->nout;
  olay->rows = rows;
  for (int e = 0; e < pp->nout; e++) {
    SaOutCol &oc = pp->oc[e];
    SaCol &d = olay->c[e];
    d.dtype = oc.dtype;
    d.width = oc.width;
    d.in_off = d.in_voff = d.in_data = d.out_data = SA_NONE;
    d.data_base = 0;
    const bool utf8 = oc.dtype == SQLRS_UTF8, boolean = oc.dtype == SQLRS_BOOLEAN;
    d.out_off = (uint32_t)out_at;
    out_at = up64(out_at + (utf8 ? 4 * ((size_t)rows + 1) : boolean ? (size_t)vbytes + 8 : (size_t)oc.width * rows));
    d.out_voff = (uint32_t)out_at;
    out_at = up64(out_at + vbytes + 8);
    if (utf8) {
      const sqlrs_column_t &col = in->columns[pj->exprs[(size_t)e].nodes[0].index];
      d.out_data = (uint32_t)out_at;
      out_at = up64(out_at + (size_t)(col.offsets[rows] - col.offsets[0]));
    }
    oc.out_off = d.out_off;
    oc.out_voff = d.out_voff;
  }
  if (in_at > SA_AREA || out_at > SA_AREA) return false;
  // the copies: what the programs read into the input area, bare columns straight into the output area
  for (int c = 0; c < in->num_columns; c++) {
    const SaCol &d = lay.c[c];
    if (d.in_off == SA_NONE) continue;
    const sqlrs_column_t &col = in->columns[c];
    if (rows) std::memcpy(in_area + d.in_off, col.values, (size_t)d.width * rows);
    if (d.in_voff != SA_NONE) std::memcpy(in_area + d.in_voff, col.validity, vbytes);
  }
  if (pp->nprog) std::memcpy(in_area + pp->prog_off, progs, sizeof(SaProgram) * (size_t)pp->nprog);
  for (int e = 0; e < pp->nout; e++) {
    SaOutCol &oc = pp->oc[e];
    if (oc.prog >= 0) continue;
    const sqlrs_column_t &col = in->columns[pj->exprs[(size_t)e].nodes[0].index];
    const SaCol &d = olay->c[e];
    if (oc.dtype == SQLRS_UTF8) {
      int32_t *oo = (int32_t *)(out_area + d.out_off);
      const int32_t base = col.offsets[0];
      for (uint32_t i = 0; i <= rows; i++) oo[i] = col.offsets[i] - base;
      if (oo[rows]) std::memcpy(out_area + d.out_data, (const uint8_t *)col.values + base, (size_t)oo[rows]);
    } else if (oc.dtype == SQLRS_BOOLEAN) {
      if (rows) std::memcpy(out_area + d.out_off, col.values, vbytes);
    } else if (rows)
      std::memcpy(out_area + d.out_off, col.values, (size_t)oc.width * rows);
    if (col.validity && col.null_count != 0) {
      std::memcpy(out_area + d.out_voff, col.validity, vbytes);
      int64_t nulls = col.null_count;
      if (nulls < 0) { // unknown: counted here
        nulls = 0;
        for (uint32_t i = 0; i < rows; i++) nulls += !((col.validity[i >> 3] >> (i & 7)) & 1);
      }
      oc.pre_nulls = (uint32_t)nulls;
    }
  }
  return true;
}
} // namespace sq

extern "C" {

// sqlrs_project_push without the wait (see sqlrs_filter_push_async): *ticket stands for the HOST batch
// sqlrs_project_push(p, in, SQLRS_MEM_HOST, ..) would return.  [ref: project.rs:15-27, polled one batch at a time]
int sqlrs_project_push_async(sqlrs_project_t *p, const sqlrs_batch_t *in, sqlrs_ticket_t **ticket) {
  if (ticket) *ticket = nullptr;
  return guard(p->ctx, [&] {
    Ctx *ctx = p->ctx;
    if (!ticket) fail(SQLRS_ERR_INTERNAL, "push_async: null ticket");
    SQ_HIP(hipSetDevice(ctx->device));
    auto t = std::unique_ptr<sqlrs_ticket>(new sqlrs_ticket());
    t->ctx = ctx;
    const char *off_e = hook("SQLRS_ASYNC_FAST"); // test hook, read per call: 0 = every batch through the synchronous operator
    if (!(off_e && off_e[0] == '0')) {
      SaRing *r = sa_ring(ctx);
      const int slot = r ? sa_take_slot(r) : -1;
      if (slot >= 0) {
        SaProjectParams pp;
        if (sa_project_stage(p, in, r->in_area(slot), r->out_area(slot), &pp, &t->lay)) {
          pp.in = r->in_area(slot);
          pp.out = r->out_area(slot);
          pp.seq = ++r->seq;
          sa_enqueue(ctx, r, p, sa_project_launch, pp, slot);
          t->slot = slot;
          t->seq = pp.seq;
          *ticket = t.release();
          return;
        }
        r->busy[slot] = false;
      }
    }
    sa_flush(ctx); // (tickets complete in issue order: what waits for a launch goes first)
    InBatch ib(ctx, in); // the synchronous operator, its batch parked in the ticket
    t->done = emit_batch(ctx, project_batch(p, ib, SQLRS_MEM_HOST), SQLRS_MEM_HOST);
    *ticket = t.release();
  });
}

int sqlrs_project_push_many(sqlrs_project_t *p, int n, const sqlrs_batch_t *const *in, int out_mem, sqlrs_batch_t **out) {
  return guard(p->ctx, [&] {
    Ctx *ctx = p->ctx;
    SQ_HIP(hipSetDevice(ctx->device));
    for (int i = 0; i < n; i++) out[i] = nullptr;
    if (n <= 0) return;
    if (!p->stage) {
      p->stage.reset(new HostStage());
      p->stage->ctx = ctx;
    }
    HostStage &st = *p->stage;
    bool stageable = out_mem == SQLRS_MEM_HOST && !st.has_schema && n > 1;
    int64_t total_rows = 0;
    for (int i = 0; i < n && stageable; i++) {
      stageable = st.accepts(in[i]) && in[i]->num_columns == in[0]->num_columns;
      for (int c = 0; c < in[i]->num_columns && stageable; c++) stageable = in[i]->columns[c].dtype == in[0]->columns[c].dtype;
      total_rows += in[i] ? in[i]->num_rows : 0;
    }
    auto one_by_one = [&] {
      int i = 0;
      try {
        for (; i < n; i++) {
          InBatch ib(ctx, in[i]);
          out[i] = emit_batch(ctx, project_batch(p, ib, out_mem), out_mem);
        }
      } catch (...) {
        for (int k = 0; k < i; k++) {
          sqlrs_batch_release(out[k]);
          out[k] = nullptr;
        }
        throw;
      }
    };
    if (!stageable || total_rows == 0 || total_rows > (1ll << 30)) {
      one_by_one();
      return;
    }
    std::vector<int64_t> bounds((size_t)n + 1, 0);
    try {
      for (int i = 0; i < n; i++) {
        st.append(in[i]);
        bounds[(size_t)i + 1] = bounds[(size_t)i] + in[i]->num_rows;
      }
    } catch (...) {
      p->stage.reset(); // (a half-staged call must not leave its schema and rows behind)
      throw;
    }
    sqlrs_batch_t *dev = st.take(); // one upload per column
    struct Rel {
      sqlrs_batch_t *b;
      ~Rel() { if (b) sqlrs_batch_release(b); }
    } rel{dev};
    DBatch o;
    {
      InBatch ib(ctx, dev);
      o = project_batch(p, ib, SQLRS_MEM_DEVICE);
    }
    if (!all_fixed_width(o)) { // Utf8 / Boolean results, constants (stride 0): the ordinary path, batch by batch
      one_by_one();
      return;
    }
    split_rows_to_host(ctx, o, bounds, &p->pin_out, &p->pin_cap, n, out); // one copy per column, one slice per input batch
  });
}
void sqlrs_project_destroy(sqlrs_project_t *p) {
  if (p) sa_flush(p->ctx); // (a group of its batches may still wait for its launch; their tickets stay valid)
  delete p;
}

// ------------------------------------------------------------------------- CrossJoin --
int sqlrs_cross_join_create(sqlrs_ctx_t *ctx, sqlrs_cross_join_t **out) {
  return guard(ctx, [&] {
    auto j = std::unique_ptr<sqlrs_cross_join>(new sqlrs_cross_join());
    j->ctx = ctx;
    *out = j.release();
  });
}
int sqlrs_cross_join_build_push(sqlrs_cross_join_t *j, const sqlrs_batch_t *left) {
  return guard(j->ctx, [&] {
    Ctx *ctx = j->ctx;
    SQ_HIP(hipSetDevice(ctx->device));
    if (j->concatenated) fail(SQLRS_ERR_INTERNAL, "cross join: build_push after the first probe batch");
    InBatch ib(ctx, left);
    j->left.push_back(ib.materialize(true));
  });
}
// [ref: cross_join.rs:38-56] the (left row, right batch) outputs of one right batch for left rows [left_begin, left_begin +
// left_rows), left row major; left_rows < 0 = all of them
static void cross_join_probe(sqlrs_cross_join *j, const sqlrs_batch_t *right, int64_t left_begin, int64_t left_rows, int out_mem,
                             sqlrs_batch_t **out) {
  Ctx *ctx = j->ctx;
  SQ_HIP(hipSetDevice(ctx->device));
  *out = nullptr;
  if (j->left.empty()) return; // cross_join.rs:32-34
  if (!j->concatenated) {
    j->left_all.rows = 0;
    for (const DBatch &b : j->left) j->left_all.rows += b.rows;
    for (size_t c = 0; c < j->left[0].cols.size(); c++) {
      std::vector<const DCol *> parts;
      for (const DBatch &b : j->left) {
        if (b.cols.size() != j->left[0].cols.size()) fail(SQLRS_ERR_ARROW, "cross join: left batches of different schemas");
        parts.push_back(&b.cols[c]);
      }
      j->left_all.cols.push_back(concat_columns(ctx, parts));
    }
    j->concatenated = true;
  }
  InBatch ib(ctx, right);
  const int64_t Lall = j->left_all.rows, R = ib.rows();
  if (left_rows < 0) {
    left_begin = 0;
    left_rows = Lall;
  }
  if (left_begin < 0 || left_begin + left_rows > Lall) fail(SQLRS_ERR_INTERNAL, "cross join: left row range outside the build side");
  if (Lall > 0xffffffffll) fail(SQLRS_ERR_INTERNAL, "cross join: more than 2^32 left rows");
  const int64_t L = left_rows;
  if (L == 0) return;
  // one library batch holds < 2^31 rows (the reference emits one batch per left row and has no such limit: callers with more
  // take the left rows in ranges, sqlrs_cross_join_probe_push_range)
  if (R > 0 && L > 0x7fffffffll / R) fail(SQLRS_ERR_INTERNAL, "cross join: more than 2^31 output rows in one call (use sqlrs_cross_join_probe_push_range)");
  const int64_t n = L * R;
  DBatch o;
  o.rows = n;
  BufP li = ctx->alloc(4 * (size_t)std::max<int64_t>(n, 1)), ri = ctx->alloc(4 * (size_t)std::max<int64_t>(n, 1));
  if (n) {
    cross_index_kernel<<<dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream>>>(li->as<uint32_t>(), ri->as<uint32_t>(), n, (uint32_t)R,
                                                                                      (uint32_t)left_begin);
    SQ_HIP(hipGetLastError());
  }
  for (const DCol &c : j->left_all.cols) o.cols.push_back(gather_column(ctx, c, li->p, false, nullptr, n));
  for (int c = 0; c < ib.num_columns(); c++) o.cols.push_back(gather_column(ctx, ib.col(c), ri->p, false, nullptr, n));
  *out = emit_batch(ctx, std::move(o), out_mem);
}
int sqlrs_cross_join_probe_push(sqlrs_cross_join_t *j, const sqlrs_batch_t *right, int out_mem, sqlrs_batch_t **out) {
  return guard(j->ctx, [&] { cross_join_probe(j, right, 0, -1, out_mem, out); });
}
int sqlrs_cross_join_probe_push_range(sqlrs_cross_join_t *j, const sqlrs_batch_t *right, int64_t left_begin, int64_t left_rows, int out_mem,
                                      sqlrs_batch_t **out) {
  return guard(j->ctx, [&] {
    if (left_rows < 0) fail(SQLRS_ERR_INTERNAL, "cross join: negative row count");
    cross_join_probe(j, right, left_begin, left_rows, out_mem, out);
  });
}
int64_t sqlrs_cross_join_left_rows(const sqlrs_cross_join_t *j) {
  int64_t n = 0;
  for (const DBatch &b : j->left) n += b.rows;
  return n;
}
void sqlrs_cross_join_destroy(sqlrs_cross_join_t *j) { delete j; }

// ----------------------------------------------------------------------------- Limit --
int sqlrs_limit_create(sqlrs_ctx_t *ctx, int has_limit, int64_t limit, int has_offset, int64_t offset, sqlrs_limit_t **out) {
  return guard(ctx, [&] {
    if ((has_limit && limit < 0) || (has_offset && offset < 0)) fail(SQLRS_ERR_INTERNAL, "negative limit / offset");
    auto l = std::unique_ptr<sqlrs_limit>(new sqlrs_limit());
    l->ctx = ctx;
    l->has_limit = has_limit != 0;
    l->limit = (uint64_t)limit;
    l->offset_val = has_offset ? (uint64_t)offset : 0; // limit.rs:21-27
    if (l->has_limit && l->limit == 0) l->done = true;  // limit.rs:29-31
    *out = l.release();
  });
}
// one iteration of the for_await loop [ref: limit.rs:35-79]; *out = NULL when the batch yields nothing,
// *done = 1 once no later batch can contribute (the reference `break`s / returned early)
int sqlrs_limit_push(sqlrs_limit_t *l, const sqlrs_batch_t *in, int out_mem, sqlrs_batch_t **out, int *done) {
  return guard(l->ctx, [&] {
    Ctx *ctx = l->ctx;
    SQ_HIP(hipSetDevice(ctx->device));
    *out = nullptr;
    if (l->done) {
      if (done) *done = 1;
      return;
    }
    InBatch ib(ctx, in);
    const uint64_t cardinality = (uint64_t)ib.rows();
    const uint64_t limit_val = l->has_limit ? l->limit : cardinality;
    const uint64_t start = std::max(l->returned_count, l->offset_val) - l->returned_count;
    const uint64_t total_end = l->offset_val + limit_val;
    const uint64_t current_batch_end = l->returned_count + cardinality;
    const uint64_t end = std::min(total_end, current_batch_end) - l->returned_count;
    l->returned_count += cardinality;
    if (start < end) {
      DBatch o;
      o.rows = (int64_t)(end - start);
      if (start == 0 && end == cardinality) {
        o = ib.materialize(out_mem == SQLRS_MEM_DEVICE); // the batch itself (limit.rs:65-66)
      } else { // batch.slice(start, length): device bitmaps are word aligned, so a slice is a gather of [start, end)
        const int64_t m = (int64_t)(end - start);
        BufP idx = ctx->alloc(4 * (size_t)m);
        iota_offset_u32_kernel<<<dim3((unsigned)ceil_div(m, 256)), dim3(256), 0, ctx->stream>>>(idx->as<uint32_t>(), m, (uint32_t)start);
        SQ_HIP(hipGetLastError());
        for (int c = 0; c < ib.num_columns(); c++) o.cols.push_back(gather_column(ctx, ib.col(c), idx->p, false, nullptr, m));
      }
      *out = emit_batch(ctx, std::move(o), out_mem);
      if (l->returned_count >= l->offset_val + limit_val) l->done = true; // limit.rs:76-78
    }
    if (done) *done = l->done ? 1 : 0;
  });
}
void sqlrs_limit_destroy(sqlrs_limit_t *l) { delete l; }

// ------------------------------------------------------------------------- SimpleAgg --
int sqlrs_simple_agg_create(sqlrs_ctx_t *ctx, int num_aggs, const sqlrs_agg_func_t *aggs, sqlrs_simple_agg_t **out) {
  return guard(ctx, [&] {
    auto a = std::unique_ptr<sqlrs_simple_agg>(new sqlrs_simple_agg());
    a->ctx = ctx;
    sqlrs_expr_node_t key;
    std::memset(&key, 0, sizeof(key));
    key.op = SQLRS_EXPR_CONSTANT;
    key.dtype = SQLRS_INT64;
    sqlrs_expr_t gb{&key, 1, 0};
    int st = sqlrs_hash_agg_create(ctx, 1, &gb, num_aggs, aggs, &a->inner);
    if (st != SQLRS_OK) fail(st, ctx->last_error);
    for (int i = 0; i < num_aggs; i++) {
      a->funcs.push_back(aggs[i].func);
      a->return_dtypes.push_back(aggs[i].return_dtype);
    }
    *out = a.release();
  });
}
int sqlrs_simple_agg_push(sqlrs_simple_agg_t *a, const sqlrs_batch_t *in) {
  a->saw_batch = true;
  return sqlrs_hash_agg_push(a->inner, in);
}
// [ref: simple_agg.rs:58-64] exactly one row; no input rows at all: COUNT = 0, everything else NULL
int sqlrs_simple_agg_finish(sqlrs_simple_agg_t *a, int out_mem, sqlrs_batch_t **out) {
  sqlrs_batch_t *g = nullptr;
  if (!a->saw_batch) return guard(a->ctx, [&] { fail(SQLRS_ERR_INTERNAL, "simple agg finished without any input batch"); });
  int st = sqlrs_hash_agg_finish(a->inner, SQLRS_MEM_DEVICE, &g);
  if (st != SQLRS_OK) return st;
  st = guard(a->ctx, [&] {
    Ctx *ctx = a->ctx;
    DBatch o;
    o.rows = 1;
    if (g->num_rows == 1) {
      InBatch ib(ctx, g); // library-owned: columns are shared, not copied
      for (int c = 1; c < ib.num_columns(); c++) o.cols.push_back(ib.col(c));
    } else if (g->num_rows == 0) {
      for (size_t i = 0; i < a->funcs.size(); i++) {
        if (a->funcs[i] == SQLRS_AGG_COUNT) {
          DCol z;
          z.dtype = SQLRS_INT64;
          z.length = 1;
          z.own_values = ctx->alloc_zero(8);
          z.values = z.own_values->p;
          o.cols.push_back(std::move(z));
        } else
          o.cols.push_back(make_null_column(ctx, a->return_dtypes[i], 1));
      }
    } else
      fail(SQLRS_ERR_INTERNAL, "simple agg produced more than one group");
    *out = emit_batch(ctx, std::move(o), out_mem);
  });
  sqlrs_batch_release(g);
  return st;
}
void sqlrs_simple_agg_destroy(sqlrs_simple_agg_t *a) { delete a; }

// ------------------------------------------------------------- record_batch_to_string --
// [ref: src/util/mod.rs:53-80]
int sqlrs_batch_to_string(sqlrs_ctx_t *ctx, const sqlrs_batch_t *in, char **out) {
  return guard(ctx, [&] {
    SQ_HIP(hipSetDevice(ctx->device));
    sqlrs_batch_t *host = nullptr;
    const sqlrs_batch_t *b = in;
    bool device = false;
    for (int c = 0; c < in->num_columns; c++) device |= in->columns[c].mem == SQLRS_MEM_DEVICE;
    if (device) {
      InBatch ib(ctx, in);
      host = emit_batch(ctx, ib.materialize(false), SQLRS_MEM_HOST);
      b = host;
    }
    std::string s;
    for (int64_t row = 0; row < b->num_rows; row++) {
      for (int c = 0; c < b->num_columns; c++) {
        if (c) s.push_back(' ');
        const sqlrs_column_t &col = b->columns[c];
        if (col.validity && col.null_count != 0 && !((col.validity[row >> 3] >> (row & 7)) & 1)) {
          s += "NULL";
          continue;
        }
        switch (col.dtype) {
        case SQLRS_UTF8: {
          const int32_t lo = col.offsets[row], hi = col.offsets[row + 1];
          if (lo == hi) s += "(empty)";
          else s.append((const char *)col.values + lo, (size_t)(hi - lo));
          break;
        }
        case SQLRS_BOOLEAN:
          s += ((((const uint8_t *)col.values)[row >> 3] >> (row & 7)) & 1) ? "true" : "false";
          break;
        case SQLRS_FLOAT64: { // Rust's Display: shortest digits that round-trip, positional notation
          const double v = ((const double *)col.values)[row];
          if (std::isnan(v)) s += "NaN";
          else if (std::isinf(v)) s += v < 0 ? "-inf" : "inf";
          else {
            char buf[400];
            auto r = std::to_chars(buf, buf + sizeof(buf), v, std::chars_format::fixed);
            s.append(buf, r.ptr);
          }
          break;
        }
        case SQLRS_INT32: s += std::to_string(((const int32_t *)col.values)[row]); break;
        case SQLRS_UINT32: s += std::to_string(((const uint32_t *)col.values)[row]); break;
        case SQLRS_INT64: s += std::to_string(((const int64_t *)col.values)[row]); break;
        case SQLRS_UINT64: s += std::to_string(((const uint64_t *)col.values)[row]); break;
        default:
          if (host) sqlrs_batch_release(host);
          fail(SQLRS_ERR_INTERNAL, "unsupported column type in record_batch_to_string");
        }
      }
      s.push_back('\n');
    }
    if (host) sqlrs_batch_release(host);
    char *p = (char *)std::malloc(s.size() + 1);
    if (!p) fail(SQLRS_ERR_INTERNAL, "allocation failed");
    std::memcpy(p, s.c_str(), s.size() + 1);
    *out = p;
  });
}
void sqlrs_string_free(char *s) { std::free(s); }

// ------------------------------------------------------------------------------- CSV --
// [ref: src/storage/csv.rs:92-106 CsvConfig defaults, :124-133 infer_arrow_schema, :190-241 reader]
int sqlrs_csv_open(sqlrs_ctx_t *ctx, const char *path, int has_header, char delimiter, int64_t batch_size,
                   int64_t infer_max_records, sqlrs_csv_t **out) {
  return guard(ctx, [&] {
    auto r = std::unique_ptr<sqlrs_csv>(new sqlrs_csv());
    r->ctx = ctx;
    r->delimiter = delimiter ? delimiter : ',';
    r->has_header = has_header != 0;
    r->batch_size = batch_size > 0 ? batch_size : 1024;
    // pass 1: header + type inference over the first records (arrow-csv infer_reader_schema)
    std::ifstream probe(path, std::ios::binary);
    if (!probe) fail(SQLRS_ERR_STORAGE, std::string("cannot open ") + path);
    std::vector<std::string> f;
    if (has_header) {
      if (!read_record(probe, r->delimiter, f)) fail(SQLRS_ERR_STORAGE, "empty csv file");
      r->names = f;
    }
    std::vector<int> kinds;
    int64_t seen = 0;
    while ((infer_max_records <= 0 || seen < infer_max_records) && read_record(probe, r->delimiter, f)) {
      if (r->names.empty())
        for (size_t i = 0; i < f.size(); i++) r->names.push_back("column_" + std::to_string(i + 1));
      kinds.resize(r->names.size(), 0);
      for (size_t i = 0; i < f.size() && i < kinds.size(); i++) {
        if (f[i].empty()) continue; // a missing value says nothing about the type
        kinds[i] |= is_bool(f[i]) ? CSV_BOOL : is_int(f[i]) ? CSV_INT : is_float(f[i]) ? CSV_FLOAT : CSV_TEXT;
      }
      seen++;
    }
    kinds.resize(r->names.size(), 0);
    for (int k : kinds) {
      int32_t dt = SQLRS_UTF8;
      if (k == CSV_INT) dt = SQLRS_INT64;
      else if (k == CSV_FLOAT || k == (CSV_INT | CSV_FLOAT)) dt = SQLRS_FLOAT64;
      else if (k == CSV_BOOL) dt = SQLRS_BOOLEAN;
      r->dtypes.push_back(dt);
    }
    for (size_t i = 0; i < r->names.size(); i++) r->projection.push_back((int)i);
    // pass 2 starts behind the header
    r->file.open(path, std::ios::binary);
    if (!r->file) fail(SQLRS_ERR_STORAGE, std::string("cannot open ") + path);
    if (has_header) read_record(r->file, r->delimiter, f);
    *out = r.release();
  });
}
int sqlrs_csv_num_columns(const sqlrs_csv_t *r) { return (int)r->names.size(); }
const char *sqlrs_csv_column_name(const sqlrs_csv_t *r, int i) { return r->names[(size_t)i].c_str(); }
int sqlrs_csv_column_dtype(const sqlrs_csv_t *r, int i) { return r->dtypes[(size_t)i]; }
// Bounds of the scan (offset, limit) over the data records [ref: csv.rs:207-215]; limit < 0 = unbounded
int sqlrs_csv_set_bounds(sqlrs_csv_t *r, int64_t offset, int64_t limit) {
  return guard(r->ctx, [&] {
    std::vector<std::string> f;
    for (int64_t i = 0; i < offset; i++)
      if (!read_record(r->file, r->delimiter, f)) break;
    // csv.rs:216-223 turns (offset, limit) into arrow-csv's line bounds (offset, offset + limit + 1), and arrow-csv 28
    // starts its line counter at offset + 1 only when the file has a header: a headerless scan yields limit + 1
    // records (the LimitExecutor above trims the batch either way)
    r->remaining = limit < 0 ? ~0ull : (uint64_t)limit + (r->has_header ? 0 : 1);
  });
}
int sqlrs_csv_set_projection(sqlrs_csv_t *r, int num_columns, const int32_t *columns) {
  return guard(r->ctx, [&] {
    r->projection.clear();
    for (int i = 0; i < num_columns; i++) {
      if (columns[i] < 0 || (size_t)columns[i] >= r->names.size()) fail(SQLRS_ERR_INTERNAL, "projection out of range");
      r->projection.push_back(columns[i]);
    }
  });
}
// next batch of <= batch_size records, *out = NULL at the end of the scan [ref: csv.rs:236-241]
int sqlrs_csv_next_batch(sqlrs_csv_t *r, int out_mem, sqlrs_batch_t **out) {
  return guard(r->ctx, [&] {
    Ctx *ctx = r->ctx;
    *out = nullptr;
    const size_t nc = r->projection.size();
    std::vector<std::vector<uint8_t>> values(nc), valid(nc);
    std::vector<std::vector<int32_t>> offsets(nc);
    std::vector<int64_t> nulls(nc, 0);
    for (size_t c = 0; c < nc; c++)
      if (r->dtypes[(size_t)r->projection[c]] == SQLRS_UTF8) offsets[c].push_back(0);
    std::vector<std::string> f;
    int64_t rows = 0;
    while (rows < r->batch_size && r->remaining > 0 && read_record(r->file, r->delimiter, f)) {
      r->line++;
      if (r->remaining != ~0ull) r->remaining--;
      // a record with another number of fields than the schema is an error of the csv crate (UnequalLengths), which
      // arrow-csv surfaces as an ArrowError — not a row padded with NULLs
      if (f.size() != r->names.size())
        fail(SQLRS_ERR_ARROW, "Error parsing line " + std::to_string(r->line) + ": found record with " + std::to_string(f.size()) +
                                  " fields, but the previous record has " + std::to_string(r->names.size()) + " fields");
      for (size_t c = 0; c < nc; c++) {
        const int src = r->projection[c];
        const std::string &s = (size_t)src < f.size() ? f[(size_t)src] : std::string();
        const int32_t dt = r->dtypes[(size_t)src];
        bool ok = true;
        if (dt == SQLRS_UTF8) { // an empty field is the empty string, not NULL
          values[c].insert(values[c].end(), s.begin(), s.end());
          offsets[c].push_back((int32_t)values[c].size());
        } else if (s.empty()) { // missing value of a typed column -> NULL
          ok = false;
          if (dt == SQLRS_BOOLEAN) {
            if ((rows & 7) == 0) values[c].push_back(0);
          } else
            values[c].resize(values[c].size() + 8, 0);
        } else if (dt == SQLRS_INT64) {
          int64_t v = 0;
          auto pr = std::from_chars(s.data(), s.data() + s.size(), v);
          if (pr.ec != std::errc() || pr.ptr != s.data() + s.size())
            fail(SQLRS_ERR_ARROW, "Error while parsing value " + s + " for column " + std::to_string(src) + " at line " + std::to_string(r->line));
          const uint8_t *p = (const uint8_t *)&v;
          values[c].insert(values[c].end(), p, p + 8);
        } else if (dt == SQLRS_FLOAT64) {
          double v = 0;
          auto pr = std::from_chars(s.data(), s.data() + s.size(), v);
          if (pr.ec != std::errc() || pr.ptr != s.data() + s.size())
            fail(SQLRS_ERR_ARROW, "Error while parsing value " + s + " for column " + std::to_string(src) + " at line " + std::to_string(r->line));
          const uint8_t *p = (const uint8_t *)&v;
          values[c].insert(values[c].end(), p, p + 8);
        } else { // BOOLEAN
          if (!is_bool(s))
            fail(SQLRS_ERR_ARROW, "Error while parsing value " + s + " for column " + std::to_string(src) + " at line " + std::to_string(r->line));
          if ((rows & 7) == 0) values[c].push_back(0);
          if (s.size() == 4) values[c].back() |= (uint8_t)(1u << (rows & 7));
        }
        if ((rows & 7) == 0) valid[c].push_back(0);
        if (ok) valid[c].back() |= (uint8_t)(1u << (rows & 7));
        else nulls[c]++;
      }
      rows++;
    }
    if (rows == 0) return; // end of the scan
    std::vector<sqlrs_column_t> cols(nc);
    auto take_bytes = [&](const void *src, size_t bytes) -> void * {
      void *p = std::malloc(bytes + 64);
      if (!p) fail(SQLRS_ERR_INTERNAL, "host allocation failed");
      std::memset(p, 0, bytes + 64);
      if (bytes) std::memcpy(p, src, bytes);
      return p;
    };
    for (size_t c = 0; c < nc; c++) {
      sqlrs_column_t &d = cols[c];
      d.dtype = r->dtypes[(size_t)r->projection[c]];
      d.mem = SQLRS_MEM_HOST;
      d.length = rows;
      d.null_count = nulls[c];
      d.values = take_bytes(values[c].data(), values[c].size());
      d.validity = nulls[c] ? (const uint8_t *)take_bytes(valid[c].data(), valid[c].size()) : nullptr;
      d.offsets = d.dtype == SQLRS_UTF8 ? (const int32_t *)take_bytes(offsets[c].data(), 4 * offsets[c].size()) : nullptr;
    }
    sqlrs_batch_t *host = emit_host_columns(ctx, std::move(cols), rows);
    if (out_mem == SQLRS_MEM_HOST) {
      *out = host;
      return;
    }
    // straight into HBM: one upload per column, the host copy is dropped
    SQ_HIP(hipSetDevice(ctx->device));
    sqlrs_batch_t *dev = nullptr;
    try {
      InBatch ib(ctx, host);
      dev = emit_batch(ctx, ib.materialize(true), SQLRS_MEM_DEVICE);
    } catch (...) {
      sqlrs_batch_release(host);
      throw;
    }
    ctx->sync(); // the uploads read the host blocks
    sqlrs_batch_release(host);
    *out = dev;
  });
}
void sqlrs_csv_close(sqlrs_csv_t *r) { delete r; }

} // extern "C"
