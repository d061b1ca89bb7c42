// small_async.hpp — the reference's calling shape without a stream synchronisation per call (round 6, review r05 #6).
//
// The reference polls ONE 1024-row batch at a time (storage/csv.rs:105, filter.rs:15-24, hash_join.rs:284-291).  A
// synchronous `push` of such a batch is a chain of round trips — upload, two to four launches, the count, the download, a
// stream synchronisation: ~47 us for 8 KB of rows, 21 Mrows/s (Filter) / 11 Mrows/s (HashJoin probe).  `push_async` hands
// back a TICKET instead of a batch and `sqlrs_batch_wait(ticket)` the batch; a caller that waits one (or a few) batches
// behind never blocks on the device.  The fast path is ONE launch per batch and no copy call at all:
//   * the batch's columns are copied (memcpy, 8 KB per column) into a slot of a pinned, device-mapped ring;
//   * one 1024-thread workgroup reads them over PCIe, evaluates the predicate — a conjunction of `column OP constant` terms directly, anything else over the
//     int32 / int64 / float64 columns as a postfix program (SaProgram, sa_eval_row: the evaluator's semantics row by row) — or looks the keys up in the
//     join's table (HashJoin, unique build keys), compacts the kept rows with ballots, gathers the build columns, and
//     writes the output columns + validity bitmaps + a header {sequence number, rows, NULL counts} back into the slot;
//   * `wait` polls the header's sequence number (system-scope release store behind `__threadfence_system`), falling
//     back to a stream synchronisation when it does not show up, and copies the rows into an ordinary HOST batch.
// Anything the fast path does not take — predicates that read a Boolean / Utf8 column or need more than 24 nodes, Utf8
// columns around a join, more than 4096 rows, DEVICE input,
// duplicate build keys, join filters, outer joins — runs the synchronous operator inside push_async and parks the finished
// batch in the ticket: same results, same one-output-per-input rule, no speed-up.
#pragma once

#include <cstring>

#include "common.hpp"

namespace sq {

constexpr int SA_SLOTS = 32, SA_MAX_COLS = 12;
constexpr uint32_t SA_MAX_ROWS = 4096, SA_NONE = 0xffffffffu;
constexpr size_t SA_AREA = 512 * 1024; // bytes of a slot's input area and of its output area

struct SaHeader { // at the start of a slot's output area
  unsigned long long seq; // written LAST by the kernel: the ticket's sequence number
  uint32_t count, pad;
  uint32_t nulls[SA_MAX_COLS];
};
struct SaCol {
  uint32_t in_off, in_voff;   // values / validity bitmap in the input area (in_voff = SA_NONE: no NULLs); build columns: unused
  uint32_t out_off, out_voff; // values / validity bitmap in the output area (always reserved)
  uint32_t width;             // 4 or 8; Utf8: 4 (in_off / out_off are the int32 offsets, rows + 1 of them)
  int32_t dtype;
  // Utf8 (Filter only): the bytes [offsets[0], offsets[rows]) of the input column sit at in_data; the kept rows' bytes go to out_data
  uint32_t in_data, out_data, data_base; // data_base = offsets[0] of the input column
};
struct SaLayout {
  int ncols = 0;
  uint32_t rows = 0;
  SaCol c[SA_MAX_COLS];
};

constexpr int SA_STREAMS = 4, SA_GROUP_MAX = 4;
constexpr size_t SA_PARAM_MAX = 1024; // bytes of one batch's kernel parameters
struct SaRing {
  uint8_t *pin = nullptr; // SA_SLOTS x (input area | output area), pinned + device mapped
  bool busy[SA_SLOTS] = {};
  int next = 0;
  unsigned long long seq = 0;
  // The one-workgroup kernels of consecutive batches are independent of each other; on ONE stream they run back to back
  // (launch latency + a PCIe read round trip + the write-back: ~9 us per batch, 111 Mrows/s measured), on SA_STREAMS side
  // streams they overlap.  A kernel that reads operator state built on the ctx stream (the join's table) is ordered behind
  // it with one event per operator (sa_order_after_ctx), and an operator that is destroyed drains the side streams first.
  hipStream_t side[SA_STREAMS] = {};
  hipEvent_t order_ev = nullptr;
  bool dirty = false; // a kernel was queued on a side stream since the last drain
  // Grouped launches: the parameters of up to `group` consecutive batches of ONE operator wait here and leave with ONE launch
  // (grid = batches, a workgroup each) — the launch call is the largest part of the host's ~6 us per batch.  Flushed when the
  // group is full, when another operator (or the synchronous path) comes, and by whoever waits for a ticket of the group.
  unsigned char pend_buf[SA_GROUP_MAX * SA_PARAM_MAX];
  int pend_n = 0, pend_first_slot = 0, group = SA_GROUP_MAX;
  const void *pend_owner = nullptr;
  void (*pend_launch)(SaRing *, Ctx *) = nullptr;
  unsigned long long launched_seq = 0; // the kernels of all tickets up to this sequence number are queued
  unsigned long long pend_last_seq = 0; // the newest ticket of the pending group (NOT `seq`: the ticket whose push flushes another
                                        // operator's group has its number already and joins the NEXT group)
  ~SaRing() {
    for (hipStream_t s : side)
      if (s) {
        (void)hipStreamSynchronize(s);
        (void)hipStreamDestroy(s);
      }
    if (order_ev) (void)hipEventDestroy(order_ev);
    if (pin) (void)hipHostFree(pin);
  }
  hipStream_t stream_of(int slot) const { return side[slot % SA_STREAMS]; }
  uint8_t *in_area(int s) const { return pin + (size_t)s * 2 * SA_AREA; }
  uint8_t *out_area(int s) const { return pin + (size_t)s * 2 * SA_AREA + SA_AREA; }
};
SaRing *sa_ring(Ctx *ctx);  // the ctx's ring, created on first use
int sa_take_slot(SaRing *r); // -1: every slot has a ticket outstanding
void sa_order_after_ctx(Ctx *ctx, SaRing *r); // every side stream waits for what the ctx stream holds now
void sa_drain(Ctx *ctx);                      // flushes, then waits for the side streams (operator teardown); no-op without a ring
void sa_flush(Ctx *ctx);                      // launches the pending group, if any
template <class P> struct SaGroup { P p[SA_GROUP_MAX]; };
// appends one batch's parameters to the pending group of `owner` (launching what another operator left there first)
template <class P> inline void sa_enqueue(Ctx *ctx, SaRing *r, const void *owner, void (*launch)(SaRing *, Ctx *), const P &p, int slot) {
  static_assert(sizeof(P) <= SA_PARAM_MAX, "parameter block too large");
  if (r->pend_n && (r->pend_owner != owner || r->pend_launch != launch)) sa_flush(ctx);
  if (!r->pend_n) r->pend_first_slot = slot;
  std::memcpy(r->pend_buf + (size_t)r->pend_n * SA_PARAM_MAX, &p, sizeof(P));
  r->pend_owner = owner;
  r->pend_launch = launch;
  r->pend_last_seq = p.seq;
  r->pend_n++;
  ctx->async_fast_batches++;
  if (r->pend_n >= r->group) sa_flush(ctx);
}

// Lays `in` (HOST columns of int32 / int64 / float64 — and Utf8 when `allow_utf8` — <= SA_MAX_ROWS rows) out in `area` and describes it in `lay`;
// `first_out_col` output columns are reserved in front of the batch's own (the join's build columns).  false = not a batch
// for the fast path (nothing written that matters).
bool sa_stage_input(const sqlrs_batch_t *in, uint8_t *area, SaLayout *lay, int first_out_col, const int32_t *front_dtypes,
                    bool allow_utf8 = false);

// ---- a postfix program over the batch's fixed-width columns, evaluated per row INSIDE the one-launch kernels -----------------
// BoundExpr::eval_column (evaluator.rs:13-28, array_compute.rs:70-90) restated for one row: the same arithmetic (integers wrap,
// x / 0 on a valid row is the Arrow error), the same comparisons (doubles in IEEE total order), Kleene AND / OR, the same casts
// (out of range -> NULL), NULL = an operand was NULL — expr.hip does this column by column with one launch per node.
constexpr int SA_PROG_MAX = 24, SA_STACK_MAX = 8;
enum SaOp : uint8_t { SAO_COL = 0, SAO_CONST, SAO_CAST, SAO_ADD, SAO_SUB, SAO_MUL, SAO_DIV, SAO_GT, SAO_LT, SAO_GE, SAO_LE, SAO_EQ, SAO_NE, SAO_AND, SAO_OR };
struct SaInstr {
  uint8_t op, dtype, from, is_null; // dtype: operand type of an arithmetic / comparison, target of a cast, type of a constant / column
  uint32_t col;                     // SAO_COL: column of the batch
  unsigned long long imm;           // SAO_CONST: the value's bits (int32 sign-extended, bool 0 / 1)
};
struct SaProgram {
  int n = 0;
  int32_t result_dtype = 0;
  SaInstr ins[SA_PROG_MAX];
};
// Expr -> program over the columns of `in`; false = not expressible here (a Utf8 / Boolean column or constant, mixed operand
// types, an unsupported cast, too long): the synchronous evaluator takes the batch and raises whatever error there is to raise
bool sa_compile(const Expr &e, const sqlrs_batch_t *in, SaProgram *out);

} // namespace sq

// what push_async returns and sqlrs_batch_wait consumes
struct sqlrs_ticket {
  sq::Ctx *ctx = nullptr;
  int slot = -1; // -1: the slow path ran, `done` is the batch (may be NULL: an operator that emits nothing)
  unsigned long long seq = 0;
  sqlrs_batch_t *done = nullptr;
  sq::SaLayout lay; // (output side: offsets, widths, dtypes of the columns the kernel writes)
};

#if defined(__HIPCC__)
#include "device_utils.hpp"
namespace sq {
// one row of the program: *valid = the result is not NULL; *div0 raised when a valid row divides by zero
__device__ __forceinline__ unsigned long long sa_eval_row(const SaProgram &pr, const SaLayout &lay, const uint8_t *in, uint32_t r, bool *valid,
                                                          bool *div0) {
  unsigned long long v[SA_STACK_MAX];
  bool ok[SA_STACK_MAX];
  int sp = 0;
  for (int k = 0; k < pr.n; k++) {
    const SaInstr I = pr.ins[k];
    if (I.op == SAO_COL) {
      const SaCol &c = lay.c[I.col];
      ok[sp] = c.in_voff == SA_NONE || ((in[c.in_voff + (r >> 3)] >> (r & 7)) & 1);
      v[sp] = c.width == 8 ? ((const unsigned long long *)(in + c.in_off))[r] : (unsigned long long)(long long)((const int32_t *)(in + c.in_off))[r];
      sp++;
    } else if (I.op == SAO_CONST) {
      ok[sp] = !I.is_null;
      v[sp] = I.imm;
      sp++;
    } else if (I.op == SAO_CAST) {
      unsigned long long x = v[sp - 1];
      bool o = ok[sp - 1];
      if (I.from == SQLRS_BOOLEAN || I.from == SQLRS_INT32 || (I.from == SQLRS_INT64 && I.dtype != SQLRS_INT32)) { // widening / exact sources
        const long long iv = (long long)x;
        if (I.dtype == SQLRS_FLOAT64) x = (unsigned long long)__double_as_longlong((double)iv);
      } else if (I.from == SQLRS_INT64) { // -> int32: out of range = NULL
        const long long iv = (long long)x;
        const bool in_range = iv <= 2147483647ll && iv >= -2147483648ll;
        x = in_range ? x : 0ull;
        o = o && in_range;
      } else { // float64 -> integer
        const double f = __longlong_as_double((long long)x);
        const double lim = I.dtype == SQLRS_INT32 ? 2147483648.0 : 9223372036854775808.0;
        const bool in_range = (f > -lim - 1) && (f < lim);
        x = in_range ? (I.dtype == SQLRS_INT32 ? (unsigned long long)(long long)(int32_t)f : (unsigned long long)(long long)f) : 0ull;
        o = o && in_range;
      }
      v[sp - 1] = x;
      ok[sp - 1] = o;
    } else {
      const unsigned long long b = v[sp - 1], a = v[sp - 2];
      const bool bo = ok[sp - 1], ao = ok[sp - 2];
      sp -= 2;
      unsigned long long res = 0;
      bool ro = ao && bo;
      if (I.op >= SAO_ADD && I.op <= SAO_DIV) {
        if (I.dtype == SQLRS_FLOAT64) {
          const double x = __longlong_as_double((long long)a), y = __longlong_as_double((long long)b);
          double q = 0;
          if (I.op == SAO_DIV) {
            if (ro && y == 0.0) *div0 = true;
            else if (ro) q = x / y;
          } else
            q = I.op == SAO_ADD ? x + y : I.op == SAO_SUB ? x - y : x * y;
          res = (unsigned long long)__double_as_longlong(q);
        } else if (I.dtype == SQLRS_INT64) {
          if (I.op == SAO_DIV) {
            if (ro && b == 0) *div0 = true;
            else if (ro) res = (long long)b == -1 ? 0ull - a : (unsigned long long)((long long)a / (long long)b);
          } else
            res = I.op == SAO_ADD ? a + b : I.op == SAO_SUB ? a - b : a * b;
        } else { // int32: wraps at 32 bits, kept sign-extended
          const uint32_t x = (uint32_t)a, y = (uint32_t)b;
          uint32_t q = 0;
          if (I.op == SAO_DIV) {
            if (ro && y == 0) *div0 = true;
            else if (ro) q = (int32_t)y == -1 ? 0u - x : (uint32_t)((int32_t)x / (int32_t)y);
          } else
            q = I.op == SAO_ADD ? x + y : I.op == SAO_SUB ? x - y : x * y;
          res = (unsigned long long)(long long)(int32_t)q;
        }
      } else if (I.op >= SAO_GT && I.op <= SAO_NE) {
        bool lt, eq;
        if (I.dtype == SQLRS_FLOAT64) {
          const unsigned long long x = f64_to_ordered(__longlong_as_double((long long)a)), y = f64_to_ordered(__longlong_as_double((long long)b));
          lt = x < y;
          eq = x == y;
        } else { // int32 (sign-extended), int64, bool (0 / 1)
          lt = (long long)a < (long long)b;
          eq = a == b;
        }
        res = I.op == SAO_GT ? (!lt && !eq) : I.op == SAO_LT ? lt : I.op == SAO_GE ? !lt : I.op == SAO_LE ? (lt || eq) : I.op == SAO_EQ ? eq : !eq;
      } else if (I.op == SAO_AND) { // Kleene (bool_words_kernel, mode 6)
        const bool kf = (ao && !a) || (bo && !b), kt = ao && a && bo && b;
        res = kt;
        ro = kf || kt;
      } else { // SAO_OR (mode 7)
        const bool kt = (ao && a) || (bo && b), kf = ao && !a && bo && !b;
        res = kt;
        ro = kf || kt;
      }
      v[sp] = res;
      ok[sp] = ro;
      sp++;
    }
  }
  *valid = ok[0];
  return v[0];
}

// Output positions of the kept rows of <= 4096 rows on ONE 1024-thread workgroup: pos[t] for row t * 1024 + tid, bit t of
// the return value = kept; *total = kept rows.  `s_w`: 17 words of LDS.
template <class KeepFn>
__device__ __forceinline__ uint32_t sa_positions(uint32_t rows, KeepFn keep_of, uint32_t (&pos)[4], uint32_t *s_w, uint32_t *total) {
  const int lane = lane_id(), w = wave_id();
  uint32_t base = 0, bits = 0;
#pragma unroll
  for (int t = 0; t < 4; t++) {
    if ((uint32_t)t * 1024u >= rows) break; // (uniform)
    const uint32_t r = (uint32_t)t * 1024u + threadIdx.x;
    const bool k = r < rows && keep_of(r, t);
    const uint64_t bm = __ballot(k);
    if (lane == 0) s_w[w] = (uint32_t)__popcll(bm);
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const uint32_t c = s_w[q];
      before += q < w ? c : 0;
      all += c;
    }
    pos[t] = base + before + mbcnt(bm);
    bits |= k ? 1u << t : 0u;
    base += all;
    __syncthreads();
  }
  *total = base;
  return bits;
}
// the validity bitmap of one output column from per-row flags in LDS (`s_v[pos]` = 1 valid / 0 NULL, written by the kept
// rows before the call's first barrier); returns nothing, adds the NULLs to *s_nulls (LDS)
__device__ __forceinline__ void sa_pack_validity(const uint8_t *s_v, uint32_t total, uint8_t *out_bits, uint32_t *s_nulls) {
  __syncthreads();
  uint32_t nulls = 0;
  for (uint32_t i = threadIdx.x; i < (total + 7) / 8; i += blockDim.x) {
    uint32_t byte = 0;
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const uint32_t p = i * 8 + b;
      const uint32_t v = p < total ? s_v[p] : 0u;
      byte |= v << b;
      nulls += p < total && !v;
    }
    out_bits[i] = (uint8_t)byte;
  }
  if (nulls) atomicAdd(s_nulls, nulls);
  __syncthreads();
}
// the last step of a fast-path kernel: every thread's stores to the pinned output are pushed out, then ONE thread
// publishes the header
__device__ __forceinline__ void sa_publish(SaHeader *hdr, unsigned long long seq, uint32_t total, const uint32_t *s_nulls, int ncols,
                                           uint32_t error = 0) {
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    hdr->count = total;
    hdr->pad = error; // (1: a valid row divided by zero — sqlrs_batch_wait turns it into the evaluator's Arrow error)
    for (int c = 0; c < ncols; c++) hdr->nulls[c] = s_nulls[c];
    __threadfence_system();
    __hip_atomic_store(&hdr->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
} // namespace sq
#endif
