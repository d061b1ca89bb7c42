// common.hpp — host-side plumbing of libsqlrs_hip: errors, ctx (stream + device memory
// pool + profiling), device columns / batches and the ABI <-> device conversions.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <memory>
#include <string>
#include <vector>

#include "sqlrs_hip.h"

namespace sq {

// Tuning / test hooks (the SQLRS_* names tools/HOOKS.md lists): consulted ONLY in a process that opted in with SQLRS_HOOKS=1
// (read once) — tests/conftest.py, bench.py's A/B legs and the tools/ scripts do; a production process never calls getenv
// on an operator's path and cannot have its routes bent by a stray variable.  (Review r05 #9: 64 switches were live in
// every process.)  Deployment knobs stay plain environment reads at ctx / operator creation: SQLRS_POOL_RESERVE_GB,
// SQLRS_POOL_VMM, SQLRS_JOIN_COMPOSITE (include/sqlrs_hip.h).
inline bool hooks_enabled() {
  static const bool on = [] {
    const char *e = std::getenv("SQLRS_HOOKS");
    return e && e[0] == '1';
  }();
  return on;
}
inline const char *hook(const char *name) { return hooks_enabled() ? std::getenv(name) : nullptr; }

struct Error {
  int status;
  std::string msg;
};
[[noreturn]] inline void fail(int status, const std::string &m) { throw Error{status, m}; }

#define SQ_HIP(expr)                                                                              \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess)                                                                         \
      ::sq::fail(SQLRS_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));            \
  } while (0)

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t round_up(size_t a, size_t b) { return (a + b - 1) / b * b; }
// bytes of an Arrow bitmap padded to whole u64 words (device kernels read/write words)
inline size_t bitmap_bytes(int64_t rows) { return (size_t)ceil_div(rows, 64) * 8; }

inline size_t width_of(int32_t dtype) {
  switch (dtype) {
  case SQLRS_INT32:
  case SQLRS_UINT32:
    return 4;
  case SQLRS_INT64:
  case SQLRS_UINT64:
  case SQLRS_FLOAT64:
    return 8;
  default:
    return 0;
  }
}

// ------------------------------------------------------------------ memory pool --
// Stream-ordered caching allocator: every kernel of a ctx runs on ctx->stream, so a block
// released while work is in flight may be handed to later work on the same stream.
// The pool outlives its ctx while blocks of it are still referenced (batches released after
// sqlrs_ctx_destroy): every Buf holds a reference, and a block returned to a closed pool is freed
// instead of cached.
struct Pool {
  std::multimap<size_t, void *> free_blocks;
  size_t live_bytes = 0, cached_bytes = 0;
  bool closed = false; // the ctx is gone: no stream to order reuse on, blocks go straight back to the driver
  int device = 0;
  // optional arena (SQLRS_POOL_RESERVE_GB): ONE hipMalloc at ctx creation that new blocks are carved from
  uint8_t *arena = nullptr;
  size_t arena_bytes = 0, arena_used = 0;
  bool in_arena(const void *p) const { return arena && (const uint8_t *)p >= arena && (const uint8_t *)p < arena + arena_bytes; }
  // blocks backed through the virtual-memory API (SQLRS_POOL_VMM=1, placement experiment of round 6: one physical handle
  // per block at the recommended granularity): pointer -> {handle, mapped size}
  struct VmmBlock {
    std::vector<hipMemGenericAllocationHandle_t> handles; // equal pieces of `size`
    size_t size = 0;
  };
  std::map<void *, VmmBlock> vmm_blocks;
  int vmm = -1; // -1: environment not read yet
  void *vmm_alloc(size_t want);
  bool vmm_free(void *p); // false: not a VMM block
  void *alloc(size_t bytes, size_t *cap);
  void release(void *p, size_t cap);
  void trim();
};

struct Ctx;
struct Buf {
  Ctx *ctx; // only valid while the ctx lives: never dereferenced by the destructor
  std::shared_ptr<Pool> pool;
  void *p;
  size_t cap;
  bool owned = true; // false: a view of memory this library does not own (never released)
  std::shared_ptr<Buf> parent; // a view into a larger block of this library: keeps that block alive
  Buf(Ctx *c, void *ptr, size_t n);
  ~Buf();
  Buf(const Buf &) = delete;
  Buf &operator=(const Buf &) = delete;
  template <class T> T *as() const { return (T *)p; }
};
using BufP = std::shared_ptr<Buf>;
// `bytes` at `p` inside `parent`, kept alive by it (never released on its own)
inline BufP buf_view(const BufP &parent, const void *p, size_t bytes) {
  BufP v = std::make_shared<Buf>(parent->ctx, const_cast<void *>(p), bytes);
  v->owned = false;
  v->parent = parent;
  return v;
}

struct ProfEntry {
  const char *name;
  double total_ms = 0;
  int64_t launches = 0;
};
struct ProfPending {
  int entry;
  hipEvent_t a, b;
};

struct Ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string last_error;
  std::shared_ptr<Pool> pool_ref = std::make_shared<Pool>();
  Pool &pool = *pool_ref;
  int num_cus = 256;
  // profiling
  bool prof_on = false;
  std::vector<ProfEntry> prof;
  std::vector<ProfPending> prof_pending;
  std::vector<hipEvent_t> event_pool;
  hipEvent_t order_event = nullptr; // sqlrs_ctx_wait_stream / release_to_stream
  // small pinned staging area for device->host scalars
  void *pinned = nullptr;
  size_t pinned_bytes = 0;

  BufP alloc(size_t bytes);
  BufP alloc_zero(size_t bytes);
  // small zeroed buffers (flags, counters, descriptors) are carved out of a block that was zeroed
  // with ONE memset: every hipMemsetAsync is a ~5 us device operation of its own
  BufP zero_block;
  size_t zero_used = 0;
  // kernels of this device that were granted > 64 KiB of dynamic LDS (hipFuncSetAttribute is per device)
  std::set<const void *> big_lds_set;
  std::map<const void *, int> occ_cache;
  int64_t lb_timeouts = 0; // look-back launches that timed out and were redone with tickets
  int64_t order_lb_fallbacks = 0; // Order split passes whose chained look-back ran out of spins (redone in the counting form)
  int lb_backoff = 0; // resident blocks per CU of the persistent kernels, per kernel
  int64_t async_fast_batches = 0;   // batches the single-batch async path took with its one-launch kernels (profile: "async_fast_batches")
  std::shared_ptr<void> small_ring; // the pinned ring of the single-batch async path (small_async.hpp), created on first use
  void sync() { SQ_HIP(hipStreamSynchronize(stream)); }
  // copies `bytes` from device to the pinned area and synchronises; returns host pointer
  const void *fetch(const void *dptr, size_t bytes);
  template <class T> T fetch_value(const T *dptr) { return *(const T *)fetch(dptr, sizeof(T)); }
  // the same in two halves (round 6): the copy is queued where the words are ready, the host waits for it — an event, not the
  // stream — after it has queued the kernels that do not depend on the answer: the round trip hides behind them
  void *pinned_early = nullptr;
  hipEvent_t early_event = nullptr;
  bool fetch_early(const void *dptr, size_t bytes); // false: not available (nothing queued)
  const void *fetch_early_wait();
  int prof_entry(const char *name);
  void prof_resolve();
};

// > 64 KiB of dynamic LDS needs an opt-in per kernel — and per device: remembered in the ctx, not in a
// function-local static (distinct ctxs may sit on distinct GPUs and threads)
template <class K> inline void allow_big_lds(Ctx *ctx, K kfn, int bytes = 150 * 1024) {
  const void *f = (const void *)kfn;
  if (ctx->big_lds_set.count(f)) return;
  SQ_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  ctx->big_lds_set.insert(f);
}

// RAII profiling scope: two event records around a launch group when enabled.
struct ProfScope {
  Ctx *ctx;
  int entry = -1;
  hipEvent_t a = nullptr, b = nullptr;
  ProfScope(Ctx *c, const char *name);
  ~ProfScope();
};

// ------------------------------------------------------------- device columns --
// A column whose buffers are in HBM.  Bitmaps (validity, BOOLEAN values) are padded to
// whole u64 words.  `stride` 0 marks a broadcast scalar (1 element / 1 word buffers).
struct DCol {
  int32_t dtype = SQLRS_NULLTYPE;
  int64_t length = 0;
  int64_t null_count = 0; // -1 = unknown
  int stride = 1;
  const void *values = nullptr;
  const uint64_t *validity = nullptr; // nullptr = all valid
  const int32_t *offsets = nullptr;   // UTF8
  int64_t data_bytes = 0;             // UTF8: bytes in `values`
  uint64_t scalar_bits = 0;           // stride 0: host copy of the value (bit pattern)
  bool scalar_null = false;           // stride 0: ScalarValue::X(None)
  BufP own_values, own_validity, own_offsets; // keep-alive (null when borrowed)
  bool has_nulls() const { return validity != nullptr && null_count != 0; }
  template <class T> const T *v() const { return (const T *)values; }
};

struct DBatch {
  int64_t rows = 0;
  std::vector<DCol> cols;
};

// Lazily uploading view of a caller batch: a column is moved to HBM the first time an
// operator touches it ("Arrow column buffers move to HBM once per pipeline").
struct InBatch {
  Ctx *ctx;
  const sqlrs_batch_t *abi;
  std::vector<DCol> cache;
  std::vector<uint8_t> loaded;
  bool host_upload = false; // an async H2D copy out of the caller's memory was queued
  const DBatch *shared = nullptr; // the batch is one of this library's own device batches (ctx.hip)
  InBatch(Ctx *c, const sqlrs_batch_t *b);
  // The caller's buffers are only borrowed for the duration of the call: uploads out of host
  // memory must have been read before the entry point returns (operators that stage their input
  // no longer synchronise on their own).
  ~InBatch();
  InBatch(const InBatch &) = delete;
  InBatch &operator=(const InBatch &) = delete;
  int64_t rows() const { return abi->num_rows; }
  int num_columns() const { return abi->num_columns; }
  int32_t dtype(int i) const { return abi->columns[i].dtype; }
  const DCol &col(int i);
  // all columns, device resident; owned=true forces private copies (for retaining operators)
  DBatch materialize(bool owned);
};

DCol upload_column(Ctx *ctx, const sqlrs_column_t &c, bool force_copy);
// Device batch -> library-owned ABI batch in `out_mem` (host: D2H into malloc'd buffers).
sqlrs_batch_t *emit_batch(Ctx *ctx, DBatch &&b, int out_mem);
// a HOST batch of `rows` rows copied out of host memory (values[c]: rows x width bytes; validity[c]: bitmap or null;
// fixed-width dtypes only) — what the single-batch async path hands out (small_async.hpp)
// (a Utf8 column: offsets[c] = its rows + 1 int32 offsets starting at 0, values[c] = its bytes)
sqlrs_batch_t *emit_host_copy(Ctx *ctx, int ncols, const int32_t *dtypes, int64_t rows, const void *const *values,
                              const uint8_t *const *validity, const int64_t *null_counts, const int32_t *const *offsets = nullptr);
sqlrs_batch_t *emit_host_columns(Ctx *ctx, std::vector<sqlrs_column_t> &&cols, int64_t rows); // takes over malloc'd blocks
// *_push_many (small HOST batches handled together): every column fixed width? / rows [cut[i], cut[i + 1]) of a device batch
// as n library-owned HOST batches through one pinned copy per column (ctx.hip)
bool all_fixed_width(const DBatch &b);
void split_rows_to_host(Ctx *ctx, const DBatch &o, const std::vector<int64_t> &cut, void **pin, size_t *pin_cap, int n,
                        sqlrs_batch_t **out);
DCol make_null_column(Ctx *ctx, int32_t dtype, int64_t n);
int64_t count_nulls(Ctx *ctx, const DCol &c);

template <class F> int guard(Ctx *ctx, F &&f) {
  try {
    f();
    return SQLRS_OK;
  } catch (const Error &e) {
    if (ctx) ctx->last_error = e.msg;
    return e.status;
  } catch (const std::exception &e) {
    if (ctx) ctx->last_error = e.what();
    return SQLRS_ERR_INTERNAL;
  }
}

// Test hook (read once): SQLRS_FORCE_TICKET=1 makes the look-back kernels skip their fast launch
// (tile id = block index, bounded spin) and use the ticketed fallback launch right away, so the
// fallback is exercised by the parity tests instead of only by a timeout.
inline int first_lookback_mode() {
  static const int forced = [] {
    const char *e = hook("SQLRS_FORCE_TICKET");
    return (e && e[0] == '1') ? 1 : 0;
  }();
  return forced;
}

// First launch mode of a look-back kernel on this ctx: 0 = fast (tile id = block index, needs every
// block resident, bounded spin), 1 = ticketed.  After a spin timeout (another ctx / process / RCCL kernel
// on the GPU broke residency) the ctx goes straight to tickets for the next LB_BACKOFF_CALLS calls
// instead of sitting out the timeout again, and the event is counted (sqlrs_ctx_profile_read reports it
// as "lookback_ticket_reruns").
constexpr int LB_BACKOFF_CALLS = 64;
inline int lookback_start_mode(Ctx *ctx) {
  if (first_lookback_mode()) return 1;
  if (ctx->lb_backoff > 0) {
    ctx->lb_backoff--;
    return 1;
  }
  return 0;
}
inline void lookback_timed_out(Ctx *ctx) {
  ctx->lb_timeouts++;
  ctx->lb_backoff = LB_BACKOFF_CALLS;
}

// Expression (postfix) owned copy
struct Expr {
  std::vector<sqlrs_expr_node_t> nodes;
  std::vector<std::string> strings;
  bool empty() const { return nodes.empty(); }
};
Expr expr_from_abi(const sqlrs_expr_t *e);

} // namespace sq

struct sqlrs_ctx : sq::Ctx {};
