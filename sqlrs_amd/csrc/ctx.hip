// ctx.hip — ctx lifecycle, device memory pool, ABI <-> HBM column conversion, timers.
#include <cstdlib>

#include <array>
#include <atomic>
#include <mutex>
#include <unordered_set>

#include "common.hpp"
#include "host_stage.hpp"
#include "prims.hpp"
#include "small_async.hpp"

extern "C" void sqlrs_batch_release(sqlrs_batch_t *batch);
namespace sq {

// ------------------------------------------------------------------------ pool --
static size_t pool_size_class(size_t bytes) {
  if (bytes < 512) return 512;
  if (bytes <= (1u << 20)) { // next power of two
    size_t s = 512;
    while (s < bytes) s <<= 1;
    return s;
  }
  return round_up(bytes, (size_t)2 << 20); // 2 MiB granules
}

void *Pool::vmm_alloc(size_t want) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || !gran) return nullptr;
  // (one physical handle per block; pieces of 32 MiB ... 1 GiB per handle and 1 / 4 GiB virtual alignment were measured too:
  //  profiles/r06e_placement.txt — pieces are slower, alignment makes no difference)
  const size_t chunk = 0, size = round_up(want, gran), align = gran;
  void *va = nullptr;
  if (hipMemAddressReserve(&va, size, align, nullptr, 0) != hipSuccess) return nullptr;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  VmmBlock blk;
  blk.size = size;
  const size_t step = chunk ? chunk : size;
  bool ok = true;
  for (size_t off = 0; off < size && ok; off += step) {
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, step, &prop, 0) != hipSuccess) {
      ok = false;
      break;
    }
    if (hipMemMap((uint8_t *)va + off, step, 0, h, 0) != hipSuccess) {
      (void)hipMemRelease(h);
      ok = false;
      break;
    }
    blk.handles.push_back(h);
  }
  if (ok && hipMemSetAccess(va, size, &acc, 1) != hipSuccess) ok = false;
  if (!ok) {
    for (size_t i = 0; i < blk.handles.size(); i++) {
      (void)hipMemUnmap((uint8_t *)va + i * step, step);
      (void)hipMemRelease(blk.handles[i]);
    }
    (void)hipMemAddressFree(va, size);
    return nullptr;
  }
  vmm_blocks[va] = std::move(blk);
  return va;
}
bool Pool::vmm_free(void *p) {
  auto it = vmm_blocks.find(p);
  if (it == vmm_blocks.end()) return false;
  const VmmBlock &blk = it->second;
  const size_t step = blk.size / blk.handles.size();
  for (size_t i = 0; i < blk.handles.size(); i++) {
    (void)hipMemUnmap((uint8_t *)p + i * step, step);
    (void)hipMemRelease(blk.handles[i]);
  }
  (void)hipMemAddressFree(p, blk.size);
  vmm_blocks.erase(it);
  return true;
}
void *Pool::alloc(size_t bytes, size_t *cap) {
  size_t want = pool_size_class(bytes);
  auto it = free_blocks.lower_bound(want);
  if (it != free_blocks.end() && it->first <= want + want / 4 + (1u << 20)) {
    void *p = it->second;
    *cap = it->first;
    cached_bytes -= it->first;
    live_bytes += it->first;
    free_blocks.erase(it);
    return p;
  }
  void *p = nullptr;
  if (arena && arena_used + want <= arena_bytes) { // carve from the reserved arena
    p = arena + arena_used;
    arena_used += want;
    *cap = want;
    live_bytes += want;
    return p;
  }
  if (vmm < 0) {
    const char *v = std::getenv("SQLRS_POOL_VMM");
    vmm = v ? std::atoi(v) : 0;
  }
  if (vmm > 0 && want >= ((size_t)vmm << 20)) { // (blocks of SQLRS_POOL_VMM MiB and more)
    if ((p = vmm_alloc(want))) {
      *cap = want;
      live_bytes += want;
      return p;
    }
  }
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess) {
    trim();
    e = hipMalloc(&p, want);
  }
  if (e != hipSuccess)
    fail(SQLRS_ERR_DEVICE, "hipMalloc(" + std::to_string(want) + "): " + hipGetErrorString(e));
  *cap = want;
  live_bytes += want;
  return p;
}
void Pool::release(void *p, size_t cap) {
  live_bytes -= cap;
  if (closed) { // the ctx (and its stream) is gone: nothing can be in flight on it any more
    if (in_arena(p)) return; // (the arena itself is leaked with a closed pool: only when batches outlive their ctx)
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != device) (void)hipSetDevice(device);
    if (!vmm_free(p)) (void)hipFree(p);
    if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
    return;
  }
  cached_bytes += cap;
  free_blocks.emplace(cap, p);
}
void Pool::trim() {
  size_t kept = 0;
  std::multimap<size_t, void *> keep;
  for (auto &kv : free_blocks) {
    if (in_arena(kv.second)) { // arena blocks cannot go back to the driver one by one: they stay cached
      keep.emplace(kv.first, kv.second);
      kept += kv.first;
    } else if (!vmm_free(kv.second))
      (void)hipFree(kv.second);
  }
  free_blocks.swap(keep);
  cached_bytes = kept;
}

Buf::Buf(Ctx *c, void *ptr, size_t n) : ctx(c), pool(c->pool_ref), p(ptr), cap(n) {}
Buf::~Buf() {
  if (p && owned) pool->release(p, cap);
}

BufP Ctx::alloc(size_t bytes) {
  size_t cap = 0;
  void *p = pool.alloc(bytes ? bytes : 8, &cap);
  return std::make_shared<Buf>(this, p, cap);
}
BufP Ctx::alloc_zero(size_t bytes) {
  constexpr size_t ZERO_BLOCK = 64 * 1024, ZERO_SMALL = 4096;
  if (bytes <= ZERO_SMALL) {
    const size_t need = round_up(bytes ? bytes : 8, 256);
    if (!zero_block || zero_used + need > ZERO_BLOCK) {
      zero_block = alloc(ZERO_BLOCK); // the old block lives on while views of it do
      SQ_HIP(hipMemsetAsync(zero_block->p, 0, ZERO_BLOCK, stream));
      zero_used = 0;
    }
    BufP v = std::make_shared<Buf>(this, (uint8_t *)zero_block->p + zero_used, need);
    v->owned = false;
    v->parent = zero_block;
    zero_used += need;
    return v;
  }
  BufP b = alloc(bytes);
  SQ_HIP(hipMemsetAsync(b->p, 0, bytes ? bytes : 8, stream));
  return b;
}
// (Round 6 measured a POLLED form — one workgroup copies the bytes into a coherent host-mapped block and release-stores a sequence
//  word the host spins on, as the single-batch async path publishes its headers: C4 (four fetches per query) 2.199 / 2.191 ms
//  against 2.207 / 2.201, C3 within the noise.  The runtime's synchronisation already spins; the ~20 us between a fetched word and
//  the next kernel are the launch path, not the wake-up.  The copy call stays.)
const void *Ctx::fetch(const void *dptr, size_t bytes) {
  if (bytes > pinned_bytes) fail(SQLRS_ERR_INTERNAL, "fetch too large");
  SQ_HIP(hipMemcpyAsync(pinned, dptr, bytes, hipMemcpyDeviceToHost, stream));
  sync();
  return pinned;
}
bool Ctx::fetch_early(const void *dptr, size_t bytes) {
  if (bytes > pinned_bytes) return false;
  if (!pinned_early && hipHostMalloc(&pinned_early, pinned_bytes, hipHostMallocDefault) != hipSuccess) {
    pinned_early = nullptr;
    return false;
  }
  if (!early_event && hipEventCreateWithFlags(&early_event, hipEventDisableTiming) != hipSuccess) {
    early_event = nullptr;
    return false;
  }
  SQ_HIP(hipMemcpyAsync(pinned_early, dptr, bytes, hipMemcpyDeviceToHost, stream));
  SQ_HIP(hipEventRecord(early_event, stream));
  return true;
}
const void *Ctx::fetch_early_wait() {
  SQ_HIP(hipEventSynchronize(early_event));
  return pinned_early;
}
int Ctx::prof_entry(const char *name) {
  for (size_t i = 0; i < prof.size(); i++)
    if (prof[i].name == name || std::strcmp(prof[i].name, name) == 0) return (int)i;
  prof.push_back(ProfEntry{name});
  return (int)prof.size() - 1;
}
void Ctx::prof_resolve() {
  if (prof_pending.empty()) return;
  sync();
  for (auto &p : prof_pending) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      prof[p.entry].total_ms += ms;
      prof[p.entry].launches++;
    }
    event_pool.push_back(p.a);
    event_pool.push_back(p.b);
  }
  prof_pending.clear();
}

ProfScope::ProfScope(Ctx *c, const char *name) : ctx(c) {
  if (!c->prof_on) return;
  entry = c->prof_entry(name);
  auto get = [&]() {
    if (!c->event_pool.empty()) {
      hipEvent_t e = c->event_pool.back();
      c->event_pool.pop_back();
      return e;
    }
    hipEvent_t e;
    SQ_HIP(hipEventCreate(&e));
    return e;
  };
  a = get();
  b = get();
  (void)hipEventRecord(a, c->stream);
}
ProfScope::~ProfScope() {
  if (entry < 0) return;
  (void)hipEventRecord(b, ctx->stream);
  ctx->prof_pending.push_back(ProfPending{entry, a, b});
  if (ctx->prof_pending.size() > 4096) {
    try {
      ctx->prof_resolve();
    } catch (...) {
    }
  }
}

// ------------------------------------------------------------------- columns --
Expr expr_from_abi(const sqlrs_expr_t *e) {
  Expr x;
  if (!e || e->num_nodes <= 0 || !e->nodes) fail(SQLRS_ERR_INTERNAL, "empty expression");
  x.nodes.assign(e->nodes, e->nodes + e->num_nodes);
  x.strings.resize(x.nodes.size());
  for (size_t i = 0; i < x.nodes.size(); i++) {
    if (x.nodes[i].s) x.strings[i] = x.nodes[i].s;
    x.nodes[i].s = nullptr;
  }
  return x;
}

static void copy_in(Ctx *ctx, void *dst, const void *src, size_t bytes, int mem) {
  if (!bytes) return;
  SQ_HIP(hipMemcpyAsync(dst, src, bytes,
                        mem == SQLRS_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice,
                        ctx->stream));
}

DCol upload_column(Ctx *ctx, const sqlrs_column_t &c, bool force_copy) {
  DCol d;
  d.dtype = c.dtype;
  d.length = c.length;
  d.null_count = c.validity ? c.null_count : 0;
  bool copy = force_copy || c.mem == SQLRS_MEM_HOST;
  if (c.mem != SQLRS_MEM_HOST && c.mem != SQLRS_MEM_DEVICE) fail(SQLRS_ERR_ARROW, "bad column mem");
  size_t nbits = (size_t)ceil_div(c.length, 8);
  // device bitmaps are read as whole u64 words: a misaligned one is copied into a padded buffer
  const bool val_misaligned = c.validity && ((uintptr_t)c.validity & 7);
  if (c.validity && c.null_count != 0) {
    if (copy || val_misaligned) {
      d.own_validity = ctx->alloc_zero(bitmap_bytes(c.length));
      copy_in(ctx, d.own_validity->p, c.validity, nbits, c.mem);
      d.validity = d.own_validity->as<uint64_t>();
    } else
      d.validity = (const uint64_t *)c.validity;
  }
  switch (c.dtype) {
  case SQLRS_BOOLEAN:
    if (copy || ((uintptr_t)c.values & 7)) {
      d.own_values = ctx->alloc_zero(bitmap_bytes(c.length));
      copy_in(ctx, d.own_values->p, c.values, nbits, c.mem);
      d.values = d.own_values->p;
    } else
      d.values = c.values;
    break;
  case SQLRS_UTF8: {
    if (!c.offsets) fail(SQLRS_ERR_ARROW, "utf8 column without offsets");
    int32_t last = 0;
    if (c.mem == SQLRS_MEM_HOST)
      last = c.offsets[c.length];
    else {
      SQ_HIP(hipMemcpyAsync(&last, c.offsets + c.length, 4, hipMemcpyDeviceToHost, ctx->stream));
      ctx->sync();
    }
    d.data_bytes = last;
    if (copy) {
      d.own_offsets = ctx->alloc(4 * (size_t)(c.length + 1));
      copy_in(ctx, d.own_offsets->p, c.offsets, 4 * (size_t)(c.length + 1), c.mem);
      d.offsets = d.own_offsets->as<int32_t>();
      d.own_values = ctx->alloc((size_t)last + 8);
      copy_in(ctx, d.own_values->p, c.values, (size_t)last, c.mem);
      d.values = d.own_values->p;
    } else {
      d.offsets = c.offsets;
      d.values = c.values;
    }
    break;
  }
  default: {
    size_t w = width_of(c.dtype);
    if (!w) fail(SQLRS_ERR_INTERNAL, "unsupported column dtype " + std::to_string(c.dtype));
    if (copy) {
      d.own_values = ctx->alloc(w * (size_t)c.length + 16);
      copy_in(ctx, d.own_values->p, c.values, w * (size_t)c.length, c.mem);
      d.values = d.own_values->p;
    } else
      d.values = c.values;
  }
  }
  return d;
}

// A batch this library produced (sqlrs_batch_t::owner points here).  Live records are registered,
// so an `owner` field that a caller forgot to clear is never dereferenced.
constexpr uint64_t OWNED_BATCH_MAGIC = 0x5351425443483031ull;
static std::mutex g_owned_mu;
static std::unordered_set<const void *> g_owned;
struct OwnedBatch {
  sqlrs_batch_t abi;
  std::vector<sqlrs_column_t> descs;
  DBatch dev;                     // keeps device buffers alive (out_mem == DEVICE)
  std::vector<void *> host_blocks; // malloc'd (out_mem == HOST)
  uint64_t magic = OWNED_BATCH_MAGIC;
  Ctx *ctx = nullptr;
  int out_mem = SQLRS_MEM_HOST;
  ArrowArray imported = {}; // sqlrs_batch_import_arrow: the array whose buffers the columns point into (released with the batch)
  OwnedBatch() {
    std::lock_guard<std::mutex> lk(g_owned_mu);
    g_owned.insert(this);
  }
  ~OwnedBatch() {
    {
      std::lock_guard<std::mutex> lk(g_owned_mu);
      g_owned.erase(this);
    }
    magic = 0;
    for (void *p : host_blocks) std::free(p);
    if (imported.release) imported.release(&imported);
  }
};

// A device batch produced by this library on the same ctx and handed back unchanged: its columns'
// buffers are reference counted (BufP), so an operator that retains input (join build side,
// staged aggregation input) shares them instead of taking private copies; the caller's
// sqlrs_batch_release only drops its own reference.
static const DBatch *shared_columns_of(Ctx *c, const sqlrs_batch_t *b) {
  if (!b->owner) return nullptr;
  {
    std::lock_guard<std::mutex> lk(g_owned_mu);
    if (!g_owned.count(b->owner)) return nullptr;
  }
  const OwnedBatch *o = (const OwnedBatch *)b->owner;
  if (&o->abi != b || o->magic != OWNED_BATCH_MAGIC || o->ctx != c || o->out_mem != SQLRS_MEM_DEVICE) return nullptr;
  if (b->columns != o->descs.data() || b->num_columns != (int)o->dev.cols.size() || b->num_rows != o->dev.rows)
    return nullptr;
  return &o->dev;
}

InBatch::InBatch(Ctx *c, const sqlrs_batch_t *b) : ctx(c), abi(b) {
  if (!b) fail(SQLRS_ERR_ARROW, "null batch");
  if (b->num_columns < 0 || (b->num_columns > 0 && !b->columns)) fail(SQLRS_ERR_ARROW, "bad batch");
  // ONE limit for every operator (sqlrs_hip.h, sqlrs_batch_t): row ids are 32-bit (30-bit in the slim partition records and
  // the gather indices), so a batch holds fewer than 2^31 rows; larger inputs arrive as several batches
  if (b->num_rows < 0 || b->num_rows >= (1ll << 31)) fail(SQLRS_ERR_ARROW, "batch num_rows outside [0, 2^31)");
  for (int i = 0; i < b->num_columns; i++)
    if (b->columns[i].length != b->num_rows) fail(SQLRS_ERR_ARROW, "column length != num_rows");
  cache.resize((size_t)b->num_columns);
  loaded.assign((size_t)b->num_columns, 0);
  shared = shared_columns_of(c, b);
}
InBatch::~InBatch() {
  if (host_upload) (void)hipStreamSynchronize(ctx->stream);
}
const DCol &InBatch::col(int i) {
  if (i < 0 || i >= abi->num_columns) fail(SQLRS_ERR_INTERNAL, "input ref out of range");
  if (!loaded[(size_t)i] && shared && shared->cols[(size_t)i].values == abi->columns[i].values &&
      shared->cols[(size_t)i].length == abi->columns[i].length) {
    cache[(size_t)i] = shared->cols[(size_t)i]; // carries the owning BufPs
    loaded[(size_t)i] = 1;
  }
  if (!loaded[(size_t)i]) {
    if (abi->columns[i].mem == SQLRS_MEM_HOST && abi->columns[i].length > 0) host_upload = true;
    cache[(size_t)i] = upload_column(ctx, abi->columns[i], false);
    loaded[(size_t)i] = 1;
  }
  return cache[(size_t)i];
}
DBatch InBatch::materialize(bool owned) {
  DBatch b;
  b.rows = abi->num_rows;
  for (int i = 0; i < abi->num_columns; i++) {
    const DCol &c = col(i);
    const bool borrowed = (c.values && !c.own_values) || (c.validity && !c.own_validity) || (c.offsets && !c.own_offsets);
    if (owned && abi->columns[i].mem == SQLRS_MEM_DEVICE && borrowed)
      b.cols.push_back(upload_column(ctx, abi->columns[i], true));
    else
      b.cols.push_back(c);
  }
  return b;
}

int64_t count_nulls(Ctx *ctx, const DCol &c) {
  if (!c.validity) return 0;
  if (c.null_count >= 0) return c.null_count;
  return count_clear_bits(ctx, c.validity, c.length);
}

// ------------------------------------------------------------ emitting batches --
sqlrs_batch_t *emit_batch(Ctx *ctx, DBatch &&b, int out_mem) {
  if (out_mem != SQLRS_MEM_HOST && out_mem != SQLRS_MEM_DEVICE)
    fail(SQLRS_ERR_INTERNAL, "bad out_mem");
  auto o = std::unique_ptr<OwnedBatch>(new OwnedBatch());
  o->dev = std::move(b);
  o->ctx = ctx;
  o->out_mem = out_mem;
  DBatch &d = o->dev;
  const int64_t rows = d.rows;
  o->descs.resize(d.cols.size());
  for (size_t i = 0; i < d.cols.size(); i++) {
    DCol &c = d.cols[i];
    if (c.stride == 0) c = materialize_scalar(ctx, c, d.rows);
    if (c.length != d.rows) fail(SQLRS_ERR_INTERNAL, "emit: column length mismatch");
    if (c.validity && c.null_count < 0) c.null_count = count_clear_bits(ctx, c.validity, c.length);
    if (c.validity && c.null_count == 0) {
      c.validity = nullptr;
      c.own_validity.reset();
    }
    sqlrs_column_t &s = o->descs[i];
    s.dtype = c.dtype;
    s.mem = out_mem;
    s.length = c.length;
    s.null_count = c.validity ? c.null_count : 0;
    s.values = nullptr;
    s.validity = nullptr;
    s.offsets = nullptr;
    if (out_mem == SQLRS_MEM_DEVICE) {
      s.values = c.values;
      s.validity = (const uint8_t *)c.validity;
      s.offsets = c.offsets;
      continue;
    }
    auto down = [&](const void *dptr, size_t bytes) -> void * {
      void *h = std::malloc(bytes + 64);
      if (!h) fail(SQLRS_ERR_INTERNAL, "host allocation failed");
      o->host_blocks.push_back(h);
      if (bytes) SQ_HIP(hipMemcpyAsync(h, dptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
      return h;
    };
    size_t nbits = (size_t)ceil_div(c.length, 8);
    if (c.validity) s.validity = (const uint8_t *)down(c.validity, nbits);
    if (c.dtype == SQLRS_BOOLEAN)
      s.values = down(c.values, nbits);
    else if (c.dtype == SQLRS_UTF8) {
      s.offsets = (const int32_t *)down(c.offsets, 4 * (size_t)(c.length + 1));
      s.values = down(c.values, (size_t)c.data_bytes);
    } else
      s.values = down(c.values, width_of(c.dtype) * (size_t)c.length);
  }
  if (out_mem == SQLRS_MEM_HOST) {
    ctx->sync();
    o->dev = DBatch(); // device buffers go back to the pool
  }
  o->abi.num_rows = rows;
  o->abi.num_columns = (int32_t)o->descs.size();
  o->abi.reserved = 0;
  o->abi.columns = o->descs.data();
  o->abi.owner = o.get();
  return &o.release()->abi;
}

sqlrs_batch_t *emit_host_copy(Ctx *ctx, int ncols, const int32_t *dtypes, int64_t rows, const void *const *values,
                              const uint8_t *const *validity, const int64_t *null_counts, const int32_t *const *offsets) {
  auto o = std::unique_ptr<OwnedBatch>(new OwnedBatch());
  o->ctx = ctx;
  o->out_mem = SQLRS_MEM_HOST;
  o->descs.resize((size_t)ncols);
  for (int c = 0; c < ncols; c++) {
    const bool utf8 = dtypes[c] == SQLRS_UTF8 && offsets && offsets[c], boolean = dtypes[c] == SQLRS_BOOLEAN;
    const size_t w = (utf8 || boolean) ? 1 : width_of(dtypes[c]);
    if (!w) fail(SQLRS_ERR_INTERNAL, "emit_host_copy: fixed-width, Boolean and Utf8 columns only");
    sqlrs_column_t &d = o->descs[(size_t)c];
    d.dtype = dtypes[c];
    d.mem = SQLRS_MEM_HOST;
    d.length = rows;
    d.offsets = nullptr;
    d.validity = nullptr;
    d.null_count = 0;
    const size_t nbytes = utf8 ? (size_t)offsets[c][rows] : boolean ? (size_t)(rows + 7) / 8 : w * (size_t)rows; // (Boolean: bit-packed)
    void *v = std::malloc(nbytes + 64);
    if (!v) fail(SQLRS_ERR_INTERNAL, "host allocation failed");
    o->host_blocks.push_back(v);
    if (nbytes) std::memcpy(v, values[c], nbytes);
    d.values = v;
    if (utf8) {
      int32_t *po = (int32_t *)std::malloc(4 * ((size_t)rows + 1) + 64);
      if (!po) fail(SQLRS_ERR_INTERNAL, "host allocation failed");
      o->host_blocks.push_back(po);
      std::memcpy(po, offsets[c], 4 * ((size_t)rows + 1));
      d.offsets = po;
    }
    if (validity && validity[c] && null_counts[c] > 0) {
      const size_t nb = (size_t)(rows + 7) / 8;
      uint8_t *b = (uint8_t *)std::malloc(nb + 64);
      if (!b) fail(SQLRS_ERR_INTERNAL, "host allocation failed");
      o->host_blocks.push_back(b);
      std::memcpy(b, validity[c], nb);
      d.validity = b;
      d.null_count = null_counts[c];
    }
  }
  o->abi.num_rows = rows;
  o->abi.num_columns = ncols;
  o->abi.reserved = 0;
  o->abi.columns = o->descs.data();
  o->abi.owner = o.get();
  return &o.release()->abi;
}

// --------------------------------------------------------------- host staging --
bool HostStage::accepts(const sqlrs_batch_t *b) const {
  if (!b || b->num_columns <= 0 || b->num_rows < 0 || b->num_rows > HOST_STAGE_MAX_BATCH) return false; // (< 0: InBatch rejects it)
  if (has_schema && (size_t)b->num_columns != cols.size()) return false;
  for (int i = 0; i < b->num_columns; i++) {
    const sqlrs_column_t &c = b->columns[i];
    if (c.mem != SQLRS_MEM_HOST || !width_of(c.dtype) || c.length != b->num_rows) return false;
    if (has_schema && cols[(size_t)i].dtype != c.dtype) return false;
  }
  return true;
}

void HostStage::append(const sqlrs_batch_t *b) {
  if (!has_schema) {
    if (cols.size() != (size_t)b->num_columns) { // (column slots keep their pinned block across flushes of one schema)
      for (Col &c : cols)
        if (c.pin) (void)hipHostFree(c.pin);
      cols.clear();
      cols.resize((size_t)b->num_columns);
    }
    for (int i = 0; i < b->num_columns; i++) cols[(size_t)i].dtype = b->columns[i].dtype;
    has_schema = true;
  }
  const int64_t n = b->num_rows;
  // every column's room first, then the copies: a failed pinned allocation leaves the stage as it was (no column has
  // taken the batch while `rows` has not advanced; advisor, round 3)
  for (int i = 0; i < b->num_columns && n > 0; i++) {
    Col &d = cols[(size_t)i];
    const size_t w = width_of(b->columns[i].dtype), add = w * (size_t)n;
    if (!d.pin && rows + n < HOST_STAGE_PIN_ROWS) {
      d.vals.reserve(d.vals.size() + add);
      continue;
    }
    const size_t have = d.pin ? d.pin_size : d.vals.size();
    if (have + add > d.pin_cap) { // grow (doubling, at least 2^20 rows' worth): pinned blocks are kept across flushes
      size_t cap = std::max<size_t>(d.pin_cap * 2, w << 20);
      while (cap < have + add) cap *= 2;
      void *np = nullptr;
      SQ_HIP(hipHostMalloc(&np, cap, hipHostMallocDefault));
      if (have) std::memcpy(np, d.pin ? d.pin : d.vals.data(), have);
      uint8_t *old_pin = d.pin;
      d.pin = (uint8_t *)np;
      d.pin_cap = cap;
      d.pin_size = have;
      d.vals.clear();
      d.vals.shrink_to_fit();
      if (old_pin) (void)hipHostFree(old_pin);
    }
  }
  for (int i = 0; i < b->num_columns && n > 0; i++) {
    const sqlrs_column_t &c = b->columns[i];
    Col &d = cols[(size_t)i];
    const size_t w = width_of(c.dtype);
    const uint8_t *src = (const uint8_t *)c.values;
    const size_t add = w * (size_t)n;
    if (!d.pin && rows + n < HOST_STAGE_PIN_ROWS) {
      d.vals.insert(d.vals.end(), src, src + add);
    } else {
      if (!d.pin_size && !d.vals.empty()) { // (a kept buffer, values still in the vector: cannot happen, kept for safety)
        std::memcpy(d.pin, d.vals.data(), d.vals.size());
        d.pin_size = d.vals.size();
        d.vals.clear();
      }
      std::memcpy(d.pin + d.pin_size, src, add);
      d.pin_size += add;
    }
    const bool nullable = c.validity && c.null_count != 0;
    if (nullable || !d.valid.empty()) {
      const size_t words = (size_t)ceil_div(rows + n, 64);
      if (d.valid.empty()) { // first NULL-bearing batch: everything before it was valid
        d.valid.assign((size_t)ceil_div(rows, 64), ~0ull);
        if (rows & 63) d.valid.back() &= (1ull << (rows & 63)) - 1;
      }
      d.valid.resize(words, 0);
      for (int64_t r = 0; r < n; r++) {
        const bool v = !nullable || ((c.validity[r >> 3] >> (r & 7)) & 1);
        if (v) d.valid[(size_t)((rows + r) >> 6)] |= 1ull << ((rows + r) & 63);
        else d.nulls++;
      }
    }
  }
  rows += n;
}

sqlrs_batch_t *HostStage::take() {
  if (!has_schema) return nullptr;
  std::vector<sqlrs_column_t> descs(cols.size());
  for (size_t i = 0; i < cols.size(); i++) {
    sqlrs_column_t &d = descs[i];
    d.dtype = cols[i].dtype;
    d.mem = SQLRS_MEM_HOST;
    d.length = rows;
    d.null_count = cols[i].nulls;
    d.values = cols[i].pin_size ? (const void *)cols[i].pin : (const void *)cols[i].vals.data();
    d.validity = cols[i].nulls ? (const uint8_t *)cols[i].valid.data() : nullptr;
    d.offsets = nullptr;
  }
  sqlrs_batch_t hb;
  hb.num_rows = rows;
  hb.num_columns = (int32_t)descs.size();
  hb.reserved = 0;
  hb.columns = descs.data();
  hb.owner = nullptr;
  sqlrs_batch_t *dev = nullptr;
  {
    InBatch ib(ctx, &hb);
    dev = emit_batch(ctx, ib.materialize(true), SQLRS_MEM_DEVICE);
  } // (~InBatch waits for the uploads: the staged values may go now)
  // the pinned blocks stay with their column slot for the next flush (same schema); everything else is dropped
  for (Col &c : cols) {
    c.vals.clear();
    c.valid.clear();
    c.pin_size = 0;
    c.nulls = 0;
  }
  has_schema = false;
  rows = 0;
  return dev;
}

// rows [lo, hi) of a HOST column image (values / validity bitmap) as malloc'd blocks of their own
static sqlrs_column_t slice_host_column(int32_t dtype, const uint8_t *vals, const uint8_t *validity, int64_t lo, int64_t hi) {
  sqlrs_column_t c;
  std::memset(&c, 0, sizeof(c));
  const size_t w = width_of(dtype);
  const int64_t n = hi - lo;
  c.dtype = dtype;
  c.mem = SQLRS_MEM_HOST;
  c.length = n;
  void *v = std::malloc(std::max<size_t>(w * (size_t)n, 8));
  if (!v) fail(SQLRS_ERR_INTERNAL, "out of host memory");
  if (n) std::memcpy(v, vals + w * (size_t)lo, w * (size_t)n);
  c.values = v;
  if (validity) {
    int64_t nulls = 0;
    uint8_t *vb = (uint8_t *)std::calloc((size_t)(n + 7) / 8 + 8, 1);
    if (!vb) {
      std::free(v);
      fail(SQLRS_ERR_INTERNAL, "out of host memory");
    }
    for (int64_t r = 0; r < n; r++) {
      const int64_t s = lo + r;
      if ((validity[s >> 3] >> (s & 7)) & 1) vb[r >> 3] |= (uint8_t)(1u << (r & 7));
      else nulls++;
    }
    if (nulls) {
      c.validity = vb;
      c.null_count = nulls;
    } else {
      std::free(vb);
    }
  }
  return c;
}

bool all_fixed_width(const DBatch &b) {
  for (const DCol &c : b.cols)
    if (!width_of(c.dtype) || c.stride == 0) return false;
  return true;
}

// out[i] = rows [cut[i], cut[i + 1]) of the device batch `o` (fixed-width columns) as a library-owned HOST batch, i < n:
// one device-to-host copy per column into the pinned block *pin (grown on demand, kept by the caller), then one slice per
// output batch.  Synchronises the ctx stream.  On error no output batch is left allocated.
void split_rows_to_host(Ctx *ctx, const DBatch &o, const std::vector<int64_t> &cut, void **pin_p, size_t *pin_cap, int n,
                        sqlrs_batch_t **out) {
  const int nc = (int)o.cols.size();
  size_t need = 64;
  std::vector<size_t> voff((size_t)nc), boff((size_t)nc, (size_t)-1);
  for (int c = 0; c < nc; c++) {
    const DCol &col = o.cols[(size_t)c];
    voff[(size_t)c] = need;
    need += round_up(std::max<size_t>(width_of(col.dtype) * (size_t)o.rows, 8), 64);
    if (col.validity && col.null_count != 0) {
      boff[(size_t)c] = need;
      need += round_up(bitmap_bytes(o.rows) + 8, 64);
    }
  }
  if (need > *pin_cap) {
    if (*pin_p) SQ_HIP(hipHostFree(*pin_p));
    *pin_p = nullptr;
    *pin_cap = 0;
    SQ_HIP(hipHostMalloc(pin_p, need + need / 4, hipHostMallocDefault));
    *pin_cap = need + need / 4;
  }
  uint8_t *pin = (uint8_t *)*pin_p;
  for (int c = 0; c < nc; c++) {
    const DCol &col = o.cols[(size_t)c];
    const size_t w = width_of(col.dtype);
    if (o.rows) SQ_HIP(hipMemcpyAsync(pin + voff[(size_t)c], col.values, w * (size_t)o.rows, hipMemcpyDeviceToHost, ctx->stream));
    if (boff[(size_t)c] != (size_t)-1)
      SQ_HIP(hipMemcpyAsync(pin + boff[(size_t)c], col.validity, bitmap_bytes(o.rows), hipMemcpyDeviceToHost, ctx->stream));
  }
  ctx->sync();
  int made = 0;
  try {
    for (; made < n; made++) {
      const int64_t lo = cut[(size_t)made], hi = cut[(size_t)made + 1];
      std::vector<sqlrs_column_t> cols;
      try {
        for (int c = 0; c < nc; c++)
          cols.push_back(slice_host_column(o.cols[(size_t)c].dtype, pin + voff[(size_t)c],
                                           boff[(size_t)c] == (size_t)-1 ? nullptr : pin + boff[(size_t)c], lo, hi));
      } catch (...) {
        for (sqlrs_column_t &c : cols) {
          std::free(const_cast<void *>(c.values));
          std::free(const_cast<uint8_t *>(c.validity));
        }
        throw;
      }
      out[made] = emit_host_columns(ctx, std::move(cols), hi - lo);
    }
  } catch (...) {
    for (int k = 0; k < made; k++) {
      sqlrs_batch_release(out[k]);
      out[k] = nullptr;
    }
    throw;
  }
}

HostStage::~HostStage() {
  for (Col &c : cols)
    if (c.pin) (void)hipHostFree(c.pin);
}

// Host columns built by the library itself (CSV ingest) -> library-owned HOST batch.  Every pointer of
// `cols` is a malloc'd block that the batch takes over.
sqlrs_batch_t *emit_host_columns(Ctx *ctx, std::vector<sqlrs_column_t> &&cols, int64_t rows) {
  auto o = std::unique_ptr<OwnedBatch>(new OwnedBatch());
  o->ctx = ctx;
  o->out_mem = SQLRS_MEM_HOST;
  o->descs = std::move(cols);
  for (sqlrs_column_t &c : o->descs) {
    c.mem = SQLRS_MEM_HOST;
    if (c.values) o->host_blocks.push_back(const_cast<void *>(c.values));
    if (c.validity) o->host_blocks.push_back(const_cast<uint8_t *>(c.validity));
    if (c.offsets) o->host_blocks.push_back(const_cast<int32_t *>(c.offsets));
  }
  o->abi.num_rows = rows;
  o->abi.num_columns = (int32_t)o->descs.size();
  o->abi.reserved = 0;
  o->abi.columns = o->descs.data();
  o->abi.owner = o.get();
  return &o.release()->abi;
}

} // namespace sq

using namespace sq;

// =============================================================== C entry points ==
extern "C" {

int sqlrs_ctx_create(int device_id, sqlrs_ctx_t **out) {
  if (!out) return SQLRS_ERR_INTERNAL;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n)
    return SQLRS_ERR_DEVICE; // no CPU fallback exists: fail loudly
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return SQLRS_ERR_DEVICE;
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) return SQLRS_ERR_DEVICE;
  if (hipSetDevice(device_id) != hipSuccess) return SQLRS_ERR_DEVICE;
  auto *c = new sqlrs_ctx();
  c->device = device_id;
  c->pool.device = device_id;
  c->num_cus = prop.multiProcessorCount;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipHostMalloc(&c->pinned, 65536, hipHostMallocDefault) != hipSuccess) {
    delete c;
    return SQLRS_ERR_DEVICE;
  }
  c->pinned_bytes = 65536;
  if (const char *e = std::getenv("SQLRS_POOL_RESERVE_GB")) { // experiment / deployment knob: one up-front allocation
    const size_t bytes = (size_t)(std::atof(e) * (double)(1ull << 30)); // (fractions of a GiB count: "0.5", "1.5")
    void *p = nullptr;
    if (bytes && hipMalloc(&p, bytes) == hipSuccess) {
      c->pool.arena = (uint8_t *)p;
      c->pool.arena_bytes = bytes;
    }
  }
  *out = c;
  return SQLRS_OK;
}

void sqlrs_ctx_destroy(sqlrs_ctx_t *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto &p : ctx->prof_pending) {
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
  if (ctx->order_event) (void)hipEventDestroy(ctx->order_event);
  ctx->zero_block.reset();
  ctx->pool.trim();
  // batches this ctx produced may be released later: their blocks hold the pool alive and are handed
  // back to the driver directly from then on (operators must be destroyed BEFORE their ctx)
  if (ctx->pool.arena && ctx->pool.live_bytes == 0) { // nothing outlives the ctx: the arena goes back
    ctx->pool.free_blocks.clear();
    (void)hipFree(ctx->pool.arena);
    ctx->pool.arena = nullptr;
  }
  ctx->pool.closed = true;
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->pinned_early) (void)hipHostFree(ctx->pinned_early);
  if (ctx->early_event) (void)hipEventDestroy(ctx->early_event);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char *sqlrs_last_error(const sqlrs_ctx_t *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int sqlrs_ctx_synchronize(sqlrs_ctx_t *ctx) {
  return guard(ctx, [&] {
    ctx->sync();
    sa_drain(ctx); // (the async single-batch path's side streams, small_async.hpp)
  });
}
void *sqlrs_ctx_stream(sqlrs_ctx_t *ctx) { return (void *)ctx->stream; }

// stream-ordered hand-over of DEVICE buffers between the caller's stream and the ctx stream
static int order_streams(sqlrs_ctx_t *ctx, hipStream_t first, hipStream_t then) {
  return guard(ctx, [&] {
    SQ_HIP(hipSetDevice(ctx->device));
    // one dedicated event (no timing; NOT from the profiling pool, whose events are read with
    // hipEventElapsedTime); a recorded event may be re-recorded once the wait on it is queued
    if (!ctx->order_event) SQ_HIP(hipEventCreateWithFlags(&ctx->order_event, hipEventDisableTiming));
    SQ_HIP(hipEventRecord(ctx->order_event, first));
    SQ_HIP(hipStreamWaitEvent(then, ctx->order_event, 0));
  });
}
int sqlrs_ctx_wait_stream(sqlrs_ctx_t *ctx, void *producer_stream) {
  return order_streams(ctx, (hipStream_t)producer_stream, ctx->stream);
}
int sqlrs_ctx_release_to_stream(sqlrs_ctx_t *ctx, void *consumer_stream) {
  return order_streams(ctx, ctx->stream, (hipStream_t)consumer_stream);
}
int64_t sqlrs_ctx_pool_bytes(const sqlrs_ctx_t *ctx) {
  return (int64_t)(ctx->pool.live_bytes + ctx->pool.cached_bytes);
}
void sqlrs_ctx_pool_trim(sqlrs_ctx_t *ctx) {
  (void)hipStreamSynchronize(ctx->stream);
  ctx->zero_block.reset();
  ctx->pool.trim();
}

void sqlrs_batch_release(sqlrs_batch_t *batch) {
  if (!batch || !batch->owner) return;
  {
    std::lock_guard<std::mutex> lk(g_owned_mu);
    if (!g_owned.count(batch->owner)) return; // not (or no longer) one of ours: released twice, or a stray owner field
  }
  delete (OwnedBatch *)batch->owner;
}

int sqlrs_batch_copy(sqlrs_ctx_t *ctx, const sqlrs_batch_t *in, int out_mem, sqlrs_batch_t **out) {
  return guard(ctx, [&] {
    SQ_HIP(hipSetDevice(ctx->device));
    InBatch ib(ctx, in);
    DBatch b = ib.materialize(true);
    *out = emit_batch(ctx, std::move(b), out_mem);
  });
}

// ---- Arrow C Data Interface (sqlrs_hip.h) ---------------------------------------------------------------------------
namespace {
int32_t arrow_dtype(const char *f) {
  if (!f) return -1;
  if (!std::strcmp(f, "i")) return SQLRS_INT32;
  if (!std::strcmp(f, "l")) return SQLRS_INT64;
  if (!std::strcmp(f, "g")) return SQLRS_FLOAT64;
  if (!std::strcmp(f, "b")) return SQLRS_BOOLEAN;
  if (!std::strcmp(f, "u")) return SQLRS_UTF8;
  return -1;
}
const char *arrow_format(int32_t dtype) {
  switch (dtype) {
  case SQLRS_INT32: return "i";
  case SQLRS_INT64: return "l";
  case SQLRS_FLOAT64: return "g";
  case SQLRS_BOOLEAN: return "b";
  case SQLRS_UTF8: return "u";
  default: return nullptr;
  }
}
// `n` bits of `src` from bit `off` on, re-packed from bit 0 (a sliced array whose offset is not a whole byte)
uint8_t *repack_bits(const uint8_t *src, int64_t off, int64_t n, std::vector<void *> &blocks) {
  uint8_t *dst = (uint8_t *)std::calloc((size_t)(n + 7) / 8 + 8, 1);
  if (!dst) fail(SQLRS_ERR_INTERNAL, "host allocation failed");
  blocks.push_back(dst);
  for (int64_t i = 0; i < n; i++)
    if ((src[(off + i) >> 3] >> ((off + i) & 7)) & 1) dst[i >> 3] |= (uint8_t)(1u << (i & 7));
  return dst;
}
const uint8_t *bits_at(const void *buf, int64_t off, int64_t n, std::vector<void *> &blocks) {
  if (!buf) return nullptr;
  if ((off & 7) == 0) return (const uint8_t *)buf + (off >> 3);
  return repack_bits((const uint8_t *)buf, off, n, blocks);
}
int64_t clear_bits(const uint8_t *bits, int64_t n) {
  int64_t set = 0;
  for (int64_t i = 0; i < (n >> 3); i++) set += __builtin_popcount(bits[i]);
  for (int64_t i = n & ~7ll; i < n; i++) set += (bits[i >> 3] >> (i & 7)) & 1;
  return n - set;
}

// what an exported array / schema keeps alive; one reference per structure that still points into it
struct ExportHolder {
  std::atomic<int> refs{0};
  sqlrs_batch_t *batch = nullptr;
  std::vector<ArrowArray> children;
  std::vector<ArrowArray *> child_ptrs;
  std::vector<std::array<const void *, 3>> bufs;
  const void *top_buf[1] = {nullptr};
  void unref() {
    if (refs.fetch_sub(1) == 1) {
      sqlrs_batch_release(batch);
      delete this;
    }
  }
};
void export_child_release(ArrowArray *a) {
  if (!a || !a->release) return;
  ExportHolder *h = (ExportHolder *)a->private_data;
  a->release = nullptr;
  h->unref();
}
void export_parent_release(ArrowArray *a) {
  if (!a || !a->release) return;
  ExportHolder *h = (ExportHolder *)a->private_data;
  for (int64_t i = 0; i < a->n_children; i++) // (children the consumer has moved out carry release == NULL here)
    if (a->children[i] && a->children[i]->release) a->children[i]->release(a->children[i]);
  a->release = nullptr;
  h->unref();
}
struct SchemaHolder {
  std::vector<ArrowSchema> children;
  std::vector<ArrowSchema *> child_ptrs;
  std::vector<std::string> names;
};
void export_schema_child_release(ArrowSchema *s) {
  if (s) s->release = nullptr; // (its strings belong to the parent's holder)
}
void export_schema_release(ArrowSchema *s) {
  if (!s || !s->release) return;
  SchemaHolder *h = (SchemaHolder *)s->private_data;
  for (int64_t i = 0; i < s->n_children; i++)
    if (s->children[i] && s->children[i]->release) s->children[i]->release(s->children[i]);
  s->release = nullptr;
  delete h;
}
} // namespace

int sqlrs_batch_import_arrow(sqlrs_ctx_t *ctx, struct ArrowArray *array, struct ArrowSchema *schema, sqlrs_batch_t **out) {
  if (out) *out = nullptr;
  return guard(ctx, [&] {
    if (!array || !schema || !out || !array->release || !schema->release) fail(SQLRS_ERR_ARROW, "import_arrow: null or released structure");
    if (!schema->format || std::strcmp(schema->format, "+s") != 0) fail(SQLRS_ERR_ARROW, "import_arrow: a struct array (\"+s\", an exported RecordBatch) is expected");
    if (array->n_children != schema->n_children || array->dictionary || array->length < 0 || array->offset < 0)
      fail(SQLRS_ERR_ARROW, "import_arrow: array and schema do not match");
    if (array->null_count > 0 || (array->null_count < 0 && array->n_buffers > 0 && array->buffers && array->buffers[0]))
      fail(SQLRS_ERR_ARROW, "import_arrow: NULL rows at the struct level");
    if (array->length >= (1ll << 31)) fail(SQLRS_ERR_ARROW, "batch num_rows outside [0, 2^31)");
    const int64_t n = array->length;
    const int nc = (int)array->n_children;
    auto o = std::unique_ptr<OwnedBatch>(new OwnedBatch());
    o->ctx = ctx;
    o->out_mem = SQLRS_MEM_HOST;
    o->descs.resize((size_t)nc);
    for (int c = 0; c < nc; c++) {
      const ArrowArray *a = array->children[c];
      const ArrowSchema *sc = schema->children[c];
      const int32_t dt = (a && sc && !a->dictionary) ? arrow_dtype(sc->format) : -1;
      if (dt < 0) fail(SQLRS_ERR_ARROW, std::string("import_arrow: unsupported column type \"") + ((sc && sc->format) ? sc->format : "?") + "\"");
      const int64_t off = array->offset + a->offset;
      if (a->offset < 0 || a->length < array->offset + n) fail(SQLRS_ERR_ARROW, "import_arrow: child shorter than the struct");
      const int64_t need = dt == SQLRS_UTF8 ? 3 : 2;
      if (a->n_buffers < need || !a->buffers) fail(SQLRS_ERR_ARROW, "import_arrow: missing buffers");
      sqlrs_column_t &d = o->descs[(size_t)c];
      d.dtype = dt;
      d.mem = SQLRS_MEM_HOST;
      d.length = n;
      d.values = nullptr;
      d.validity = nullptr;
      d.offsets = nullptr;
      d.null_count = 0;
      if (a->null_count != 0 && a->buffers[0] && n > 0) {
        d.validity = bits_at(a->buffers[0], off, n, o->host_blocks);
        d.null_count = a->null_count > 0 && a->offset == 0 && a->length == n && array->offset == 0 ? a->null_count : clear_bits(d.validity, n);
        if (d.null_count == 0) d.validity = nullptr;
      }
      if (n > 0 && !a->buffers[1] && dt != SQLRS_UTF8) fail(SQLRS_ERR_ARROW, "import_arrow: missing values buffer");
      switch (dt) {
      case SQLRS_BOOLEAN: d.values = bits_at(a->buffers[1], off, n, o->host_blocks); break;
      case SQLRS_UTF8: {
        static const int32_t zero_off[1] = {0};
        const int32_t *po = a->buffers[1] ? (const int32_t *)a->buffers[1] + off : zero_off;
        if (po[0] != 0) { // a sliced string column: offsets re-based (the values are not copied)
          int32_t *no = (int32_t *)std::malloc(4 * (size_t)(n + 1) + 8);
          if (!no) fail(SQLRS_ERR_INTERNAL, "host allocation failed");
          o->host_blocks.push_back(no);
          for (int64_t i = 0; i <= n; i++) no[i] = po[i] - po[0];
          d.values = (const uint8_t *)a->buffers[2] + po[0];
          d.offsets = no;
        } else {
          d.offsets = po;
          d.values = a->buffers[2] ? a->buffers[2] : (const void *)zero_off; // (an all-empty column may come without a data buffer)
        }
        break;
      }
      default: d.values = (const uint8_t *)a->buffers[1] + (size_t)off * width_of(dt);
      }
    }
    o->abi.num_rows = n;
    o->abi.num_columns = nc;
    o->abi.reserved = 0;
    o->abi.columns = o->descs.data();
    o->abi.owner = o.get();
    // nothing can fail from here on: the move (Arrow C Data Interface, "moving an array")
    o->imported = *array;
    array->release = nullptr;
    schema->release(schema);
    *out = &o.release()->abi;
  });
}

int sqlrs_batch_export_arrow(sqlrs_ctx_t *ctx, sqlrs_batch_t *batch, const char *const *names, struct ArrowArray *out_array,
                             struct ArrowSchema *out_schema) {
  return guard(ctx, [&] {
    if (!batch || !out_array || !out_schema) fail(SQLRS_ERR_ARROW, "export_arrow: null argument");
    SQ_HIP(hipSetDevice(ctx->device));
    bool ours = false;
    if (batch->owner) {
      std::lock_guard<std::mutex> lk(g_owned_mu);
      ours = g_owned.count(batch->owner) != 0;
    }
    bool host = true;
    for (int c = 0; c < batch->num_columns; c++) host = host && batch->columns[c].mem == SQLRS_MEM_HOST;
    sqlrs_batch_t *hb = batch;
    if (!ours || !host) { // borrowed buffers / device columns: a host copy owned by the library
      InBatch ib(ctx, batch);
      hb = emit_batch(ctx, ib.materialize(true), SQLRS_MEM_HOST);
    }
    const int nc = hb->num_columns;
    for (int c = 0; c < nc; c++)
      if (!arrow_format(hb->columns[c].dtype)) {
        if (hb != batch) sqlrs_batch_release(hb);
        fail(SQLRS_ERR_ARROW, "export_arrow: column type without an Arrow format");
      }
    auto h = std::unique_ptr<ExportHolder>(new ExportHolder());
    auto sh = std::unique_ptr<SchemaHolder>(new SchemaHolder());
    h->children.resize((size_t)nc);
    h->child_ptrs.resize((size_t)nc);
    h->bufs.resize((size_t)nc);
    sh->children.resize((size_t)nc);
    sh->child_ptrs.resize((size_t)nc);
    sh->names.resize((size_t)nc);
    static const int32_t zero_off[1] = {0};
    for (int c = 0; c < nc; c++) {
      const sqlrs_column_t &col = hb->columns[c];
      ArrowArray &a = h->children[(size_t)c];
      a = ArrowArray{};
      a.length = col.length;
      a.null_count = col.validity ? col.null_count : 0;
      a.offset = 0;
      h->bufs[(size_t)c] = {col.validity && col.null_count != 0 ? col.validity : nullptr, nullptr, nullptr};
      if (col.dtype == SQLRS_UTF8) {
        a.n_buffers = 3;
        h->bufs[(size_t)c][1] = col.offsets ? (const void *)col.offsets : (const void *)zero_off;
        h->bufs[(size_t)c][2] = col.values;
      } else {
        a.n_buffers = 2;
        h->bufs[(size_t)c][1] = col.values;
      }
      a.buffers = h->bufs[(size_t)c].data();
      a.release = export_child_release;
      a.private_data = h.get();
      h->child_ptrs[(size_t)c] = &a;
      ArrowSchema &s = sh->children[(size_t)c];
      s = ArrowSchema{};
      sh->names[(size_t)c] = (names && names[c]) ? names[c] : ("c" + std::to_string(c));
      s.format = arrow_format(col.dtype);
      s.name = sh->names[(size_t)c].c_str();
      s.flags = ARROW_FLAG_NULLABLE;
      s.release = export_schema_child_release;
      sh->child_ptrs[(size_t)c] = &s;
    }
    *out_array = ArrowArray{};
    out_array->length = hb->num_rows;
    out_array->n_buffers = 1;
    out_array->buffers = h->top_buf;
    out_array->n_children = nc;
    out_array->children = h->child_ptrs.data();
    out_array->release = export_parent_release;
    out_array->private_data = h.get();
    *out_schema = ArrowSchema{};
    out_schema->format = "+s";
    out_schema->name = "";
    out_schema->n_children = nc;
    out_schema->children = sh->child_ptrs.data();
    out_schema->release = export_schema_release;
    out_schema->private_data = sh.release();
    h->batch = hb;
    h->refs.store(nc + 1);
    h.release();
    if (hb != batch) sqlrs_batch_release(batch); // (moved either way: the caller's reference is gone)
  });
}

// ---- timers
struct sqlrs_timer {
  Ctx *ctx;
  hipEvent_t a, b;
};
int sqlrs_timer_create(sqlrs_ctx_t *ctx, sqlrs_timer_t **out) {
  return guard(ctx, [&] {
    auto *t = new sqlrs_timer();
    t->ctx = ctx;
    SQ_HIP(hipEventCreate(&t->a));
    SQ_HIP(hipEventCreate(&t->b));
    *out = t;
  });
}
int sqlrs_timer_start(sqlrs_timer_t *t) {
  return guard(t->ctx, [&] { SQ_HIP(hipEventRecord(t->a, t->ctx->stream)); });
}
int sqlrs_timer_stop(sqlrs_timer_t *t) {
  return guard(t->ctx, [&] { SQ_HIP(hipEventRecord(t->b, t->ctx->stream)); });
}
int sqlrs_timer_elapsed_ms(sqlrs_timer_t *t, double *ms) {
  return guard(t->ctx, [&] {
    SQ_HIP(hipEventSynchronize(t->b));
    float f = 0;
    SQ_HIP(hipEventElapsedTime(&f, t->a, t->b));
    *ms = f;
  });
}
void sqlrs_timer_destroy(sqlrs_timer_t *t) {
  if (!t) return;
  (void)hipEventDestroy(t->a);
  (void)hipEventDestroy(t->b);
  delete t;
}

int sqlrs_ctx_profile_enable(sqlrs_ctx_t *ctx, int on) {
  return guard(ctx, [&] {
    ctx->prof_resolve();
    ctx->prof_on = on != 0;
  });
}
int sqlrs_ctx_profile_reset(sqlrs_ctx_t *ctx) {
  return guard(ctx, [&] {
    ctx->prof_resolve();
    for (auto &e : ctx->prof) {
      e.total_ms = 0;
      e.launches = 0;
    }
  });
}
int sqlrs_ctx_profile_read(sqlrs_ctx_t *ctx, int cap, const char **names, double *total_ms,
                           int64_t *launches) {
  try {
    ctx->prof_resolve();
  } catch (...) {
    return 0;
  }
  int n = 0;
  for (auto &e : ctx->prof) {
    if (n < cap) {
      names[n] = e.name;
      total_ms[n] = e.total_ms;
      launches[n] = e.launches;
    }
    n++;
  }
  if (ctx->lb_timeouts) { // look-back launches redone with tickets since the ctx was created (0 ms: a count)
    if (n < cap) {
      names[n] = "lookback_ticket_reruns";
      total_ms[n] = 0;
      launches[n] = ctx->lb_timeouts;
    }
    n++;
  }
  if (ctx->async_fast_batches) { // (a count: batches of the async path that took its one-launch kernels)
    if (n < cap) {
      names[n] = "async_fast_batches";
      total_ms[n] = 0;
      launches[n] = ctx->async_fast_batches;
    }
    n++;
  }
  if (ctx->order_lb_fallbacks) { // (advisor r05: the process-wide switch-off of the Order look-back is visible, tests/conftest.py checks it)
    if (n < cap) {
      names[n] = "order_lookback_fallbacks";
      total_ms[n] = 0;
      launches[n] = ctx->order_lb_fallbacks;
    }
    n++;
  }
  return n;
}

const char *sqlrs_version(void) { return "sqlrs-hip 0.1.0 gfx950"; }

} // extern "C"
