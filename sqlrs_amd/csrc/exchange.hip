// exchange.hip — the multi-GPU exchange step behind the C ABI: an all-to-all of hash partitions over RCCL (xGMI),
// on the ctx stream, torch-free.  One process per GPU; every rank creates one sqlrs_exchange over a shared
// ncclUniqueId and calls sqlrs_exchange_all_to_all with the partitions sqlrs_hash_partition[_filter] produced.
//
// No reference analogue (sqlrs is a single process); the place it slots under is the executor the
// builder instantiates for a join / aggregate (src/executor/mod.rs:103-114,163-174): the children of a partitioned
// HashJoin / HashAgg are wrapped in an exchange of their hash partitions, everything above and below is unchanged.
//
// RCCL is loaded with dlopen at the first sqlrs_exchange_* call that needs it: the library itself has no link-time
// dependency on it (a single-GPU host needs none), and a missing librccl is SQLRS_ERR_DEVICE, not a load failure.
#include <dlfcn.h>

#include <rccl/rccl.h>

#include "common.hpp"
#include "prims.hpp"

using namespace sq;

namespace {

struct Rccl {
  void *lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl &rccl() {
  static Rccl r = [] {
    Rccl x;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      x.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (x.lib) break;
    }
    if (!x.lib) return x;
#define SQ_SYM(f) x.f = (decltype(x.f))dlsym(x.lib, "nccl" #f)
    SQ_SYM(GetUniqueId); SQ_SYM(CommInitRank); SQ_SYM(CommDestroy); SQ_SYM(AllGather); SQ_SYM(Send); SQ_SYM(Recv);
    SQ_SYM(GroupStart); SQ_SYM(GroupEnd); SQ_SYM(GetErrorString);
#undef SQ_SYM
    return x;
  }();
  if (!r.lib || !r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.Send || !r.Recv || !r.GroupStart ||
      !r.GroupEnd || !r.GetErrorString)
    fail(SQLRS_ERR_DEVICE, "exchange: librccl not found (dlopen librccl.so.1) or incomplete");
  return r;
}

#define SQ_NCCL(expr)                                                                               \
  do {                                                                                              \
    ncclResult_t _r = (expr);                                                                       \
    if (_r != ncclSuccess) fail(SQLRS_ERR_DEVICE, std::string(#expr) + ": " + rccl().GetErrorString(_r)); \
  } while (0)

} // namespace

extern "C" int sqlrs_exchange_plan(int world, int rank, const int64_t *send_rows_all, int64_t *recv_rows, int64_t *recv_start, int64_t *total);

namespace {
// validity travels as one BYTE per row: partitions start at arbitrary rows, a bitmap cannot be cut there
__global__ void xb_bits_to_bytes_kernel(const uint64_t *__restrict__ bits, int64_t n, uint8_t *__restrict__ bytes) {
  const int64_t i = blockIdx.x * 256ll + threadIdx.x;
  if (i < n) bytes[i] = (uint8_t)((bits[i >> 6] >> (i & 63)) & 1ull);
}
__global__ void xb_fill_kernel(uint8_t *__restrict__ p, int64_t n, uint8_t v) {
  const int64_t i = blockIdx.x * 256ll + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void xb_bytes_to_bits_kernel(const uint8_t *__restrict__ bytes, int64_t n, uint64_t *__restrict__ bits) {
  const int64_t i = blockIdx.x * 256ll + threadIdx.x; // (n rounded up to 64: whole waves)
  const uint64_t m = __ballot(i < n && bytes[i < n ? i : 0] != 0);
  if ((threadIdx.x & 63) == 0 && (i >> 6) < (n + 63) / 64) bits[i >> 6] = m;
}
constexpr int XB_MAX_COLS = 48;       // columns whose nullability the count words can carry
constexpr uint64_t XB_ERR = 1ull;     // flag word: bit 0 = this rank rejected its arguments, bits 8.. = column c has NULLs here
} // namespace

struct sqlrs_exchange {
  Ctx *ctx = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  int64_t bytes_sent_off_rank = 0, calls = 0;
  // count words of a chunk: [world rows per destination, flag word] per rank; two slots so that the words of chunk k + 1
  // are gathered (and copied to the host, asynchronously) while the payload of chunk k is sent
  BufP dsend[2], dall[2];
  int64_t *pin = nullptr; // pinned: per slot [world + 1 to send | world * (world + 1) gathered]
  hipEvent_t ev[2] = {nullptr, nullptr};
  int64_t *pin_send(int s) const { return pin + (size_t)s * ((size_t)(world + 1) * (size_t)(world + 1)); }
  int64_t *pin_all(int s) const { return pin_send(s) + (world + 1); }
  // chunk sequence (begin .. send_chunk* .. finish): ONE receive batch that grows
  struct Chunk {
    DBatch rows;                 // retained (shared when the batch is this library's, else a private copy)
    std::vector<BufP> vbytes;    // per column: its validity as bytes, or null
    std::vector<int64_t> start, cnt;
    uint64_t flags = 0;
    int slot = 0;
  };
  bool open = false, has_pending = false;
  Chunk pending;
  std::vector<int32_t> dtypes;
  std::vector<BufP> rbuf, rvalid; // received values per column; validity bytes per column (null until a NULL arrives)
  int64_t cap = 0, filled = 0, chunk_no = 0;
  ~sqlrs_exchange() {
    if (pin) (void)hipHostFree(pin);
    for (hipEvent_t e : ev)
      if (e) (void)hipEventDestroy(e);
  }
};

namespace {
// the count words of one chunk onto the wire: H2D out of the pinned slot, all-gather, D2H into the pinned slot, event —
// nothing here waits; wait_counts() does, when the words are needed
void gather_counts(sqlrs_exchange *x, int slot, const int64_t *part_rows, uint64_t flags) {
  Ctx *ctx = x->ctx;
  const int W = x->world;
  int64_t *ps = x->pin_send(slot);
  for (int p = 0; p < W; p++) ps[p] = part_rows ? part_rows[p] : 0;
  ps[W] = (int64_t)flags;
  SQ_HIP(hipMemcpyAsync(x->dsend[slot]->p, ps, 8 * (size_t)(W + 1), hipMemcpyHostToDevice, ctx->stream));
  SQ_NCCL(rccl().AllGather(x->dsend[slot]->p, x->dall[slot]->p, (size_t)(W + 1), ncclInt64, x->comm, ctx->stream));
  SQ_HIP(hipMemcpyAsync(x->pin_all(slot), x->dall[slot]->p, 8 * (size_t)W * (size_t)(W + 1), hipMemcpyDeviceToHost, ctx->stream));
  SQ_HIP(hipEventRecord(x->ev[slot], ctx->stream));
}
struct Counts {
  std::vector<int64_t> recv_rows, recv_start;
  int64_t total = 0;
  uint64_t flags_or = 0;
};
Counts wait_counts(sqlrs_exchange *x, int slot) {
  SQ_HIP(hipEventSynchronize(x->ev[slot]));
  const int W = x->world;
  const int64_t *all = x->pin_all(slot);
  std::vector<int64_t> matrix((size_t)W * W);
  Counts c;
  for (int q = 0; q < W; q++) {
    for (int p = 0; p < W; p++) matrix[(size_t)q * W + p] = all[(size_t)q * (W + 1) + p];
    c.flags_or |= (uint64_t)all[(size_t)q * (W + 1) + W];
  }
  // every rank sees the same flag words: a rank that rejected its arguments stops ALL of them here, not just itself
  if (c.flags_or & XB_ERR) fail(SQLRS_ERR_INTERNAL, "exchange: a rank rejected its arguments (column types / NULLs / partition bounds); nothing was sent");
  c.recv_rows.resize((size_t)W);
  c.recv_start.resize((size_t)W);
  if (sqlrs_exchange_plan(W, x->rank, matrix.data(), c.recv_rows.data(), c.recv_start.data(), &c.total) != SQLRS_OK)
    fail(SQLRS_ERR_INTERNAL, "exchange: inconsistent counts");
  return c;
}
// local checks; never throws before the ranks have met in the all-gather: problems become the ERR flag
uint64_t check_chunk(sqlrs_exchange *x, InBatch &ib, const int64_t *part_start, const int64_t *part_rows, const std::vector<int32_t> *dtypes) {
  uint64_t flags = 0;
  const int W = x->world, nc = ib.num_columns();
  if (!part_start || !part_rows) return XB_ERR;
  if (dtypes && (int)dtypes->size() != nc) flags |= XB_ERR;
  for (int c = 0; c < nc; c++) {
    const DCol &col = ib.col(c);
    if (!width_of(col.dtype) || col.stride == 0) flags |= XB_ERR; // fixed-width columns only (int32 / int64 / float64)
    if (dtypes && c < (int)dtypes->size() && (*dtypes)[(size_t)c] != col.dtype) flags |= XB_ERR;
    if (col.validity && col.null_count != 0) {
      if (c < XB_MAX_COLS) flags |= 1ull << (8 + c);
      else flags |= XB_ERR;
    }
  }
  for (int p = 0; p < W; p++)
    if (part_start[p] < 0 || part_rows[p] < 0 || part_start[p] + part_rows[p] > ib.rows()) flags |= XB_ERR;
  return flags;
}
// validity bytes of the columns that are nullable on ANY rank (mask): a rank whose column has no bitmap sends ones
std::vector<BufP> validity_bytes(Ctx *ctx, const DBatch &b, uint64_t mask) {
  std::vector<BufP> out(b.cols.size());
  for (size_t c = 0; c < b.cols.size() && c < (size_t)XB_MAX_COLS; c++) {
    if (!((mask >> (8 + c)) & 1)) continue;
    const DCol &col = b.cols[c];
    const int64_t n = std::max<int64_t>(b.rows, 1);
    out[c] = ctx->alloc((size_t)n);
    if (col.validity && col.null_count != 0)
      xb_bits_to_bytes_kernel<<<dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream>>>(col.validity, b.rows, out[c]->as<uint8_t>());
    else
      xb_fill_kernel<<<dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream>>>(out[c]->as<uint8_t>(), n, 1);
    SQ_HIP(hipGetLastError());
  }
  return out;
}
// the payload of one chunk: every column (values, then validity bytes) to and from every rank inside ONE group = one
// all-to-all-v of RCCL per chunk; the slice to this rank itself is a device copy inside the same group
void send_payload(sqlrs_exchange *x, const DBatch &rows, const std::vector<BufP> &vbytes, const std::vector<int64_t> &start,
                  const std::vector<int64_t> &cnt, const Counts &c, const std::vector<uint8_t *> &dst, const std::vector<uint8_t *> &vdst,
                  int64_t at) {
  Ctx *ctx = x->ctx;
  Rccl &R = rccl();
  const int W = x->world;
  SQ_NCCL(R.GroupStart());
  for (size_t ci = 0; ci < rows.cols.size(); ci++) {
    const size_t w = width_of(rows.cols[ci].dtype);
    const uint8_t *src = (const uint8_t *)rows.cols[ci].values;
    for (int p = 0; p < W; p++) {
      if (cnt[(size_t)p]) SQ_NCCL(R.Send(src + w * (size_t)start[(size_t)p], w * (size_t)cnt[(size_t)p], ncclUint8, p, x->comm, ctx->stream));
      if (c.recv_rows[(size_t)p])
        SQ_NCCL(R.Recv(dst[ci] + w * (size_t)(at + c.recv_start[(size_t)p]), w * (size_t)c.recv_rows[(size_t)p], ncclUint8, p, x->comm, ctx->stream));
      if (p != x->rank) x->bytes_sent_off_rank += (int64_t)(w * (size_t)cnt[(size_t)p]);
    }
    if (ci < vbytes.size() && vbytes[ci]) {
      const uint8_t *vs = vbytes[ci]->as<uint8_t>();
      for (int p = 0; p < W; p++) {
        if (cnt[(size_t)p]) SQ_NCCL(R.Send(vs + (size_t)start[(size_t)p], (size_t)cnt[(size_t)p], ncclUint8, p, x->comm, ctx->stream));
        if (c.recv_rows[(size_t)p]) SQ_NCCL(R.Recv(vdst[ci] + (size_t)(at + c.recv_start[(size_t)p]), (size_t)c.recv_rows[(size_t)p], ncclUint8, p, x->comm, ctx->stream));
        if (p != x->rank) x->bytes_sent_off_rank += cnt[(size_t)p];
      }
    }
  }
  SQ_NCCL(R.GroupEnd());
}
DCol finished_column(Ctx *ctx, int32_t dtype, BufP values, BufP vbytes, int64_t rows) {
  DCol d;
  d.dtype = dtype;
  d.length = rows;
  d.null_count = 0;
  d.own_values = values;
  d.values = values->p;
  if (vbytes && rows > 0) {
    d.own_validity = ctx->alloc(bitmap_bytes(rows) + 8);
    const int64_t n64 = (int64_t)round_up((size_t)rows, 64);
    xb_bytes_to_bits_kernel<<<dim3((unsigned)ceil_div(n64, 256)), dim3(256), 0, ctx->stream>>>(vbytes->as<uint8_t>(), rows, d.own_validity->as<uint64_t>());
    SQ_HIP(hipGetLastError());
    d.validity = d.own_validity->as<uint64_t>();
    d.null_count = -1; // (counted when someone needs it)
  }
  return d;
}
} // namespace

extern "C" {

// rows this rank receives from every rank (and where they start in the received batch) from the full count matrix:
// send_rows_all[q * world + p] = rows rank q sends to rank p.  Pure host arithmetic (no device, no RCCL): the CPU tests
// drive the same bookkeeping over gloo.
int sqlrs_exchange_plan(int world, int rank, const int64_t *send_rows_all, int64_t *recv_rows, int64_t *recv_start, int64_t *total) {
  if (world <= 0 || rank < 0 || rank >= world || !send_rows_all || !recv_rows) return SQLRS_ERR_INTERNAL;
  int64_t at = 0;
  for (int q = 0; q < world; q++) {
    const int64_t r = send_rows_all[(size_t)q * world + rank];
    if (r < 0) return SQLRS_ERR_INTERNAL;
    recv_rows[q] = r;
    if (recv_start) recv_start[q] = at;
    at += r;
  }
  if (total) *total = at;
  return SQLRS_OK;
}

int sqlrs_exchange_unique_id(sqlrs_ctx_t *ctx, void *id_out) {
  return guard(ctx, [&] {
    static_assert(SQLRS_EXCHANGE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    ncclUniqueId id;
    SQ_NCCL(rccl().GetUniqueId(&id));
    std::memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
  });
}

int sqlrs_exchange_create(sqlrs_ctx_t *ctx, const void *unique_id, int rank, int world, sqlrs_exchange_t **out) {
  return guard(ctx, [&] {
    if (world <= 0 || rank < 0 || rank >= world || !unique_id) fail(SQLRS_ERR_INTERNAL, "exchange: bad rank / world / id");
    SQ_HIP(hipSetDevice(ctx->device));
    auto x = std::unique_ptr<sqlrs_exchange>(new sqlrs_exchange());
    x->ctx = ctx;
    x->rank = rank;
    x->world = world;
    ncclUniqueId id;
    std::memcpy(id.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
    SQ_NCCL(rccl().CommInitRank(&x->comm, world, id, rank)); // (collective: every rank of the id calls it)
    for (int s2 = 0; s2 < 2; s2++) {
      x->dsend[s2] = ctx->alloc(8 * ((size_t)world + 1));
      x->dall[s2] = ctx->alloc(8 * (size_t)world * ((size_t)world + 1));
      SQ_HIP(hipEventCreateWithFlags(&x->ev[s2], hipEventDisableTiming));
    }
    SQ_HIP(hipHostMalloc((void **)&x->pin, 8 * 2 * (size_t)(world + 1) * (size_t)(world + 1), hipHostMallocDefault));
    *out = x.release();
  });
}

int sqlrs_exchange_all_to_all(sqlrs_exchange_t *x, const sqlrs_batch_t *in, const int64_t *part_start, const int64_t *part_rows,
                              sqlrs_batch_t **out, int64_t *recv_rows_out) {
  if (out) *out = nullptr;
  return guard(x->ctx, [&] {
    Ctx *ctx = x->ctx;
    SQ_HIP(hipSetDevice(ctx->device));
    const int W = x->world;
    // (advisor r05) EVERY local failure ahead of the collectives — a null argument, an open chunk sequence, a batch the
    // entrance rejects, a column that cannot be brought to the device — becomes the ERR flag of this rank's count words:
    // the ranks then fail TOGETHER in wait_counts instead of this one returning while its peers sit in the all-gather
    std::unique_ptr<InBatch> ibp;
    DBatch rows;
    uint64_t flags = 0;
    std::string why;
    try {
      if (!in || !out) fail(SQLRS_ERR_INTERNAL, "exchange: null argument");
      if (x->open) fail(SQLRS_ERR_INTERNAL, "exchange: a chunk sequence is open (sqlrs_exchange_finish it first)");
      ibp.reset(new InBatch(ctx, in));
      flags = check_chunk(x, *ibp, part_start, part_rows, nullptr);
      if (!(flags & XB_ERR)) rows = ibp->materialize(false); // (before the count words go out: it may throw)
    } catch (const Error &e) {
      flags = XB_ERR;
      why = e.msg;
    }
    // 1. who sends how much to whom — and whether every rank accepted its arguments: ONE all-gather, one fetch
    gather_counts(x, 0, (flags & XB_ERR) ? nullptr : part_rows, flags);
    Counts c;
    try {
      c = wait_counts(x, 0); // (throws on every rank alike when one of them raised ERR)
    } catch (const Error &e) {
      if (!why.empty()) fail(e.status, e.msg + " [this rank: " + why + "]");
      throw;
    }
    InBatch &ib = *ibp;
    // 2. the payload
    std::vector<BufP> vb = validity_bytes(ctx, rows, c.flags_or);
    const int nc = ib.num_columns();
    std::vector<BufP> vals((size_t)nc), rv((size_t)nc);
    std::vector<uint8_t *> dst((size_t)nc), vdst((size_t)nc, nullptr);
    for (int ci = 0; ci < nc; ci++) {
      vals[(size_t)ci] = ctx->alloc(width_of(rows.cols[(size_t)ci].dtype) * (size_t)std::max<int64_t>(c.total, 1));
      dst[(size_t)ci] = vals[(size_t)ci]->as<uint8_t>();
      if (vb[(size_t)ci]) {
        rv[(size_t)ci] = ctx->alloc((size_t)std::max<int64_t>(c.total, 1));
        vdst[(size_t)ci] = rv[(size_t)ci]->as<uint8_t>();
      }
    }
    std::vector<int64_t> st(part_start, part_start + W), cn(part_rows, part_rows + W);
    send_payload(x, rows, vb, st, cn, c, dst, vdst, 0);
    DBatch o;
    o.rows = c.total;
    for (int ci = 0; ci < nc; ci++) o.cols.push_back(finished_column(ctx, rows.cols[(size_t)ci].dtype, vals[(size_t)ci], rv[(size_t)ci], c.total));
    x->calls++;
    if (recv_rows_out) std::memcpy(recv_rows_out, c.recv_rows.data(), 8 * (size_t)W);
    *out = emit_batch(ctx, std::move(o), SQLRS_MEM_DEVICE);
  });
}

// ---- a SEQUENCE of chunks into one receive batch (the fact rows of the partitioned join, chunk by chunk) -------------
// send_chunk(k) puts the count words of chunk k on the wire and sends the payload of chunk k - 1, whose words have
// arrived on the host long since: the host never blocks on the device inside the loop (one event wait that is already
// complete), chunk k's partitioning overlaps chunk k - 1's transfer on the stream, and finish() sends the last chunk.
int sqlrs_exchange_begin(sqlrs_exchange_t *x, int num_columns, const int32_t *dtypes, int64_t capacity_rows) {
  return guard(x->ctx, [&] {
    if (x->open) fail(SQLRS_ERR_INTERNAL, "exchange: a chunk sequence is already open");
    if (num_columns <= 0 || !dtypes) fail(SQLRS_ERR_INTERNAL, "exchange_begin: columns?");
    Ctx *ctx = x->ctx;
    SQ_HIP(hipSetDevice(ctx->device));
    x->dtypes.assign(dtypes, dtypes + num_columns);
    for (int32_t dt : x->dtypes)
      if (!width_of(dt)) fail(SQLRS_ERR_INTERNAL, "exchange: fixed-width columns only (int32 / int64 / float64)");
    x->cap = std::max<int64_t>(capacity_rows, 1024);
    x->rbuf.assign((size_t)num_columns, nullptr);
    x->rvalid.assign((size_t)num_columns, nullptr);
    for (int c = 0; c < num_columns; c++) x->rbuf[(size_t)c] = ctx->alloc(width_of(x->dtypes[(size_t)c]) * (size_t)x->cap);
    x->filled = 0;
    x->chunk_no = 0;
    x->has_pending = false;
    x->open = true;
  });
}

static void flush_pending(sqlrs_exchange *x) { // payload of the pending chunk (its count words have been gathered)
  if (!x->has_pending) return;
  Ctx *ctx = x->ctx;
  sqlrs_exchange::Chunk ch = std::move(x->pending);
  x->has_pending = false;
  x->pending = sqlrs_exchange::Chunk();
  Counts c = wait_counts(x, ch.slot);
  const size_t nc = x->dtypes.size();
  if (x->filled + c.total > x->cap) { // grow: new blocks, what has arrived so far copied over (stream ordered)
    const int64_t ncap = std::max<int64_t>(x->filled + c.total, 2 * x->cap);
    for (size_t ci = 0; ci < nc; ci++) {
      const size_t w = width_of(x->dtypes[ci]);
      BufP nb = ctx->alloc(w * (size_t)ncap);
      if (x->filled) SQ_HIP(hipMemcpyAsync(nb->p, x->rbuf[ci]->p, w * (size_t)x->filled, hipMemcpyDeviceToDevice, ctx->stream));
      x->rbuf[ci] = nb;
      if (x->rvalid[ci]) {
        BufP nv = ctx->alloc((size_t)ncap);
        if (x->filled) SQ_HIP(hipMemcpyAsync(nv->p, x->rvalid[ci]->p, (size_t)x->filled, hipMemcpyDeviceToDevice, ctx->stream));
        x->rvalid[ci] = nv;
      }
    }
    x->cap = ncap;
  }
  std::vector<BufP> vb = validity_bytes(ctx, ch.rows, c.flags_or);
  std::vector<uint8_t *> dst(nc), vdst(nc, nullptr);
  for (size_t ci = 0; ci < nc; ci++) {
    dst[ci] = x->rbuf[ci]->as<uint8_t>();
    if (vb[ci]) {
      if (!x->rvalid[ci]) { // the first NULL of this column anywhere: the rows received so far are all valid
        x->rvalid[ci] = ctx->alloc((size_t)x->cap);
        if (x->filled) {
          xb_fill_kernel<<<dim3((unsigned)ceil_div(x->filled, 256)), dim3(256), 0, ctx->stream>>>(x->rvalid[ci]->as<uint8_t>(), x->filled, 1);
          SQ_HIP(hipGetLastError());
        }
      }
      vdst[ci] = x->rvalid[ci]->as<uint8_t>();
    } else if (x->rvalid[ci] && c.total) { // nullable earlier, no NULL on any rank in this chunk: ones
      xb_fill_kernel<<<dim3((unsigned)ceil_div(c.total, 256)), dim3(256), 0, ctx->stream>>>(x->rvalid[ci]->as<uint8_t>() + x->filled, c.total, 1);
      SQ_HIP(hipGetLastError());
    }
  }
  send_payload(x, ch.rows, vb, ch.start, ch.cnt, c, dst, vdst, x->filled);
  x->filled += c.total;
  x->calls++;
}

int sqlrs_exchange_send_chunk(sqlrs_exchange_t *x, const sqlrs_batch_t *in, const int64_t *part_start, const int64_t *part_rows) {
  return guard(x->ctx, [&] {
    Ctx *ctx = x->ctx;
    SQ_HIP(hipSetDevice(ctx->device));
    if (!x->open) fail(SQLRS_ERR_INTERNAL, "exchange: send_chunk without begin");
    const int W = x->world;
    // (advisor r05: local failures become the ERR flag, see sqlrs_exchange_all_to_all; the rows are materialised BEFORE the
    //  chunk's count words are queued, so nothing can throw between this rank's all-gather and its peers')
    sqlrs_exchange::Chunk ch;
    uint64_t flags = 0;
    try {
      if (!in) fail(SQLRS_ERR_INTERNAL, "exchange: null argument");
      InBatch ib(ctx, in);
      flags = check_chunk(x, ib, part_start, part_rows, &x->dtypes);
      if (!(flags & XB_ERR)) ch.rows = ib.materialize(true); // retained until its payload is queued (the library's own batches are shared, not copied)
    } catch (const Error &) {
      flags = XB_ERR;
    }
    const int slot = (int)(x->chunk_no & 1);
    // (slot `slot` was last used by chunk_no - 2, whose payload went out in the previous call: its words have been read)
    gather_counts(x, slot, (flags & XB_ERR) ? nullptr : part_rows, flags);
    ch.slot = slot;
    ch.flags = flags;
    if (!(flags & XB_ERR)) {
      ch.start.assign(part_start, part_start + W);
      ch.cnt.assign(part_rows, part_rows + W);
    } else {
      ch.start.assign((size_t)W, 0);
      ch.cnt.assign((size_t)W, 0);
    }
    x->chunk_no++;
    try {
      flush_pending(x); // the previous chunk's payload, behind this chunk's count words on the stream
    } catch (...) {
      x->open = false; // (every rank fails in the same call: the sequence is over)
      x->has_pending = false;
      throw;
    }
    x->pending = std::move(ch);
    x->has_pending = true;
  });
}

int sqlrs_exchange_finish(sqlrs_exchange_t *x, sqlrs_batch_t **out) {
  if (out) *out = nullptr;
  return guard(x->ctx, [&] {
    Ctx *ctx = x->ctx;
    SQ_HIP(hipSetDevice(ctx->device));
    if (!x->open || !out) fail(SQLRS_ERR_INTERNAL, "exchange: finish without begin");
    x->open = false;
    flush_pending(x);
    DBatch o;
    o.rows = x->filled;
    for (size_t ci = 0; ci < x->dtypes.size(); ci++)
      o.cols.push_back(finished_column(ctx, x->dtypes[ci], x->rbuf[ci], x->rvalid[ci], x->filled));
    x->rbuf.clear();
    x->rvalid.clear();
    *out = emit_batch(ctx, std::move(o), SQLRS_MEM_DEVICE);
  });
}

int64_t sqlrs_exchange_bytes_off_rank(const sqlrs_exchange_t *x) { return x->bytes_sent_off_rank; }

void sqlrs_exchange_destroy(sqlrs_exchange_t *x) {
  if (!x) return;
  if (x->comm) {
    (void)hipSetDevice(x->ctx->device);
    (void)hipStreamSynchronize(x->ctx->stream);
    rccl().CommDestroy(x->comm);
  }
  delete x;
}

} // extern "C"
