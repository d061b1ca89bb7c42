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

struct sqlrs_exchange {
  Ctx *ctx = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  BufP counts_dev; // [world] send counts + [world * world] gathered
  int64_t bytes_sent_off_rank = 0, calls = 0;
};

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
    x->counts_dev = ctx->alloc(8 * ((size_t)world + (size_t)world * world));
    *out = x.release();
  });
}

int sqlrs_exchange_all_to_all(sqlrs_exchange_t *x, const sqlrs_batch_t *in, const int64_t *part_start, const int64_t *part_rows,
                              sqlrs_batch_t **out, int64_t *recv_rows_out) {
  return guard(x->ctx, [&] {
    Ctx *ctx = x->ctx;
    Rccl &R = rccl();
    SQ_HIP(hipSetDevice(ctx->device));
    const int W = x->world;
    if (!in || !part_start || !part_rows) fail(SQLRS_ERR_INTERNAL, "exchange: null argument");
    InBatch ib(ctx, in);
    const int nc = ib.num_columns();
    std::vector<const DCol *> cols;
    std::vector<size_t> width;
    for (int c = 0; c < nc; c++) {
      const DCol &col = ib.col(c);
      const size_t w = width_of(col.dtype);
      if (!w || col.stride == 0) fail(SQLRS_ERR_INTERNAL, "exchange: fixed-width columns only (int32 / int64 / float64)");
      if (col.validity && col.null_count != 0) fail(SQLRS_ERR_INTERNAL, "exchange: columns with NULLs are not supported");
      cols.push_back(&col);
      width.push_back(w);
    }
    for (int p = 0; p < W; p++)
      if (part_start[p] < 0 || part_rows[p] < 0 || part_start[p] + part_rows[p] > ib.rows())
        fail(SQLRS_ERR_INTERNAL, "exchange: partition outside the batch");
    // 1. who sends how much to whom: all-gather of the send counts, then one fetch
    int64_t *dsend = x->counts_dev->as<int64_t>(), *dall = dsend + W;
    SQ_HIP(hipMemcpyAsync(dsend, part_rows, 8 * (size_t)W, hipMemcpyHostToDevice, ctx->stream));
    SQ_NCCL(R.AllGather(dsend, dall, (size_t)W, ncclInt64, x->comm, ctx->stream));
    std::vector<int64_t> all((size_t)W * W), recv_rows((size_t)W), recv_start((size_t)W);
    SQ_HIP(hipMemcpyAsync(all.data(), dall, 8 * all.size(), hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync(); // (also: `part_rows` has been read)
    int64_t total = 0;
    if (sqlrs_exchange_plan(W, x->rank, all.data(), recv_rows.data(), recv_start.data(), &total) != SQLRS_OK)
      fail(SQLRS_ERR_INTERNAL, "exchange: inconsistent counts");
    // 2. the payload: per column one grouped send / recv to and from every rank (ncclSend / ncclRecv inside one group
    //    = RCCL's all-to-all-v), bytes as ncclUint8; the slice to this rank itself is a device copy inside the group too
    DBatch o;
    o.rows = total;
    for (int c = 0; c < nc; c++) {
      DCol d;
      d.dtype = cols[(size_t)c]->dtype;
      d.length = total;
      d.null_count = 0;
      d.own_values = ctx->alloc(width[(size_t)c] * (size_t)std::max<int64_t>(total, 1));
      d.values = d.own_values->p;
      o.cols.push_back(std::move(d));
    }
    for (int c = 0; c < nc; c++) {
      const size_t w = width[(size_t)c];
      const uint8_t *src = (const uint8_t *)cols[(size_t)c]->values;
      uint8_t *dst = (uint8_t *)o.cols[(size_t)c].own_values->p;
      SQ_NCCL(R.GroupStart());
      for (int p = 0; p < W; p++) {
        if (part_rows[p]) SQ_NCCL(R.Send(src + w * (size_t)part_start[p], w * (size_t)part_rows[p], ncclUint8, p, x->comm, ctx->stream));
        if (recv_rows[(size_t)p]) SQ_NCCL(R.Recv(dst + w * (size_t)recv_start[(size_t)p], w * (size_t)recv_rows[(size_t)p], ncclUint8, p, x->comm, ctx->stream));
        if (p != x->rank) x->bytes_sent_off_rank += (int64_t)(w * (size_t)part_rows[p]);
      }
      SQ_NCCL(R.GroupEnd());
    }
    x->calls++;
    if (recv_rows_out) std::memcpy(recv_rows_out, recv_rows.data(), 8 * (size_t)W);
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
