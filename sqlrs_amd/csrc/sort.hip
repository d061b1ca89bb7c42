// sort.hip — stable LSD radix sort of (u64 key, u32 row id) pairs, 8 bits per pass.
//
// Used by OrderExecutor (order.rs:45 lexsort_to_indices), by the agg finalize (groups in
// first-seen order, hash_agg.rs:98,132) and by the join build when a key repeats (rows of one
// key in insertion order, hash_join.rs:172-177).  Per pass: per-tile digit histogram ->
// exclusive scan over (digit, tile) -> stable scatter.  HBM traffic per pass:
// hist reads 8 B/key, scatter reads 12 B and writes 12 B per key.
#include "common.hpp"
#include "device_utils.hpp"
#include "prims.hpp"

namespace sq {

constexpr int RS_ITEMS = 16;
constexpr int RS_TILE = BLOCK * RS_ITEMS; // 4096 keys per block

__global__ __launch_bounds__(BLOCK) void rs_hist_kernel(const uint64_t *__restrict__ keys, int64_t n,
                                                        int shift, int64_t nblocks,
                                                        uint32_t *__restrict__ hist) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll 4
  for (int r = 0; r < RS_ITEMS; r++) {
    int64_t i = base + r * BLOCK + threadIdx.x;
    if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255], 1u);
  }
  __syncthreads();
  hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(BLOCK) void rs_scatter_kernel(
    const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals, int64_t n, int shift,
    int64_t nblocks, const uint32_t *__restrict__ offsets, uint64_t *__restrict__ keys_out,
    uint32_t *__restrict__ vals_out) {
  __shared__ uint32_t base[256];
  __shared__ uint32_t wcnt[WAVES_PER_BLOCK][256];
  base[threadIdx.x] = offsets[(int64_t)threadIdx.x * nblocks + blockIdx.x];
#pragma unroll
  for (int w = 0; w < WAVES_PER_BLOCK; w++) wcnt[w][threadIdx.x] = 0;
  __syncthreads();
  const int w = wave_id();
  const int64_t tbase = (int64_t)blockIdx.x * RS_TILE;
  for (int r = 0; r < RS_ITEMS; r++) {
    int64_t i = tbase + r * BLOCK + threadIdx.x;
    bool valid = i < n;
    uint64_t k = valid ? keys[i] : 0;
    uint32_t v = valid ? vals[i] : 0;
    uint32_t d = (uint32_t)(k >> shift) & 255u;
    // lanes of this wave holding the same digit
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      bool bit = (d >> b) & 1;
      uint64_t bm = __ballot(bit);
      peers &= bit ? bm : ~bm;
    }
    uint32_t rank = (uint32_t)mbcnt(peers);
    uint32_t cnt = (uint32_t)__popcll(peers);
    bool leader = valid && rank == 0;
    if (leader) wcnt[w][d] = cnt;
    __syncthreads();
    uint32_t before = 0, total = 0;
    if (valid) {
#pragma unroll
      for (int q = 0; q < WAVES_PER_BLOCK; q++) {
        uint32_t c = wcnt[q][d];
        before += (q < w) ? c : 0;
        total += c;
      }
    }
    uint32_t pos = valid ? base[d] + before + rank : 0;
    __syncthreads();
    if (leader) {
      wcnt[w][d] = 0;
      if (before == 0) base[d] += total; // the lowest wave holding this digit advances it
    }
    if (valid) {
      keys_out[pos] = k;
      vals_out[pos] = v;
    }
    __syncthreads();
  }
}

void radix_sort_pairs(Ctx *ctx, uint64_t *keys, uint32_t *vals, int64_t n, int begin_bit,
                      int end_bit) {
  if (n <= 1 || end_bit <= begin_bit) return;
  if (n > 0xffffffffll) fail(SQLRS_ERR_INTERNAL, "radix sort: more than 2^32 rows");
  ProfScope ps(ctx, "radix_sort");
  int64_t nblocks = ceil_div(n, RS_TILE);
  BufP k2 = ctx->alloc(8 * (size_t)n), v2 = ctx->alloc(4 * (size_t)n);
  BufP hist = ctx->alloc(4 * (size_t)(256 * nblocks)), offs = ctx->alloc(4 * (size_t)(256 * nblocks));
  BufP total = ctx->alloc(8);
  uint64_t *ka = keys, *kb = k2->as<uint64_t>();
  uint32_t *va = vals, *vb = v2->as<uint32_t>();
  for (int shift = begin_bit; shift < end_bit; shift += 8) {
    rs_hist_kernel<<<dim3((unsigned)nblocks), dim3(BLOCK), 0, ctx->stream>>>(ka, n, shift, nblocks,
                                                                            hist->as<uint32_t>());
    SQ_HIP(hipGetLastError());
    exclusive_scan_u32(ctx, hist->as<uint32_t>(), 256 * nblocks, nullptr, offs->as<uint32_t>(),
                       total->as<uint64_t>());
    rs_scatter_kernel<<<dim3((unsigned)nblocks), dim3(BLOCK), 0, ctx->stream>>>(
        ka, va, n, shift, nblocks, offs->as<uint32_t>(), kb, vb);
    SQ_HIP(hipGetLastError());
    std::swap(ka, kb);
    std::swap(va, vb);
  }
  if (ka != keys) {
    SQ_HIP(hipMemcpyAsync(keys, ka, 8 * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
    SQ_HIP(hipMemcpyAsync(vals, va, 4 * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
  }
}

} // namespace sq
