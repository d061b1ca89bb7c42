// sort.hip — stable LSD radix sort of (u64 key, u32 row id) pairs, 8 bits per pass.
//
// Used by OrderExecutor (order.rs:45 lexsort_to_indices), by the agg finalize (groups in
// first-seen order, hash_agg.rs:98,132) and by the join build when a key repeats (rows of one
// key in insertion order, hash_join.rs:172-177).  Per pass: per-tile digit histogram ->
// exclusive scan over (digit, tile) -> stable LDS-staged scatter.  HBM traffic per pass:
// hist reads 8 B/key, scatter reads 12 B and writes 12 B per key.
#include "common.hpp"
#include "device_utils.hpp"
#include "prims.hpp"

namespace sq {

// Tile = 512 threads x RS_ITEMS rows.  Row order inside a tile is (wave, chunk, lane): wave w owns
// the RS_ITEMS * 64 consecutive rows [w * RS_ITEMS * 64, ...), chunk j of it is 64 consecutive rows.
constexpr int RS_WG = 512;
constexpr int RS_WAVES = RS_WG / 64;
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = RS_WG * RS_ITEMS; // 4096 keys per block: 56 KiB of LDS, two blocks per CU

// KEY32: `keys` is an array of u32 keys (the first pass of radix_sort_index_u32)
template <bool KEY32 = false>
__global__ __launch_bounds__(RS_WG) void rs_hist_kernel(const uint64_t *__restrict__ keys, int64_t n,
                                                        int shift, int64_t nblocks,
                                                        uint32_t *__restrict__ hist) {
  __shared__ uint32_t h[256];
  if (threadIdx.x < 256) h[threadIdx.x] = 0;
  const int64_t base = (int64_t)blockIdx.x * RS_TILE + threadIdx.x;
  uint64_t k[RS_ITEMS];
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++)
    k[r] = KEY32 ? (uint64_t)__builtin_nontemporal_load((const uint32_t *)keys + min(base + r * RS_WG, n - 1))
                 : __builtin_nontemporal_load(keys + min(base + r * RS_WG, n - 1));
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++)
    if (base + r * RS_WG < n) atomicAdd(&h[(k[r] >> shift) & 255], 1u);
  __syncthreads();
  if (threadIdx.x < 256) hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// Stable scatter of one tile, staged through LDS so that every digit's rows leave as one
// contiguous run (direct per-row stores ran at the random-store rate: 0.8 TB/s per pass).
//   1. per wave, chunk by chunk: lanes with the same digit find each other with 8 ballots; the
//      first of them bumps the wave's own counter of that digit (no atomics: one writer per digit
//      and chunk, chunks in order) -> rank of the row among the wave's rows of that digit
//   2. per digit: exclusive prefix of the 8 wave counters, exclusive scan over the 256 digits
//   3. rows are written to their tile-local position in LDS (digit-major, row order inside a digit)
//   4. the tile is copied out; position p of digit d goes to offsets[d][tile] + (p - start[d])
// Record formats.  PAIR: (u64 key, u32 row id) in two arrays.  PACKED: one u64 word
// (key_part << 32) | row id, used between the first and the last pass when the varying key bits
// fit 32 bits: 8 instead of 12 bytes per row and pass.  IN / OUT select what a pass reads / writes;
// a PACKED -> PAIR pass rebuilds the key as (word >> 32) << key_lo | key_const.
// KEY32 (with IN = RS_PAIR): `keys` is an array of u32 keys and a row's value is its index (no `vals` array): the first
// pass of radix_sort_index_u32.  keys_out == nullptr (OUT = RS_PAIR): only the values are written (its last pass).
enum { RS_PAIR = 0, RS_PACKED = 1 };
template <int IN, int OUT, bool KEY32 = false>
__global__ __launch_bounds__(RS_WG) void rs_scatter_kernel(
    const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals, int64_t n, int shift,
    int64_t nblocks, const uint32_t *__restrict__ offsets, uint64_t *__restrict__ keys_out,
    uint32_t *__restrict__ vals_out, int key_lo, uint64_t key_const) {
  __shared__ uint64_t skey[RS_TILE];
  __shared__ uint32_t sval[(IN == RS_PAIR && OUT == RS_PAIR) ? RS_TILE : 1];
  __shared__ uint32_t wcnt[RS_WAVES][256];
  __shared__ uint32_t dstart[256];
  __shared__ int64_t gbase[256];
  __shared__ uint32_t s_wsum[4];
  const int w = wave_id(), lane = lane_id();
  const int64_t tbase = (int64_t)blockIdx.x * RS_TILE;
  const int64_t wrow = tbase + (int64_t)w * (RS_ITEMS * 64) + lane;
  uint64_t k[RS_ITEMS];
  uint32_t v[RS_ITEMS];
#pragma unroll
  for (int j = 0; j < RS_ITEMS; j++) {
    const int64_t i = min(wrow + j * 64, n - 1);
    k[j] = KEY32 ? (uint64_t)__builtin_nontemporal_load((const uint32_t *)keys + i) : __builtin_nontemporal_load(keys + i);
    v[j] = KEY32 ? (uint32_t)i : (IN == RS_PAIR ? __builtin_nontemporal_load(vals + i) : 0u);
  }
  uint32_t goff = threadIdx.x < 256 ? offsets[(int64_t)threadIdx.x * nblocks + blockIdx.x] : 0;
#pragma unroll
  for (int q = 0; q < 4; q++) wcnt[w][lane + 64 * q] = 0;
  uint32_t rnk[RS_ITEMS];
#pragma unroll
  for (int j = 0; j < RS_ITEMS; j++) {
    const bool valid = wrow + j * 64 < n;
    const uint32_t d = (uint32_t)(k[j] >> shift) & 255u;
    uint64_t peers = __ballot(valid);
    // (lanes whose digit differs from this lane's in some bit, accumulated — order_fast.hip, stable_wave_ranks: a signed bit-field
    //  extract, one compare, two XORs and two ORs per bit instead of a select between the ballot and its complement)
    uint32_t dlo = 0, dhi = 0;
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const int32_t mask = ((int32_t)(d << (31 - b))) >> 31; // 0 or -1
      const uint64_t bm = __ballot(mask != 0);
      dlo |= (uint32_t)bm ^ (uint32_t)mask;
      dhi |= (uint32_t)(bm >> 32) ^ (uint32_t)mask;
    }
    peers &= ~(((uint64_t)dhi << 32) | dlo);
    const uint32_t r = (uint32_t)mbcnt(peers);
    uint32_t old = 0;
    if (valid && r == 0) { // first lane of the digit in this chunk
      old = wcnt[w][d];
      wcnt[w][d] = old + (uint32_t)__popcll(peers);
    }
    old = (uint32_t)__shfl((int)old, valid ? __builtin_ctzll(peers) : 0, 64);
    rnk[j] = old + r;
  }
  __syncthreads();
  if (threadIdx.x < 256) { // digit d = threadIdx.x: wave counters -> exclusive prefix over the waves
    uint32_t acc = 0;
#pragma unroll
    for (int q = 0; q < RS_WAVES; q++) {
      uint32_t c = wcnt[q][threadIdx.x];
      wcnt[q][threadIdx.x] = acc;
      acc += c;
    }
    uint32_t inc = wave_iscan_u32(acc);
    if (lane == 63) s_wsum[w] = inc;
    dstart[threadIdx.x] = inc - acc; // wave-local for now
  }
  __syncthreads();
  if (threadIdx.x < 256) {
    uint32_t wb = 0;
    for (int q = 0; q < w; q++) wb += s_wsum[q];
    uint32_t ds = dstart[threadIdx.x] + wb;
    dstart[threadIdx.x] = ds;
    gbase[threadIdx.x] = (int64_t)goff - (int64_t)ds;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < RS_ITEMS; j++) {
    if (wrow + j * 64 >= n) continue;
    const uint32_t d = (uint32_t)(k[j] >> shift) & 255u;
    const uint32_t p = dstart[d] + wcnt[w][d] + rnk[j];
    // a PAIR -> PACKED pass packs here: the digit was taken from the unpacked key at `shift`,
    // later passes take it from the word at shift - key_lo + 32
    if (IN == RS_PAIR && OUT == RS_PACKED) skey[p] = ((k[j] >> key_lo) << 32) | v[j];
    else skey[p] = k[j];
    if (IN == RS_PAIR && OUT == RS_PAIR) sval[p] = v[j];
  }
  __syncthreads();
  const uint32_t len = (uint32_t)min<int64_t>(RS_TILE, n - tbase);
#pragma unroll
  for (int j = 0; j < RS_ITEMS; j++) {
    const uint32_t p = j * RS_WG + threadIdx.x;
    if (p < len) {
      const uint64_t kk = skey[p];
      // digit of the staged record: a freshly packed word carries the key at bit 32 - key_lo
      const int sh = (IN == RS_PAIR && OUT == RS_PACKED) ? shift - key_lo + 32 : shift;
      const int64_t g = gbase[(uint32_t)(kk >> sh) & 255u] + p;
      if (OUT == RS_PACKED) {
        keys_out[g] = kk;
      } else if (IN == RS_PACKED) {
        if (keys_out) keys_out[g] = ((kk >> 32) << key_lo) | key_const;
        vals_out[g] = (uint32_t)kk;
      } else {
        if (keys_out) keys_out[g] = kk;
        vals_out[g] = sval[p];
      }
    }
  }
}

// OR of (key ^ key[0]) over all rows: the bit positions on which the keys differ at all
__global__ __launch_bounds__(256) void rs_diff_kernel(const uint64_t *__restrict__ keys, int64_t n,
                                                      unsigned long long *diff_or) {
  const uint64_t k0 = keys[0];
  uint64_t d = 0;
  for (int64_t i = blockIdx.x * (int64_t)(256 * 8) + threadIdx.x; i < n; i += (int64_t)gridDim.x * (256 * 8)) {
    uint64_t k[8];
#pragma unroll
    for (int u = 0; u < 8; u++) k[u] = keys[min(i + u * 256, n - 1)];
#pragma unroll
    for (int u = 0; u < 8; u++) d |= k[u] ^ k0;
  }
  for (int m = 32; m >= 1; m >>= 1) d |= shfl_xor_u64(d, m);
  // one atomic per block: thousands of atomics on one address cost more than the read of the keys
  __shared__ unsigned long long s_d[4];
  if (lane_id() == 0) s_d[wave_id()] = d;
  __syncthreads();
  if (threadIdx.x == 0) {
    d = s_d[0] | s_d[1] | s_d[2] | s_d[3];
    if (d) atomicOr(diff_or, (unsigned long long)d);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) diff_or[1] = k0; // the bits all keys share are read off any key
}

void radix_sort_pairs(Ctx *ctx, uint64_t *keys, uint32_t *vals, int64_t n, int begin_bit,
                      int end_bit, bool keys_below_end_bit) {
  if (n <= 1 || end_bit <= begin_bit) return;
  if (n > 0xffffffffll) fail(SQLRS_ERR_INTERNAL, "radix sort: more than 2^32 rows");
  ProfScope ps(ctx, "radix_sort");
  // Passes over bytes on which no two keys differ would be the identity (stable): skip them.
  // One extra read of the keys (8 B/row) against 32 B/row per skipped pass; worth asking when
  // more than two passes are requested (int64 sort keys of a 31-bit column: 8 passes -> 4).
  uint64_t varying = ~0ull, key0 = 0;
  // (below 2^22 rows a pass is launch-bound, ~35 us: the same as this question — kernel + host round trip — costs,
  //  and callers that know their keys' width pass it as end_bit)
  if (keys_below_end_bit && begin_bit == 0 && end_bit <= 32) {
    varying = (1ull << end_bit) - 1; // (all of it sorted, nothing above it: the packed form applies without asking)
    key0 = 0;
  } else if ((end_bit - begin_bit > 16 && n >= (1 << 22)) || (end_bit - begin_bit > 32 && n >= (1 << 15))) {
    // (more than four passes requested of a column that is small enough for its passes to be launch-bound, ~23 us each: the
    //  question costs about two passes and an int64 column of 31-bit values answers "four of the eight" — ORDER BY ... LIMIT's
    //  sample and candidate sorts 0.19 -> ms each)
    BufP diff = ctx->alloc_zero(16);
    unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(n, 256 * 8), 4 * (int64_t)ctx->num_cus);
    rs_diff_kernel<<<dim3(blocks), dim3(256), 0, ctx->stream>>>(keys, n, diff->as<unsigned long long>());
    SQ_HIP(hipGetLastError());
    const uint64_t *h = (const uint64_t *)ctx->fetch(diff->p, 16);
    varying = h[0];
    key0 = h[1];
  }
  int64_t nblocks = ceil_div(n, RS_TILE);
  BufP k2 = ctx->alloc(8 * (size_t)n), v2 = ctx->alloc(4 * (size_t)n);
  BufP hist = ctx->alloc(4 * (size_t)(256 * nblocks)), offs = ctx->alloc(4 * (size_t)(256 * nblocks));
  BufP total = ctx->alloc(8);
  uint64_t *ka = keys, *kb = k2->as<uint64_t>();
  uint32_t *va = vals, *vb = v2->as<uint32_t>();
  std::vector<int> shifts;
  for (int shift = begin_bit; shift < end_bit; shift += 8)
    if (((varying >> shift) & 0xff) != 0) shifts.push_back(shift);
  // packed records between the first and the last pass: the varying key bits (known exactly when
  // the diff pass ran) must fit 32 bits next to the 32-bit row id
  int key_lo = 0;
  uint64_t key_const = 0;
  bool packed = false;
  if (varying != ~0ull && varying != 0 && shifts.size() >= 2) {
    key_lo = shifts.front();
    int key_hi = 64 - __builtin_clzll(varying);
    const bool low_bits_constant = key_lo == 0 || (varying & ((1ull << key_lo) - 1)) == 0;
    if (key_hi - key_lo <= 32 && low_bits_constant) {
      packed = true;
      uint64_t k0 = key0; // constant bits come from any key
      uint64_t span = (key_hi - key_lo == 64) ? ~0ull : (((1ull << (key_hi - key_lo)) - 1) << key_lo);
      key_const = k0 & ~span;
    }
  }
  for (size_t pi = 0; pi < shifts.size(); pi++) {
    const int shift = shifts[pi];
    const bool in_packed = packed && pi > 0, out_packed = packed && pi + 1 < shifts.size();
    const int eff = in_packed ? shift - key_lo + 32 : shift; // where the digit sits in what this pass reads
    rs_hist_kernel<false><<<dim3((unsigned)nblocks), dim3(RS_WG), 0, ctx->stream>>>(ka, n, eff, nblocks,
                                                                                   hist->as<uint32_t>());
    SQ_HIP(hipGetLastError());
    exclusive_scan_u32(ctx, hist->as<uint32_t>(), 256 * nblocks, nullptr, offs->as<uint32_t>(),
                       total->as<uint64_t>());
    dim3 g((unsigned)nblocks), b(RS_WG);
    if (!in_packed && !out_packed)
      rs_scatter_kernel<RS_PAIR, RS_PAIR><<<g, b, 0, ctx->stream>>>(ka, va, n, shift, nblocks, offs->as<uint32_t>(), kb, vb, 0, 0);
    else if (!in_packed)
      rs_scatter_kernel<RS_PAIR, RS_PACKED><<<g, b, 0, ctx->stream>>>(ka, va, n, shift, nblocks, offs->as<uint32_t>(), kb, vb, key_lo, key_const);
    else if (out_packed)
      rs_scatter_kernel<RS_PACKED, RS_PACKED><<<g, b, 0, ctx->stream>>>(ka, va, n, eff, nblocks, offs->as<uint32_t>(), kb, vb, key_lo, key_const);
    else
      rs_scatter_kernel<RS_PACKED, RS_PAIR><<<g, b, 0, ctx->stream>>>(ka, va, n, eff, nblocks, offs->as<uint32_t>(), kb, vb, key_lo, key_const);
    SQ_HIP(hipGetLastError());
    std::swap(ka, kb);
    std::swap(va, vb);
  }
  if (ka != keys) {
    SQ_HIP(hipMemcpyAsync(keys, ka, 8 * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
    SQ_HIP(hipMemcpyAsync(vals, va, 4 * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
  }
}

// perm = the row indices 0 .. n-1 ordered by their u32 key (stable), keys below 2^bits: the ordering of an aggregation's
// groups by first row (hash_agg.rs:98) without widening the keys to u64, without an iota array and without sorted keys
// coming back — the first pass reads the u32 keys and packs (key << 32 | index) words, the last writes the indices only.
void radix_sort_index_u32(Ctx *ctx, const uint32_t *keys, int64_t n, int bits, uint32_t *perm) {
  if (n <= 0) return;
  if (n > 0xffffffffll) fail(SQLRS_ERR_INTERNAL, "radix sort: more than 2^32 rows");
  ProfScope ps(ctx, "radix_sort");
  const int passes = std::max(1, (std::min(bits, 32) + 7) / 8);
  const int64_t nblocks = ceil_div(n, RS_TILE);
  BufP w0 = ctx->alloc(8 * (size_t)n), w1 = passes > 2 ? ctx->alloc(8 * (size_t)n) : nullptr;
  BufP hist = ctx->alloc(4 * (size_t)(256 * nblocks)), offs = ctx->alloc(4 * (size_t)(256 * nblocks));
  BufP total = ctx->alloc(8);
  dim3 g((unsigned)nblocks), b(RS_WG);
  const uint64_t *in = (const uint64_t *)keys;
  uint64_t *out = w0->as<uint64_t>();
  for (int pi = 0; pi < passes; pi++) {
    const bool first = pi == 0, last = pi + 1 == passes;
    const int shift = first ? 0 : 32 + 8 * pi; // the u32 keys themselves, then the key half of the packed words
    if (first) rs_hist_kernel<true><<<g, b, 0, ctx->stream>>>(in, n, shift, nblocks, hist->as<uint32_t>());
    else rs_hist_kernel<false><<<g, b, 0, ctx->stream>>>(in, n, shift, nblocks, hist->as<uint32_t>());
    SQ_HIP(hipGetLastError());
    exclusive_scan_u32(ctx, hist->as<uint32_t>(), 256 * nblocks, nullptr, offs->as<uint32_t>(), total->as<uint64_t>());
    if (first && last)
      rs_scatter_kernel<RS_PAIR, RS_PAIR, true><<<g, b, 0, ctx->stream>>>(in, nullptr, n, shift, nblocks, offs->as<uint32_t>(), nullptr, perm, 0, 0);
    else if (first)
      rs_scatter_kernel<RS_PAIR, RS_PACKED, true><<<g, b, 0, ctx->stream>>>(in, nullptr, n, shift, nblocks, offs->as<uint32_t>(), out, nullptr, 0, 0);
    else if (last)
      rs_scatter_kernel<RS_PACKED, RS_PAIR><<<g, b, 0, ctx->stream>>>(in, nullptr, n, shift, nblocks, offs->as<uint32_t>(), nullptr, perm, 0, 0);
    else
      rs_scatter_kernel<RS_PACKED, RS_PACKED><<<g, b, 0, ctx->stream>>>(in, nullptr, n, shift, nblocks, offs->as<uint32_t>(), out, nullptr, 0, 0);
    SQ_HIP(hipGetLastError());
    in = out;
    out = (out == w0->as<uint64_t>() && w1) ? w1->as<uint64_t>() : w0->as<uint64_t>();
  }
}

} // namespace sq
