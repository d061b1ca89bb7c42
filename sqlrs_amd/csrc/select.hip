// select.hip — row selections (bitmaps), order-preserving compaction, and the fused
// "col OP constant" filter kernel of config C2.
//
// Roofline (all HBM-bound, no reuse): the fused filter reads 8 B/row and writes 8 B per
// kept row (+ 1 bit/row for the selection mask) = the algorithmic 8N + 8sN of SURVEY §8d.
#include "common.hpp"
#include "device_utils.hpp"
#include "prims.hpp"

namespace sq {

// ------------------------------------------------------------ bit utilities --
__global__ void mask_bits_kernel(const uint64_t *__restrict__ values,
                                 const uint64_t *__restrict__ validity, int invert, int64_t rows,
                                 int64_t nwords, uint64_t *__restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= nwords) return;
  uint64_t v = values[i];
  if (invert) v = ~v;
  if (validity) v &= validity[i];
  int64_t rem = rows - i * 64;
  if (rem < 64) v &= (rem <= 0) ? 0ull : ((1ull << rem) - 1);
  out[i] = v;
}

// one wave per tile of 64 mask words; writes tile_off[tile] = kept rows before the tile
__global__ __launch_bounds__(BLOCK) void tile_offsets_kernel(const uint64_t *__restrict__ bits,
                                                             int64_t nwords, int64_t num_tiles,
                                                             uint64_t *__restrict__ tile_off,
                                                             uint64_t *desc, unsigned *ticket,
                                                             uint64_t *total) {
  const int lane = lane_id();
  unsigned t = 0;
  if (lane == 0) t = atomicAdd(ticket, 1u); // one ticket per wave buys 8 consecutive tiles
  int64_t tile0 = (int64_t)(unsigned)__shfl((int)t, 0, 64) * LB_TILES_PER_TICKET;
  uint64_t m[LB_TILES_PER_TICKET];
#pragma unroll
  for (int sub = 0; sub < LB_TILES_PER_TICKET; sub++) {
    int64_t w = (tile0 + sub) * TILE_WORDS + lane;
    m[sub] = (w < nwords) ? bits[w] : 0ull;
  }
#pragma unroll
  for (int sub = 0; sub < LB_TILES_PER_TICKET; sub++) {
    int64_t tile = tile0 + sub;
    if (tile >= num_tiles) return;
    uint64_t agg = wave_sum_u64((uint64_t)__popcll(m[sub]));
    uint64_t excl = lookback_wave(desc, tile, agg);
    if (lane == 0) {
      tile_off[tile] = excl;
      if (tile == num_tiles - 1) *total = excl + agg;
    }
  }
}

void selection_finish(Ctx *ctx, Selection &s) {
  int64_t tiles = ceil_div(std::max<int64_t>(s.rows, 1), TILE_ROWS);
  s.tile_off = ctx->alloc(8 * (size_t)tiles);
  if (s.rows == 0) {
    s.count = 0;
    return;
  }
  BufP desc = ctx->alloc_zero(8 * (size_t)tiles + 16);
  unsigned *ticket = (unsigned *)(desc->as<uint64_t>() + tiles);
  uint64_t *total = desc->as<uint64_t>() + tiles + 1;
  {
    ProfScope ps(ctx, "tile_offsets");
    int64_t nwords = ceil_div(s.rows, 64);
    unsigned blocks = (unsigned)ceil_div(ceil_div(tiles, LB_TILES_PER_TICKET), WAVES_PER_BLOCK);
    tile_offsets_kernel<<<dim3(blocks), dim3(BLOCK), 0, ctx->stream>>>(
        s.bits, nwords, tiles, s.tile_off->as<uint64_t>(), desc->as<uint64_t>(), ticket, total);
    SQ_HIP(hipGetLastError());
  }
  s.count = (int64_t)ctx->fetch_value(total);
}

static Selection selection_from_words(Ctx *ctx, const uint64_t *values, const uint64_t *validity,
                                      bool invert, int64_t rows) {
  Selection s;
  s.rows = rows;
  int64_t nwords = ceil_div(std::max<int64_t>(rows, 1), 64);
  s.own_bits = ctx->alloc(8 * (size_t)nwords);
  s.bits = s.own_bits->as<uint64_t>();
  if (rows > 0) {
    ProfScope ps(ctx, "mask_bits");
    mask_bits_kernel<<<dim3((unsigned)ceil_div(nwords, 256)), dim3(256), 0, ctx->stream>>>(
        values, validity, invert ? 1 : 0, rows, nwords, s.own_bits->as<uint64_t>());
    SQ_HIP(hipGetLastError());
  }
  selection_finish(ctx, s);
  return s;
}

Selection selection_from_mask(Ctx *ctx, const DCol &mask) {
  if (mask.dtype != SQLRS_BOOLEAN)
    fail(SQLRS_ERR_INTERNAL, "filter executor expected evaluate boolean array");
  if (mask.stride == 0) {
    DCol m = materialize_scalar(ctx, mask, mask.length);
    return selection_from_words(ctx, m.v<uint64_t>(), m.validity, false, m.length);
  }
  return selection_from_words(ctx, mask.v<uint64_t>(), mask.validity, false, mask.length);
}
Selection selection_from_clear_bits(Ctx *ctx, const uint64_t *bits, int64_t rows) {
  return selection_from_words(ctx, bits, nullptr, true, rows);
}
Selection selection_from_set_bits(Ctx *ctx, const uint64_t *bits, int64_t rows) {
  return selection_from_words(ctx, bits, nullptr, false, rows);
}

// ------------------------------------------------------------------ compaction --
// One block per tile.  Every wave loads the tile's 64 mask words (lane l <- word l), scans the
// popcounts in-register and then streams its own 16 words: no LDS, no barrier.
template <class T>
__global__ __launch_bounds__(BLOCK) void compact_kernel(const T *__restrict__ in,
                                                        const uint64_t *__restrict__ bits,
                                                        const uint64_t *__restrict__ tile_off,
                                                        int64_t rows, int64_t nwords,
                                                        T *__restrict__ out) {
  const int64_t tile = blockIdx.x;
  const int lane = lane_id(), w = wave_id();
  int64_t wi = tile * TILE_WORDS + lane;
  uint64_t m = (wi < nwords) ? bits[wi] : 0ull;
  uint32_t pc = (uint32_t)__popcll(m);
  uint32_t excl = wave_iscan_u32(pc) - pc;
  const uint64_t base = tile_off[tile];
#pragma unroll 4
  for (int j = 0; j < 16; j++) {
    int W = w * 16 + j;
    uint64_t mw = shfl_u64(m, W);
    uint32_t off = (uint32_t)__shfl((int)excl, W, 64);
    int64_t row = (tile * TILE_WORDS + W) * 64 + lane;
    if ((mw >> lane) & 1) out[base + off + mbcnt(mw)] = in[row];
  }
}

// row ids of the kept rows
template <class T>
__global__ __launch_bounds__(BLOCK) void compact_iota_kernel(const uint64_t *__restrict__ bits,
                                                             const uint64_t *__restrict__ tile_off,
                                                             int64_t nwords, T *__restrict__ out) {
  const int64_t tile = blockIdx.x;
  const int lane = lane_id(), w = wave_id();
  int64_t wi = tile * TILE_WORDS + lane;
  uint64_t m = (wi < nwords) ? bits[wi] : 0ull;
  uint32_t pc = (uint32_t)__popcll(m);
  uint32_t excl = wave_iscan_u32(pc) - pc;
  const uint64_t base = tile_off[tile];
#pragma unroll 4
  for (int j = 0; j < 16; j++) {
    int W = w * 16 + j;
    uint64_t mw = shfl_u64(m, W);
    uint32_t off = (uint32_t)__shfl((int)excl, W, 64);
    int64_t row = (tile * TILE_WORDS + W) * 64 + lane;
    if ((mw >> lane) & 1) out[base + off + mbcnt(mw)] = (T)row;
  }
}

// bit column (BOOLEAN values / validity) -> one byte per kept row
__global__ __launch_bounds__(BLOCK) void compact_bits_to_bytes_kernel(
    const uint64_t *__restrict__ src_bits, const uint64_t *__restrict__ bits,
    const uint64_t *__restrict__ tile_off, int64_t nwords, uint8_t *__restrict__ out) {
  const int64_t tile = blockIdx.x;
  const int lane = lane_id(), w = wave_id();
  int64_t wi = tile * TILE_WORDS + lane;
  uint64_t m = (wi < nwords) ? bits[wi] : 0ull;
  uint64_t sv = (wi < nwords) ? src_bits[wi] : 0ull;
  uint32_t pc = (uint32_t)__popcll(m);
  uint32_t excl = wave_iscan_u32(pc) - pc;
  const uint64_t base = tile_off[tile];
  for (int j = 0; j < 16; j++) {
    int W = w * 16 + j;
    uint64_t mw = shfl_u64(m, W);
    uint64_t sw = shfl_u64(sv, W);
    uint32_t off = (uint32_t)__shfl((int)excl, W, 64);
    if ((mw >> lane) & 1) out[base + off + mbcnt(mw)] = (uint8_t)((sw >> lane) & 1);
  }
}

__global__ void pack_bytes_kernel(const uint8_t *__restrict__ bytes, int64_t n,
                                  uint64_t *__restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  bool b = (i < n) && bytes[i];
  uint64_t m = __ballot(b);
  if (lane_id() == 0 && (i / 64) * 64 < n) out[i >> 6] = m;
}

static BufP compact_bits(Ctx *ctx, const uint64_t *src_bits, const Selection &s) {
  BufP out = ctx->alloc_zero(bitmap_bytes(std::max<int64_t>(s.count, 1)));
  if (s.count == 0) return out;
  BufP bytes = ctx->alloc((size_t)s.count);
  int64_t tiles = ceil_div(s.rows, TILE_ROWS), nwords = ceil_div(s.rows, 64);
  compact_bits_to_bytes_kernel<<<dim3((unsigned)tiles), dim3(BLOCK), 0, ctx->stream>>>(
      src_bits, s.bits, s.tile_off->as<uint64_t>(), nwords, bytes->as<uint8_t>());
  SQ_HIP(hipGetLastError());
  int64_t n64 = round_up((size_t)s.count, 64);
  pack_bytes_kernel<<<dim3((unsigned)ceil_div(n64, 256)), dim3(256), 0, ctx->stream>>>(
      bytes->as<uint8_t>(), s.count, out->as<uint64_t>());
  SQ_HIP(hipGetLastError());
  return out;
}

BufP selection_indices_u32(Ctx *ctx, const Selection &s) {
  BufP out = ctx->alloc(4 * (size_t)std::max<int64_t>(s.count, 1));
  if (s.count == 0) return out;
  ProfScope ps(ctx, "compact_iota");
  compact_iota_kernel<uint32_t><<<dim3((unsigned)ceil_div(s.rows, TILE_ROWS)), dim3(BLOCK), 0,
                                  ctx->stream>>>(s.bits, s.tile_off->as<uint64_t>(),
                                                 ceil_div(s.rows, 64), out->as<uint32_t>());
  SQ_HIP(hipGetLastError());
  return out;
}
BufP selection_indices_u64(Ctx *ctx, const Selection &s) {
  BufP out = ctx->alloc(8 * (size_t)std::max<int64_t>(s.count, 1));
  if (s.count == 0) return out;
  ProfScope ps(ctx, "compact_iota");
  compact_iota_kernel<uint64_t><<<dim3((unsigned)ceil_div(s.rows, TILE_ROWS)), dim3(BLOCK), 0,
                                  ctx->stream>>>(s.bits, s.tile_off->as<uint64_t>(),
                                                 ceil_div(s.rows, 64), out->as<uint64_t>());
  SQ_HIP(hipGetLastError());
  return out;
}

DCol compact_column(Ctx *ctx, const DCol &cin, const Selection &s) {
  DCol c = cin.stride == 0 ? materialize_scalar(ctx, cin, s.rows) : cin;
  if (c.length != s.rows) fail(SQLRS_ERR_INTERNAL, "compact: length mismatch");
  if (c.dtype == SQLRS_UTF8) { // variable width: gather by row id
    BufP idx = selection_indices_u32(ctx, s);
    return gather_column(ctx, c, idx->p, false, nullptr, s.count);
  }
  DCol o;
  o.dtype = c.dtype;
  o.length = s.count;
  int64_t tiles = ceil_div(std::max<int64_t>(s.rows, 1), TILE_ROWS), nwords = ceil_div(s.rows, 64);
  if (c.has_nulls() || (c.validity && c.null_count < 0)) {
    o.own_validity = compact_bits(ctx, c.validity, s);
    o.validity = o.own_validity->as<uint64_t>();
    o.null_count = -1;
  }
  if (c.dtype == SQLRS_BOOLEAN) {
    o.own_values = compact_bits(ctx, c.v<uint64_t>(), s);
    o.values = o.own_values->p;
    return o;
  }
  size_t w = width_of(c.dtype);
  o.own_values = ctx->alloc(w * (size_t)std::max<int64_t>(s.count, 1) + 16);
  o.values = o.own_values->p;
  if (s.count == 0 || s.rows == 0) return o;
  ProfScope ps(ctx, "compact");
  if (w == 8)
    compact_kernel<uint64_t><<<dim3((unsigned)tiles), dim3(BLOCK), 0, ctx->stream>>>(
        c.v<uint64_t>(), s.bits, s.tile_off->as<uint64_t>(), s.rows, nwords,
        o.own_values->as<uint64_t>());
  else
    compact_kernel<uint32_t><<<dim3((unsigned)tiles), dim3(BLOCK), 0, ctx->stream>>>(
        c.v<uint32_t>(), s.bits, s.tile_off->as<uint64_t>(), s.rows, nwords,
        o.own_values->as<uint32_t>());
  SQ_HIP(hipGetLastError());
  return o;
}

// ---------------------------------------------------- fused filter (config C2) --
enum { CMP_GT = 0, CMP_LT, CMP_GE, CMP_LE, CMP_EQ, CMP_NE };

template <class T> struct CmpKey; // total-order key so that f64 compares like the oracle
template <> struct CmpKey<int64_t> {
  static __device__ __forceinline__ int64_t key(int64_t v) { return v; }
};
template <> struct CmpKey<int32_t> {
  static __device__ __forceinline__ int32_t key(int32_t v) { return v; }
};
template <> struct CmpKey<double> {
  static __device__ __forceinline__ uint64_t key(double v) { return f64_to_ordered(v); }
};

template <int OP, class K> __device__ __forceinline__ bool cmp_op(K a, K b) {
  if (OP == CMP_GT) return a > b;
  if (OP == CMP_LT) return a < b;
  if (OP == CMP_GE) return a >= b;
  if (OP == CMP_LE) return a <= b;
  if (OP == CMP_EQ) return a == b;
  return a != b;
}

// One block = FILTER_WAVES worker waves + 1 scan wave per FILTER tile of FILTER_WAVES x 2048 rows (whole multiples of the
// 4096-row tiles that `tile_off` and the other compaction kernels use).  Worker wave w owns 32
// chunks of 64 rows, so each ballot IS one word of the selection mask; all 32 loads of a wave are
// issued up front (16 KiB per wave in flight).  Ranks come from popcounts, the tile's global
// offset from the decoupled look-back, and kept values are written contiguously per wave.
//
// Why 16384 rows: the chained scan only keeps up while (tiles started per us) x (descriptor read
// latency) stays well below the 64-descriptor look-back window (device_utils.hpp).  With
// 4096-row tiles that ratio was ~1 and a tile spent 80 % of its life waiting for its prefix
// (42 K of 55 K cycles): 3.4 ms per 1e9 rows no matter how the loads and stores were scheduled.
//
// The scan wave owns the look-back so that its spinning descriptor reads (s_waitcnt vmcnt(0))
// never sit in front of a worker's data loads in the in-order vmcnt queue.
#ifdef FILTER_TIMING // phase timers (tools only): -DFILTER_TIMING via SQLRS_EXTRA_CFLAGS
__device__ unsigned long long filter_timing[8];
#define FT(i) do { if (threadIdx.x == 0) { long long now_ = clock64(); atomicAdd(&filter_timing[i], (unsigned long long)(now_ - tlast)); tlast = now_; } } while (0)
#define FT_INIT long long tlast = clock64()
#else
#define FT(i) do {} while (0)
#define FT_INIT do {} while (0)
#endif
#ifndef FILTER_LB_LOADS
#define FILTER_LB_LOADS 1
#endif
// Round 5: TEN worker waves + the scan wave (20480-row tiles).  The int64 / f64 kernel holds two tiles in 163 VGPRs, i.e. three
// waves fit a SIMD and twelve a CU, and one persistent workgroup is resident per CU: with 8 + 1 waves a quarter of the CU's wave
// slots stood empty.  One box, same process order: s = 0.5 0.349 -> 0.312 ms, s = 0.01 0.270 -> 0.246 (6 workers: 0.353 / 0.305;
// smaller tiles at higher occupancy were all slower: profiles/r05m_filter_block_shapes.txt, r05s_filter_waves.txt).
#ifndef FILTER_WAVES_N
#define FILTER_WAVES_N 10
#endif
constexpr int FILTER_WAVES = FILTER_WAVES_N; // even: two worker waves per 4096-row compaction tile
#ifndef FILTER_CHUNKS_N
#define FILTER_CHUNKS_N 32
#endif
constexpr int FILTER_CHUNKS = FILTER_CHUNKS_N;                   // 64-row chunks per worker wave
constexpr int FILTER_TILE_ROWS = FILTER_WAVES * FILTER_CHUNKS * 64; // 16384
constexpr int FILTER_SUB = FILTER_TILE_ROWS / TILE_ROWS;           // 4096-row tiles per filter tile
constexpr int FILTER_BLOCK = (FILTER_WAVES + 1) * 64;
constexpr int FILTER_WPS = TILE_ROWS / (FILTER_CHUNKS * 64);      // worker waves per 4096-row compaction tile
static_assert(FILTER_CHUNKS <= 64 && FILTER_WPS >= 1 && FILTER_SUB * FILTER_WPS == FILTER_WAVES && (FILTER_CHUNKS & (FILTER_CHUNKS - 1)) == 0,
              "tile_off mapping: whole worker waves per 4096-row tile");

template <class T, int OP, bool HASV>
__global__ __launch_bounds__(FILTER_BLOCK) void filter_cmp_const_kernel(
    const T *__restrict__ in, const uint64_t *__restrict__ validity, T k, int64_t rows,
    int64_t num_tiles, T *__restrict__ out, uint64_t *__restrict__ sel_bits,
    uint64_t *__restrict__ tile_off, uint64_t *desc, unsigned *ticket, uint64_t *total, int use_ticket) {
  __shared__ int64_t s_tile;
  __shared__ uint32_t s_wave[FILTER_WAVES];
  __shared__ uint64_t s_excl;
  unsigned *timeout = use_ticket ? nullptr : ticket + 1; // word after the ticket counter
  // tile order: blockIdx (workgroups are dispatched in index order, so a tile's predecessors are
  // running or done; the look-back spin is bounded and the host reruns the launch with tickets if
  // it ever times out), or one atomic ticket per tile in that fallback launch.
  int64_t tile = blockIdx.x;
  if (use_ticket) {
    if (threadIdx.x == 0) s_tile = (int64_t)atomicAdd(ticket, 1u);
    __syncthreads();
    tile = s_tile;
  }
  const int lane = lane_id(), w = wave_id();
  const int64_t nw = (rows + 63) >> 6;
  if (w == FILTER_WAVES) { // ---- scan wave
#ifdef FILTER_TIMING
    long long tl0 = clock64();
#endif
    __syncthreads(); // (1) the workers' counts are in s_wave
#ifdef FILTER_TIMING
    long long tl1 = clock64();
#endif
    uint32_t c = lane < FILTER_WAVES ? s_wave[lane] : 0;
    uint32_t inc = wave_iscan_u32(c);
    uint64_t agg = (uint32_t)__shfl((int)inc, 63, 64);
    uint64_t excl = lookback_wave<FILTER_LB_LOADS>(desc, tile, agg, timeout);
    // kept rows before each 4096-row tile = before waves 0, 2, 4, 6
    if (lane < FILTER_WAVES && (lane % FILTER_WPS) == 0 && tile_off) {
      int64_t t4 = tile * FILTER_SUB + (lane / FILTER_WPS);
      if (t4 * TILE_ROWS < rows) tile_off[t4] = excl + (inc - c);
    }
    if (lane == 0) {
      s_excl = excl;
      if (tile == num_tiles - 1) *total = excl + agg;
    }
#ifdef FILTER_TIMING
    if (lane == 0) { atomicAdd(&filter_timing[5], (unsigned long long)(tl1 - tl0)); atomicAdd(&filter_timing[6], (unsigned long long)(clock64() - tl1)); }
#endif
    __syncthreads(); // (2)
    return;
  }
  // ---- worker waves
  FT_INIT;
  const auto kk = CmpKey<T>::key(k);
  const int64_t wrow = tile * FILTER_TILE_ROWS + (int64_t)w * (FILTER_CHUNKS * 64) + lane;
  const int64_t wword = (tile * FILTER_WAVES + w) * FILTER_CHUNKS;
  T v[FILTER_CHUNKS];
#pragma unroll
  for (int j = 0; j < FILTER_CHUNKS; j++) v[j] = __builtin_nontemporal_load(in + min(wrow + j * 64, rows - 1));
  // the wave's 32 validity words in one coalesced load: lane j holds the word of chunk j
  const uint64_t vw = HASV ? validity[min(wword + (lane & (FILTER_CHUNKS - 1)), nw - 1)] : ~0ull;
  FT(0);
  uint64_t mine = 0; // lane j keeps the mask word of chunk j
  uint32_t wave_cnt = 0;
#pragma unroll
  for (int j = 0; j < FILTER_CHUNKS; j++) {
    bool keep = (wrow + j * 64 < rows) && cmp_op<OP>(CmpKey<T>::key(v[j]), kk);
    uint64_t b = __ballot(keep);
    if (HASV) b &= readlane_u64(vw, j); // rows past the end are not kept (keep is false there)
    mine = (lane == j) ? b : mine;
    wave_cnt += (uint32_t)__popcll(b);
  }
  if (lane == 0) s_wave[w] = wave_cnt;
  if (sel_bits && lane < FILTER_CHUNKS && wword + lane < nw) sel_bits[wword + lane] = mine;
  FT(1);
  __syncthreads(); // (1)
  FT(2);
  __syncthreads(); // (2) the scan wave has published the tile's offset
  FT(3);
  uint64_t pos = s_excl;
  for (int q = 0; q < w; q++) pos += s_wave[q];
  const uint32_t mlo = (uint32_t)mine, mhi = (uint32_t)(mine >> 32);
#pragma unroll
  for (int j = 0; j < FILTER_CHUNKS; j++) {
    uint64_t m = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)mhi, j) << 32) |
                 (uint32_t)__builtin_amdgcn_readlane((int)mlo, j);
    if ((m >> lane) & 1) out[pos + mbcnt(m)] = v[j];
    pos += (uint32_t)__popcll(m);
  }
  FT(4);
}

// Persistent, software-pipelined variant (the default launch): block b takes tiles b, b + G, ...
// and keeps TWO tiles in registers — the 32 loads of tile i+1 are issued before tile i's ballots,
// look-back wait and stores, so the ~6 us a tile waits for its prefix are spent streaming the
// next tile instead of idling the memory pipe.  The loop body is written twice (va/vb swap
// roles) instead of copying registers, loads are unconditional, and s_wave / s_excl are
// double-buffered by tile parity (no third barrier).  Without tickets a tile's predecessors are
// only guaranteed to be running if all G blocks are resident: the host sizes G by occupancy,
// the look-back spin is bounded, and on a timeout the launch is redone by the ticketed
// one-tile-per-block kernel above.
template <class T, bool HASV>
__device__ __forceinline__ void filter_load_tile(const T *__restrict__ in, const uint64_t *__restrict__ validity,
                                                 int64_t rows, int64_t tile, T (&v)[FILTER_CHUNKS], uint64_t &vw) {
  uint32_t tid = threadIdx.x; // (opaque per tile, see filter_tile)
  asm volatile("" : "+v"(tid));
  const int64_t wrow = tile * FILTER_TILE_ROWS + (int64_t)(tid >> 6) * (FILTER_CHUNKS * 64) + (tid & 63);
#pragma unroll
  for (int j = 0; j < FILTER_CHUNKS; j++) v[j] = __builtin_nontemporal_load(in + min(wrow + j * 64, rows - 1));
  // the wave's 32 validity words in one coalesced load: lane j holds the word of chunk j
  const int64_t wword = (tile * FILTER_WAVES + (int64_t)(tid >> 6)) * FILTER_CHUNKS;
  vw = HASV ? validity[min(wword + (int64_t)(tid & (FILTER_CHUNKS - 1)), ((rows + 63) >> 6) - 1)] : ~0ull;
}

template <class T, int OP, bool HASV, class KK>
__device__ __forceinline__ void filter_tile(const T (&v)[FILTER_CHUNKS], const uint64_t vw, KK kk, int64_t rows,
                                            int64_t tile,
                                            T *__restrict__ out, uint64_t *__restrict__ sel_bits,
                                            uint32_t *s_wave, const uint64_t *s_excl) {
  // The thread id goes through an opaque asm per tile (round 3): values derived from it (lane * 4, the wave's word
  // base, ...) are loop invariants of the persistent kernel, and with two tiles of 64-bit rows in registers (128 of
  // the 168 VGPRs three waves per SIMD may use) the compiler SPILLED three of them — and every reload inside the
  // loop is a scratch load behind the 32 prefetched row loads on the in-order vmcnt: `s_waitcnt vmcnt(0)` in the
  // middle of the pipeline, i.e. the prefetch was waited for as soon as it was issued.  Recomputing them per tile
  // costs a few VALU operations and leaves no scratch.
  uint32_t tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = (int)(tid & 63), w = (int)(tid >> 6);
  const int64_t nw = (rows + 63) >> 6;
  const int64_t wrow = tile * FILTER_TILE_ROWS + (int64_t)w * (FILTER_CHUNKS * 64) + lane;
  const int64_t wword = (tile * FILTER_WAVES + w) * FILTER_CHUNKS;
  uint64_t mine = 0; // lane j keeps the mask word of chunk j
  uint32_t wave_cnt = 0;
#pragma unroll
  for (int j = 0; j < FILTER_CHUNKS; j++) {
    bool keep = (wrow + j * 64 < rows) && cmp_op<OP>(CmpKey<T>::key(v[j]), kk);
    uint64_t b = __ballot(keep);
    if (HASV) b &= readlane_u64(vw, j);
    mine = (lane == j) ? b : mine;
    wave_cnt += (uint32_t)__popcll(b);
  }
  if (lane == 0) s_wave[w] = wave_cnt;
  if (sel_bits && lane < FILTER_CHUNKS && wword + lane < nw) sel_bits[wword + lane] = mine;
  __syncthreads(); // (1)
  __syncthreads(); // (2) the scan wave has published the tile's offset
  uint64_t pos = *s_excl;
  for (int q = 0; q < w; q++) pos += s_wave[q];
  const uint32_t mlo = (uint32_t)mine, mhi = (uint32_t)(mine >> 32);
#pragma unroll
  for (int j = 0; j < FILTER_CHUNKS; j++) {
    uint64_t m = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)mhi, j) << 32) |
                 (uint32_t)__builtin_amdgcn_readlane((int)mlo, j);
    if ((m >> lane) & 1) out[pos + mbcnt(m)] = v[j];
    pos += (uint32_t)__popcll(m);
  }
}

template <class T, int OP, bool HASV>
__global__ __launch_bounds__(FILTER_BLOCK) void filter_cmp_const_persistent_kernel(
    const T *__restrict__ in, const uint64_t *__restrict__ validity, T k, int64_t rows,
    int64_t num_tiles, T *__restrict__ out, uint64_t *__restrict__ sel_bits,
    uint64_t *__restrict__ tile_off, uint64_t *desc, unsigned *timeout, uint64_t *total) {
  __shared__ uint32_t s_wave[2][FILTER_WAVES];
  __shared__ uint64_t s_excl[2];
  const int64_t G = gridDim.x;
  int64_t tile = blockIdx.x; // G <= num_tiles: every block has a first tile
  if (wave_id() == FILTER_WAVES) { // ---- scan wave
    const int lane = lane_id();
    for (int p = 0; tile < num_tiles; tile += G, p ^= 1) {
      __syncthreads(); // (1) the workers' counts are in s_wave[p]
      uint32_t c = lane < FILTER_WAVES ? s_wave[p][lane] : 0;
      uint32_t inc = wave_iscan_u32(c);
      uint64_t agg = (uint32_t)__shfl((int)inc, 63, 64);
      uint64_t excl = lookback_wave<FILTER_LB_LOADS>(desc, tile, agg, timeout);
      if (lane < FILTER_WAVES && (lane % FILTER_WPS) == 0 && tile_off) {
        int64_t t4 = tile * FILTER_SUB + (lane / FILTER_WPS);
        if (t4 * TILE_ROWS < rows) tile_off[t4] = excl + (inc - c);
      }
      if (lane == 0) {
        s_excl[p] = excl;
        if (tile == num_tiles - 1) *total = excl + agg;
      }
      __syncthreads(); // (2)
    }
    return;
  }
  // ---- worker waves
  const auto kk = CmpKey<T>::key(k);
  T va[FILTER_CHUNKS], vb[FILTER_CHUNKS];
  uint64_t wa, wb;
  filter_load_tile<T, HASV>(in, validity, rows, tile, va, wa);
  while (true) {
    int64_t nt = tile + G;
    filter_load_tile<T, HASV>(in, validity, rows, min(nt, num_tiles - 1), vb, wb);
    filter_tile<T, OP, HASV>(va, wa, kk, rows, tile, out, sel_bits, s_wave[0], &s_excl[0]);
    if (nt >= num_tiles) break;
    tile = nt;
    nt = tile + G;
    filter_load_tile<T, HASV>(in, validity, rows, min(nt, num_tiles - 1), va, wa);
    filter_tile<T, OP, HASV>(vb, wb, kk, rows, tile, out, sel_bits, s_wave[1], &s_excl[1]);
    if (nt >= num_tiles) break;
    tile = nt;
  }
}

// Workgroups per CU that are resident for sure.  The occupancy API divides the CU's wave slots by
// the waves of a block, but a block's waves are dealt to the four SIMDs starting at the same one:
// two 9-wave blocks need 3 + 3 slots on SIMD 0, and at 96 registers (5 waves per SIMD) the second
// block does NOT fit although 18 <= 20.  The int32 filter (96 registers) was launched with two
// blocks per CU on the API's word, half of them never became resident and every call waited out
// the look-back timeout (1.4 s) before the ticketed rerun.
static int simd_safe_blocks(const void *kernel, int block_threads) {
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, kernel) != hipSuccess) return 1;
  const int regs = std::max(8, (int)round_up((size_t)std::max(fa.numRegs, 1), 8));
  const int per_simd = std::min(8, 512 / regs);
  const int need = (int)ceil_div(ceil_div(block_threads, 64), 4); // waves of one block on its fullest SIMD
  return std::max(1, per_simd / need);
}

template <class T>
static void launch_filter(Ctx *ctx, int op, const DCol &c, T k, int64_t rows, T *out,
                          uint64_t *sel_bits, uint64_t *tile_off, uint64_t *desc, unsigned *ticket,
                          uint64_t *total, int use_ticket) {
  int64_t tiles = ceil_div(rows, FILTER_TILE_ROWS);
  const T *in = c.v<T>();
  const uint64_t *val = c.validity;
  // use_ticket = 0: persistent pipelined kernel, as many blocks as are resident at once;
  // use_ticket = 1: one block per tile, tile ids from an atomic ticket (safe under any dispatch order)
#define SQ_LAUNCH1(OP, HV)                                                                                       \
  do {                                                                                                           \
    if (use_ticket) {                                                                                            \
      filter_cmp_const_kernel<T, OP, HV><<<dim3((unsigned)tiles), dim3(FILTER_BLOCK), 0, ctx->stream>>>(         \
          in, val, k, rows, tiles, out, sel_bits, tile_off, desc, ticket, total, 1);                             \
    } else {                                                                                                     \
      int &occ = ctx->occ_cache[(const void *)filter_cmp_const_persistent_kernel<T, OP, HV>];                   \
      if (!occ) {                                                                                                \
        SQ_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, filter_cmp_const_persistent_kernel<T, OP, HV>, \
                                                            FILTER_BLOCK, 0));                                   \
        const char *fo_e = hook("SQLRS_FILTER_OCC"); /* tuning hook: most resident workgroups per CU */   \
        occ = std::max(1, std::min({occ, fo_e ? std::max(1, std::atoi(fo_e)) : 2,                                \
                                    simd_safe_blocks((const void *)filter_cmp_const_persistent_kernel<T, OP, HV>, FILTER_BLOCK)})); \
      }                                                                                                          \
      unsigned g = (unsigned)std::min<int64_t>(tiles, (int64_t)ctx->num_cus * occ);                              \
      filter_cmp_const_persistent_kernel<T, OP, HV><<<dim3(g), dim3(FILTER_BLOCK), 0, ctx->stream>>>(            \
          in, val, k, rows, tiles, out, sel_bits, tile_off, desc, ticket + 1, total);                            \
    }                                                                                                            \
  } while (0)
#define SQ_LAUNCH(OP)                                                                                            \
  do {                                                                                                           \
    if (val) SQ_LAUNCH1(OP, true);                                                                               \
    else SQ_LAUNCH1(OP, false);                                                                                  \
  } while (0)
  switch (op) {
  case CMP_GT: SQ_LAUNCH(CMP_GT); break;
  case CMP_LT: SQ_LAUNCH(CMP_LT); break;
  case CMP_GE: SQ_LAUNCH(CMP_GE); break;
  case CMP_LE: SQ_LAUNCH(CMP_LE); break;
  case CMP_EQ: SQ_LAUNCH(CMP_EQ); break;
  default: SQ_LAUNCH(CMP_NE); break;
  }
#undef SQ_LAUNCH
#undef SQ_LAUNCH1
  SQ_HIP(hipGetLastError());
}

bool filter_fast_path(Ctx *ctx, const Expr &e, const std::function<const DCol &(int)> &col,
                      int64_t rows, int *col_index, Selection *sel, DCol *out_col) {
  // pattern: InputRef, Constant, comparison   (e.g. `v1 > k`, evaluator.rs:16-21)
  if (e.nodes.size() != 3 || rows == 0) return false;
  const sqlrs_expr_node_t &a = e.nodes[0], &b = e.nodes[1], &o = e.nodes[2];
  if (a.op != SQLRS_EXPR_INPUT_REF || b.op != SQLRS_EXPR_CONSTANT || b.is_null) return false;
  if (o.op < SQLRS_EXPR_GT || o.op > SQLRS_EXPR_NOTEQ) return false;
  const DCol &c = col(a.index);
  if (c.dtype != b.dtype || c.stride == 0) return false;
  if (c.dtype != SQLRS_INT64 && c.dtype != SQLRS_FLOAT64 && c.dtype != SQLRS_INT32) return false;
  int op = o.op - SQLRS_EXPR_GT; // GT, LT, GTEQ, LTEQ, EQ, NOTEQ in header order
  int64_t tiles = ceil_div(rows, FILTER_TILE_ROWS), nwords = ceil_div(rows, 64);
  sel->rows = rows;
  sel->own_bits = ctx->alloc(8 * (size_t)nwords);
  sel->bits = sel->own_bits->as<uint64_t>();
  sel->tile_off = ctx->alloc(8 * (size_t)ceil_div(rows, TILE_ROWS));
  BufP desc = ctx->alloc_zero(8 * (size_t)tiles + 16);
  unsigned *ticket = (unsigned *)(desc->as<uint64_t>() + tiles);
  uint64_t *total = desc->as<uint64_t>() + tiles + 1;
  size_t w = width_of(c.dtype);
  // worst case every row is kept: output sized for `rows` (288 GB of HBM: no second pass)
  BufP out = ctx->alloc(w * (size_t)rows + 16);
  // desc layout: [tiles descriptors][ticket u32, timeout u32][total u64]
  for (int use_ticket = lookback_start_mode(ctx), attempt = 0; use_ticket < 2; use_ticket++, attempt++) {
    if (attempt) SQ_HIP(hipMemsetAsync(desc->p, 0, 8 * (size_t)tiles + 16, ctx->stream)); // rerun after a timeout
    {
      ProfScope ps(ctx, "filter_cmp_const");
      uint64_t *sb = sel->own_bits->as<uint64_t>(), *to = sel->tile_off->as<uint64_t>();
      if (c.dtype == SQLRS_INT64)
        launch_filter<int64_t>(ctx, op, c, (int64_t)b.i, rows, out->as<int64_t>(), sb, to,
                               desc->as<uint64_t>(), ticket, total, use_ticket);
      else if (c.dtype == SQLRS_INT32)
        launch_filter<int32_t>(ctx, op, c, (int32_t)b.i, rows, out->as<int32_t>(), sb, to,
                               desc->as<uint64_t>(), ticket, total, use_ticket);
      else
        launch_filter<double>(ctx, op, c, b.f, rows, out->as<double>(), sb, to,
                              desc->as<uint64_t>(), ticket, total, use_ticket);
    }
#ifdef FILTER_TIMING
    {
      ctx->sync();
      unsigned long long h[8], z[8] = {0};
      SQ_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(filter_timing), sizeof(h)));
      SQ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(filter_timing), z, sizeof(z)));
      if (tiles > 1000)
        fprintf(stderr, "[filter_timing] tiles %lld: cycles/tile issue-loads %.0f ballots(load wait) %.0f bar1 %.0f bar2(lookback) %.0f stores-issue %.0f | scan wave: wait-bar1 %.0f lookback %.0f\n",
                (long long)tiles, (double)h[0] / tiles, (double)h[1] / tiles, (double)h[2] / tiles, (double)h[3] / tiles, (double)h[4] / tiles, (double)h[5] / tiles, (double)h[6] / tiles);
    }
#endif
    const uint64_t *h = (const uint64_t *)ctx->fetch(ticket, 16); // {ticket|timeout, total}
    sel->count = (int64_t)h[1];
    if (use_ticket || (h[0] >> 32) == 0) break; // no look-back timeout: done
    lookback_timed_out(ctx);
  }
  *col_index = a.index;
  out_col->dtype = c.dtype;
  out_col->length = sel->count;
  out_col->null_count = 0; // rows with a NULL predicate input are dropped
  out_col->own_values = out;
  out_col->values = out->p;
  return true;
}

} // namespace sq
