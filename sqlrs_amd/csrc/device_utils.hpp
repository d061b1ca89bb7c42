// device_utils.hpp — wave64 device helpers shared by every kernel file (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace sq {

// two 8-byte words moved with one 16-byte load / store (a {key|row word, value} record)
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

constexpr int WAVE = 64;
constexpr int BLOCK = 256;          // 4 waves
constexpr int WAVES_PER_BLOCK = 4;
constexpr int TILE_WORDS = 64;      // selection-mask words per tile
constexpr int TILE_ROWS = 4096;     // rows per tile of every mask / compaction kernel

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }

// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ int mbcnt(uint64_t mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                        __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src) {
  uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, 64);
  uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 64);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_up_u64(uint64_t v, int d) {
  uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, d, 64);
  uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), d, 64);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
  uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m, 64);
  uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m, 64);
  return ((uint64_t)hi << 32) | lo;
}

// 64-bit value of lane `src` (a compile-time constant after unrolling) broadcast to the wave
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int src) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
  return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_u64(v, m);
  return v;
}
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = min(v, shfl_xor_u64(v, m));
  return v;
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = max(v, shfl_xor_u64(v, m));
  return v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, m, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1)
    v += __longlong_as_double((long long)shfl_xor_u64((uint64_t)__double_as_longlong(v), m));
  return v;
}
// ---- DPP reductions (ALL 64 lanes must be active) ----------------------------------------------------------
// The butterfly above is six dependent ds_bpermute round trips (twelve for a 64-bit value).  Inside a 16-lane row
// the data-parallel-primitive modifiers move a register without touching the LDS crossbar: xor-1 / xor-2 as quad
// permutations, then row_half_mirror (lane i <-> 7 - i) and row_mirror (i <-> 15 - i) pair the 4- and 8-lane groups
// (any pairing of groups that already hold uniform partial results works); the four row results are combined with
// readlanes.  Used where a reduction sits on the per-row path (hot keys of the LDS bucket passes).
template <int CTRL> __device__ __forceinline__ uint32_t dpp_mov_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
template <int CTRL> __device__ __forceinline__ uint64_t dpp_mov_u64(uint64_t v) {
  return ((uint64_t)dpp_mov_u32<CTRL>((uint32_t)(v >> 32)) << 32) | dpp_mov_u32<CTRL>((uint32_t)v);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;
__device__ __forceinline__ double wave_sum_f64_dpp(double v) {
  v += __longlong_as_double((long long)dpp_mov_u64<DPP_XOR1>((uint64_t)__double_as_longlong(v)));
  v += __longlong_as_double((long long)dpp_mov_u64<DPP_XOR2>((uint64_t)__double_as_longlong(v)));
  v += __longlong_as_double((long long)dpp_mov_u64<DPP_HALF_MIRROR>((uint64_t)__double_as_longlong(v)));
  v += __longlong_as_double((long long)dpp_mov_u64<DPP_MIRROR>((uint64_t)__double_as_longlong(v)));
  const uint64_t b = (uint64_t)__double_as_longlong(v);
  return (__longlong_as_double((long long)readlane_u64(b, 0)) + __longlong_as_double((long long)readlane_u64(b, 16))) +
         (__longlong_as_double((long long)readlane_u64(b, 32)) + __longlong_as_double((long long)readlane_u64(b, 48)));
}
__device__ __forceinline__ uint64_t wave_sum_u64_dpp(uint64_t v) {
  v += dpp_mov_u64<DPP_XOR1>(v);
  v += dpp_mov_u64<DPP_XOR2>(v);
  v += dpp_mov_u64<DPP_HALF_MIRROR>(v);
  v += dpp_mov_u64<DPP_MIRROR>(v);
  return (readlane_u64(v, 0) + readlane_u64(v, 16)) + (readlane_u64(v, 32) + readlane_u64(v, 48));
}
__device__ __forceinline__ uint32_t wave_min_u32_dpp(uint32_t v) {
  v = min(v, dpp_mov_u32<DPP_XOR1>(v));
  v = min(v, dpp_mov_u32<DPP_XOR2>(v));
  v = min(v, dpp_mov_u32<DPP_HALF_MIRROR>(v));
  v = min(v, dpp_mov_u32<DPP_MIRROR>(v));
  return min(min((uint32_t)__builtin_amdgcn_readlane((int)v, 0), (uint32_t)__builtin_amdgcn_readlane((int)v, 16)),
             min((uint32_t)__builtin_amdgcn_readlane((int)v, 32), (uint32_t)__builtin_amdgcn_readlane((int)v, 48)));
}

// SEGMENTED inclusive scan over the wave without the LDS crossbar: lane l ends with the sum of v (and the minimum of m)
// over lanes [head .. l], head = first lane of l's segment (head <= l, equal for all lanes of a segment; ALL 64 lanes
// active).  row_shr:1/2/4/8 scan the 16-lane rows; row_bcast15 hands lane 15 / 47 to the row above (rows 1 and 3),
// row_bcast31 lane 31 to rows 2 and 3 — a lane takes them when its segment began before its row / before lane 32.
// (The __shfl_up form is three ds_bpermute per step, 18 per scan: the bucket pass on rows ordered by key spent its time
// there, SQ_ACTIVE_INST_LDS x 4 against random rows, profiles/r05x_sq_sorted.txt.)
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp_take_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_take_f64(double v) {
  const uint64_t b = (uint64_t)__double_as_longlong(v);
  return __longlong_as_double((long long)(((uint64_t)dpp_take_u32<CTRL, ROW_MASK>((uint32_t)(b >> 32)) << 32) | dpp_take_u32<CTRL, ROW_MASK>((uint32_t)b)));
}
__device__ __forceinline__ void wave_seg_iscan_f64_min_u32(double &v, uint32_t &m, int head, int lane) {
  const int in_row = lane & 15;
#define SQ_SEG_STEP(D)                                                      \
  {                                                                         \
    const double o = dpp_take_f64<0x110 + D, 0xf>(v);                       \
    const uint32_t om = dpp_take_u32<0x110 + D, 0xf>(m);                    \
    if (in_row >= D && lane - D >= head) {                                  \
      v += o;                                                               \
      m = om < m ? om : m;                                                  \
    }                                                                       \
  }
  SQ_SEG_STEP(1) SQ_SEG_STEP(2) SQ_SEG_STEP(4) SQ_SEG_STEP(8)
#undef SQ_SEG_STEP
  {
    const double o = dpp_take_f64<0x142, 0xa>(v); // row_bcast15 -> rows 1, 3
    const uint32_t om = dpp_take_u32<0x142, 0xa>(m);
    if ((lane & 16) && head < (lane & ~15)) {
      v += o;
      m = om < m ? om : m;
    }
  }
  {
    const double o = dpp_take_f64<0x143, 0xc>(v); // row_bcast31 -> rows 2, 3
    const uint32_t om = dpp_take_u32<0x143, 0xc>(m);
    if (lane >= 32 && head < 32) {
      v += o;
      m = om < m ? om : m;
    }
  }
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += (uint32_t)__shfl_xor((int)v, m, 64);
  return v;
}
// inclusive scan across the 64 lanes
__device__ __forceinline__ uint32_t wave_iscan_u32(uint32_t v) {
  int lane = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = (uint32_t)__shfl_up((int)v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}
__device__ __forceinline__ uint64_t wave_iscan_u64(uint64_t v) {
  int lane = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint64_t t = shfl_up_u64(v, d);
    if (lane >= d) v += t;
  }
  return v;
}

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}

// ------------------------------------------------- decoupled look-back (tiles) --
// One 8-byte descriptor per tile: bits 63..62 = status, low 62 bits = value.  The value
// IS the flag (single aligned 8-byte agent-scope store/load), so no fence is needed
// (MI355X_MICROARCH.md: data-tagged granules).  Tiles take their index from an atomic
// ticket so every predecessor of a running tile is running or finished.
// Tiles per workgroup.  Must stay 1: with K > 1 a block's first tile waits for the LAST tile of
// the previous block, which serialises the whole grid (measured: 200x slower).
constexpr int LB_TILES_PER_TICKET = 1;
// Tile order.  A single atomic ticket counter sustains only ~88 tickets/us on MI355X
// (MI355X_MICROARCH.md "dequeue"): 1e9 rows / 4096 = 244 K tickets = 2.8 ms.  The fast variant
// therefore uses blockIdx as the tile id (workgroups are dispatched in index order per XCD, so a
// tile's predecessors are running or done) with a BOUNDED spin: if a predecessor never shows
// up the wave raises *timeout and gives up, and the host reruns the launch with atomic tickets,
// which is safe under any dispatch order.
// (2^14 polls with s_sleep between them are a few milliseconds: long enough for any predecessor that is
// merely slow, short enough that a launch whose blocks are not all resident — a second ctx, another
// process or an RCCL kernel on the GPU — costs milliseconds, not the 1.4 s of the former 2^22)
constexpr unsigned LB_SPIN_LIMIT = 1u << 14;
constexpr uint64_t LB_AGG = 1ull << 62;
constexpr uint64_t LB_PFX = 2ull << 62;
constexpr uint64_t LB_VAL = (1ull << 62) - 1;

__device__ __forceinline__ uint64_t lb_load(const uint64_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lb_store(uint64_t *p, uint64_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Called by ALL 64 lanes of ONE wave.  Publishes this tile's aggregate, walks back over
// predecessor descriptors 64 * LB_LOADS at a time and returns the tile's exclusive prefix.
//
// Throughput limit of the chained scan: when tiles start at rate R (tiles/us) and one descriptor
// read costs L us (agent-scope, ~1-2 us under load), the nearest finished prefix is R*L' tiles
// back, where L' is the look-back time itself — it only stays short while R * L / window < 1.
// 4096-row tiles at 290 Grows/s are R = 70 tiles/us: the 64-wide window was saturated and the
// filter ran at the look-back's pace, not HBM's.  The cure is fewer, larger tiles (select.hip),
// not a wider window: LB_LOADS = 4 was measured slower (the descriptor polling traffic grows).
template <int LB_LOADS = 1>
__device__ __forceinline__ uint64_t lookback_wave(uint64_t *desc, int64_t tile, uint64_t aggregate,
                                                  unsigned *timeout = nullptr) {
  const int lane = lane_id();
  if (tile == 0) {
    if (lane == 0) lb_store(&desc[0], LB_PFX | aggregate);
    return 0;
  }
  if (lane == 0) lb_store(&desc[tile], LB_AGG | aggregate);
  uint64_t excl = 0;
  int64_t base = tile - 1;
  unsigned spins = 0;
  bool done = false;
  while (!done) {
    uint64_t d[LB_LOADS];
#pragma unroll
    for (int q = 0; q < LB_LOADS; q++) {
      int64_t t = base - lane - 64 * q;
      d[q] = (t >= 0) ? lb_load(&desc[t]) : LB_PFX; // virtual tiles < 0: prefix 0
    }
    int consumed = 0; // 64-descriptor groups of this window that were complete
#pragma unroll
    for (int q = 0; q < LB_LOADS; q++) {
      if (done || consumed != q) continue;
      uint64_t status = d[q] >> 62;
      uint64_t invalid = __ballot(status == 0);
      uint64_t pfx = __ballot(status == 2);
      int first_pfx = pfx ? __builtin_ctzll(pfx) : 64;
      uint64_t need = first_pfx == 64 ? ~0ull : ((1ull << first_pfx) - 1);
      if (invalid & need) continue; // a predecessor has not published yet: re-read from here
      uint64_t v = (lane <= first_pfx) ? (d[q] & LB_VAL) : 0;
      excl += wave_sum_u64(v);
      consumed = q + 1;
      if (first_pfx < 64) done = true;
    }
    base -= 64 * consumed;
    if (!done && consumed < LB_LOADS) {
      if (timeout && (++spins > LB_SPIN_LIMIT ||
                      __hip_atomic_load(timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        if (lane == 0) __hip_atomic_store(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break; // the host discards this launch and reruns it with tickets
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  if (lane == 0) lb_store(&desc[tile], LB_PFX | (excl + aggregate));
  return excl;
}

// order-preserving transforms to unsigned keys (radix sort, min/max atomics)
__device__ __forceinline__ uint64_t i64_to_ordered(int64_t v) { return (uint64_t)v ^ (1ull << 63); }
__device__ __forceinline__ int64_t ordered_to_i64(uint64_t u) { return (int64_t)(u ^ (1ull << 63)); }
__device__ __forceinline__ uint64_t f64_to_ordered(double f) { // IEEE total order
  uint64_t b = (uint64_t)__double_as_longlong(f);
  return (b >> 63) ? ~b : (b | (1ull << 63));
}
__device__ __forceinline__ double ordered_to_f64(uint64_t u) {
  uint64_t b = (u >> 63) ? (u & ~(1ull << 63)) : ~u;
  return __longlong_as_double((long long)b);
}

} // namespace sq
