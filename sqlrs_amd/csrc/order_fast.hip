// order_fast.hip — ORDER BY on ONE fixed-width key without NULLs, carrying one 8-byte column
// (order.rs:15-67 for the common `ORDER BY k` shape; everything else takes the general path of ops.hip).
//
// The general path is an LSD radix sort of (u64 key, u32 row id) pairs, 8 bits per pass over HBM, followed
// by one gather per column — and a random gather of 8-byte elements fetches a 128-byte line per element
// (12.8 GB for 1e8 rows).  Here the rows themselves travel, and only the TOP bits are sorted through HBM:
//
//   0. min / max of the keys' order-preserving image  -> off = image - min < 2^kbits (kbits <= 32)
//   1. <= 2 stable 8-bit multi-split passes over HBM on the TOP (kbits - rbits <= 16) bits of `off`; a row is
//      the word  off << 32 | row id  plus its carried column (16 B per row and pass; the first pass reads
//      the raw column and builds the word in registers)                                     [LSD order]
//      => rows are grouped by their top bits, in input order inside a group
//   2. group boundaries (a row whose predecessor has other top bits opens its group and closes the previous one)
//   3. one workgroup per group: the group (<= FIN_CAP rows) is sorted on the remaining rbits INSIDE LDS
//      (<= 2 stable 8-bit passes, same ballot ranking as the HBM passes), then key column, carried column
//      and row id (the permutation for any further column) leave in final order.
//
// HBM traffic for 1e8 rows, 31 key bits, one carried column: 0.8 (min/max) + 2 x (0.8 + 3.2) + 0.8
// (boundaries) + 3.6 = 13.2 GB, against 9.6 GB of sort passes + 12.8 GB of gather fetches before.
// Stability (ties in input order, like the general path) holds because every pass is stable.
// A group larger than FIN_CAP (heavily repeated keys with many low bits) sends the call through the same
// passes with rbits = 0 (all key bits sorted in HBM, <= 4 passes: no limit on a group there).
// Keys with MORE than 32 varying bits (random doubles, 63-bit ids) cannot ride in that word: `order_wide`
// below replaces the bits by splitters from a sorted sample and keeps the shape (two passes + in-LDS finish).
#include <atomic>
#include "common.hpp"
#include "device_utils.hpp"
#include "prims.hpp"

namespace sq {

enum { OKIND_I64 = 0, OKIND_F64 = 1, OKIND_I32 = 2 };

template <int KIND> __device__ __forceinline__ uint64_t order_image(const void *__restrict__ vals, int64_t i, int desc) {
  uint64_t u;
  if (KIND == OKIND_I64) u = i64_to_ordered(((const int64_t *)vals)[i]);
  else if (KIND == OKIND_F64) u = f64_to_ordered(((const double *)vals)[i]);
  else u = i64_to_ordered((int64_t)((const int32_t *)vals)[i]);
  return desc ? ~u : u;
}
template <int KIND> __device__ __forceinline__ uint64_t order_unimage(uint64_t u, int desc) { // bits of the original value
  if (desc) u = ~u;
  if (KIND == OKIND_F64) return (uint64_t)__double_as_longlong(ordered_to_f64(u));
  return (uint64_t)ordered_to_i64(u);
}

// `every` > 1: only every `every`-th chunk of 2048 rows is read (a SAMPLE of the column: the caller packs optimistically
// and the first split pass verifies every key against the range, order_fast_impl), plus the last chunk — sorted input has
// an extreme there
// ---- already in order? ------------------------------------------------------------------------------------------
// ORDER BY over rows that arrive in the requested order (a scan of time-ordered data, a clustered key, the output of
// another ORDER) is the identity — also for ties, which a stable sort leaves in input order.  `SAMPLE`: 64 Ki evenly
// spaced neighbour pairs (random input fails this with certainty: no cost beyond one tiny launch, the flag travels with
// the key range); the full test reads the column once (0.15 ms per 1e8 rows) and only runs when the sample found nothing.
template <int KIND, bool SAMPLE>
__global__ __launch_bounds__(256) void order_inversion_kernel(const void *__restrict__ vals, int64_t n, int desc,
                                                              unsigned int *__restrict__ inv) {
  if (SAMPLE) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, T = (int64_t)gridDim.x * blockDim.x;
    const int64_t i = (n - 1) / T * t;
    if (i + 1 < n && order_image<KIND>(vals, i, desc) > order_image<KIND>(vals, i + 1, desc)) atomicOr(inv, 1u);
    return;
  }
  bool bad = false;
  constexpr int KU = 8;
  for (int64_t base = blockIdx.x * (int64_t)(256 * KU) + threadIdx.x; base < n - 1; base += (int64_t)gridDim.x * (256 * KU)) {
    uint64_t a[KU], b[KU];
#pragma unroll
    for (int u = 0; u < KU; u++) {
      const int64_t i = min(base + u * 256, n - 2);
      a[u] = order_image<KIND>(vals, i, desc);
      b[u] = order_image<KIND>(vals, i + 1, desc); // (the neighbouring lane's element: the same lines)
    }
#pragma unroll
    for (int u = 0; u < KU; u++) bad |= a[u] > b[u];
  }
  if (__ballot(bad) && lane_id() == 0) atomicOr(inv, 1u);
}

// Does ONE VALUE hold a visible share of the rows (zeros, a default, a handful of distinct keys)?  Its group of equal top bits
// would be far larger than the in-LDS finish takes, and the attempt would be thrown away after both split passes.  2048
// sampled keys are counted in an LDS table; *out = the largest count (travels with the key range: no round trip of its own).
constexpr uint32_t OW_HEAVY_SAMPLES = 2048, OW_HEAVY_MIN = 6; // 6 of 2048: a share of ~0.3 %
template <int KIND>
__global__ __launch_bounds__(1024) void order_heavy_probe_kernel(const void *__restrict__ vals, int64_t n, int desc, unsigned int *__restrict__ out) {
  __shared__ unsigned long long skey[2 * OW_HEAVY_SAMPLES];
  __shared__ uint32_t scnt[2 * OW_HEAVY_SAMPLES];
  for (uint32_t i = threadIdx.x; i < 2 * OW_HEAVY_SAMPLES; i += 1024) {
    skey[i] = ~0ull;
    scnt[i] = 0;
  }
  __syncthreads();
  const int64_t stride = max(n / (int64_t)OW_HEAVY_SAMPLES, (int64_t)1);
  uint32_t best = 0;
  for (uint32_t s = threadIdx.x; s < OW_HEAVY_SAMPLES; s += 1024) {
    const int64_t row = min(n - 1, (int64_t)s * stride + (int64_t)(mix64((uint64_t)s) % (uint64_t)stride));
    unsigned long long k = order_image<KIND>(vals, row, desc);
    if (k == ~0ull) k = ~1ull; // (~0 marks a free slot)
    uint32_t hsh = (uint32_t)mix64(k) & (2 * OW_HEAVY_SAMPLES - 1);
    for (;;) {
      const unsigned long long old = atomicCAS(&skey[hsh], ~0ull, k);
      if (old == ~0ull || old == k) {
        best = max(best, atomicAdd(&scnt[hsh], 1u) + 1u);
        break;
      }
      hsh = (hsh + 1) & (2 * OW_HEAVY_SAMPLES - 1);
    }
  }
  for (int sft = 32; sft >= 1; sft >>= 1) best = max(best, (uint32_t)__shfl_xor((int)best, sft, 64));
  if (lane_id() == 0) atomicMax(out, best);
}

constexpr int OW_MM_SLOTS = 32; // {min, max} pairs the blocks spread their atomics over; the host reduces them
__global__ void order_minmax_init_kernel(unsigned long long *mm) { // [2 * OW_MM_SLOTS + 2]: {~0, 0} pairs, the flag word, the inversion word
  const int i = threadIdx.x;
  if (i <= 2 * OW_MM_SLOTS + 1) mm[i] = (i < 2 * OW_MM_SLOTS && !(i & 1)) ? ~0ull : 0ull;
}
template <int KIND>
__global__ __launch_bounds__(256) void order_minmax_kernel(const void *__restrict__ vals, int64_t n, int desc,
                                                           unsigned long long *mm, int every) {
  uint64_t lo = ~0ull, hi = 0;
  constexpr int KU = 8;
  if (every > 1 && blockIdx.x == 0) {
#pragma unroll
    for (int u = 0; u < KU; u++) {
      const uint64_t k = order_image<KIND>(vals, max((int64_t)0, n - 1 - ((int64_t)threadIdx.x * KU + u)), desc);
      lo = min(lo, k);
      hi = max(hi, k);
    }
  }
  for (int64_t base = blockIdx.x * (int64_t)every * (256 * KU) + threadIdx.x; base < n;
       base += (int64_t)gridDim.x * every * (256 * KU)) {
    uint64_t k[KU];
#pragma unroll
    for (int u = 0; u < KU; u++) k[u] = order_image<KIND>(vals, min(base + u * 256, n - 1), desc);
#pragma unroll
    for (int u = 0; u < KU; u++) {
      lo = min(lo, k[u]);
      hi = max(hi, k[u]);
    }
  }
  lo = wave_min_u64(lo);
  hi = wave_max_u64(hi);
  __shared__ unsigned long long s_lo[4], s_hi[4];
  if (lane_id() == 0) {
    s_lo[wave_id()] = lo;
    s_hi[wave_id()] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) {
      lo = min(lo, (uint64_t)s_lo[w]);
      hi = max(hi, (uint64_t)s_hi[w]);
    }
    // (one of OW_MM_SLOTS pairs: thousands of blocks on ONE address serialise, key statistics of agg_partition.hip)
    unsigned long long *slot = mm + 2 * (blockIdx.x % OW_MM_SLOTS);
    atomicMin(slot, (unsigned long long)lo);
    atomicMax(slot + 1, (unsigned long long)hi);
  }
}

// ---- stable 8-bit multi-split over HBM, rows = (word, carried column) -------------------------------
#ifndef OW_ITEMS_N
#define OW_ITEMS_N 8 // (rows per thread of the split passes' tile; tools/order_two_builds.py compares builds.  Round 6, one process,
                     //  1e8 rows, 12 against 8 (profiles/r06q_order_items12_ab.txt): narrow route split phase 1.68-1.71 vs 1.73-1.75 ms,
                     //  but the counting form 2.48 vs 2.32, three columns 4.72 vs 4.65, doubles 3.92 vs 3.82; 6, 10 and 16 lose
                     //  everywhere: 8 stays)
#endif
constexpr int OW_WG = 512, OW_WAVES = OW_WG / 64, OW_ITEMS = OW_ITEMS_N, OW_TILE = OW_WG * OW_ITEMS;
constexpr int OW_TWO_WGS = OW_ITEMS <= 8 ? 3 : 2; // workgroups per CU the TWO form is compiled for (its one LDS tile: 8 bytes x OW_TILE)

// RAW: the pass reads the raw key column (row id = position) and builds  off << 32 | row  in registers
template <int KIND, bool RAW, bool REC_IN = false>
__device__ __forceinline__ uint64_t ow_word(const void *__restrict__ src, int64_t i, int desc, uint64_t imin) {
  if (RAW) return ((order_image<KIND>(src, i, desc) - imin) << 32) | (uint64_t)(uint32_t)i;
  return __builtin_nontemporal_load((const uint64_t *)src + (REC_IN ? 2 * i : i)); // (REC_IN: the word of a {word, value} record)
}

// TILED: the pass runs over the tile list `tiles` (tiles aligned to the digit segments of the previous pass, see
// ow_tile_plan_kernel) instead of over fixed 4096-row blocks; a list entry of length 0 is a spare slot.
struct OwTile {
  int64_t start;
  uint32_t len, pad;
};
template <bool TILED>
__device__ __forceinline__ void ow_tile_of(const OwTile *__restrict__ tiles, int64_t n, int64_t &t0, uint32_t &tl) {
  if (TILED) {
    t0 = tiles[blockIdx.x].start;
    tl = tiles[blockIdx.x].len;
  } else {
    t0 = (int64_t)blockIdx.x * OW_TILE;
    tl = (uint32_t)min<int64_t>(OW_TILE, n - t0);
  }
}

// `oob` (RAW pass with an optimistic key range only): set when a key lies outside [imin, imin + 2^kbits) — the word
// cannot hold its offset, nothing of the attempt is valid
template <int KIND, bool RAW, bool TILED = false, bool REC_IN = false>
__global__ __launch_bounds__(OW_WG) void ow_hist_kernel(const void *__restrict__ src, int64_t n, int desc, uint64_t imin,
                                                        int shift, int64_t nblocks, uint32_t *__restrict__ hist,
                                                        const OwTile *__restrict__ tiles, unsigned int *__restrict__ oob = nullptr,
                                                        int kbits = 32) {
  __shared__ uint32_t h[256];
  int64_t t0;
  uint32_t tl;
  ow_tile_of<TILED>(tiles, n, t0, tl);
  if (TILED && tl == 0) {
    if (threadIdx.x < 256) hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = 0;
    return;
  }
  if (threadIdx.x < 256) h[threadIdx.x] = 0;
  uint64_t k[OW_ITEMS];
#pragma unroll
  for (int r = 0; r < OW_ITEMS; r++) k[r] = ow_word<KIND, RAW, REC_IN>(src, t0 + min((uint32_t)(threadIdx.x + r * OW_WG), tl - 1), desc, imin);
  if (RAW && oob) { // (the word keeps 32 bits of the offset: test the offset itself)
    bool bad = false;
#pragma unroll
    for (int r = 0; r < OW_ITEMS; r++) {
      const uint64_t off = order_image<KIND>(src, t0 + min((uint32_t)(threadIdx.x + r * OW_WG), tl - 1), desc) - imin;
      bad |= (off >> kbits) != 0;
    }
    if (__ballot(bad) && lane_id() == 0) atomicOr(oob, 1u);
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < OW_ITEMS; r++)
    if ((uint32_t)(threadIdx.x + r * OW_WG) < tl) atomicAdd(&h[(k[r] >> shift) & 255], 1u);
  __syncthreads();
  if (threadIdx.x < 256) hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// rank of every row among the rows of the same digit in its wave, in row order (wave w owns ITEMS chunks of
// 64 consecutive rows): lanes with the same digit find each other with 8 ballots, the first of them bumps the
// wave's own counter of that digit (one writer per digit and chunk, chunks in order: no atomics)
template <int ITEMS, bool SKIP_EMPTY = false>
__device__ __forceinline__ void stable_wave_ranks(const uint32_t (&dig)[ITEMS], const bool (&valid)[ITEMS],
                                                  uint32_t *__restrict__ wcnt_w /* [256] of this wave */,
                                                  uint32_t (&rnk)[ITEMS]) {
#pragma unroll
  for (int j = 0; j < ITEMS; j++) {
    uint64_t peers = __ballot(valid[j]);
    if (SKIP_EMPTY && !peers) { // (wave-uniform: a chunk without rows)
      rnk[j] = 0;
      continue;
    }
    // (lanes whose digit DIFFERS from this lane's in some bit, accumulated: mask = all ones where this lane's bit is set, so
    //  ballot ^ mask has a lane's bit set exactly when the two bits differ — a signed bit-field extract, a compare for the ballot,
    //  two XORs and v_or3 per two bits instead of a select between bm and ~bm: 1089 -> 905 VALU instructions per tile of the
    //  tiled scatter.  Measured in one process against the select form (tools/order_two_builds.py, profiles/r06q_order_rank_ab.txt):
    //  narrow route split phase 1.704 vs 1.688 ms, splitter route 2.27 vs 2.30 — NO effect either way: the passes do not run at
    //  the pace of their ranking instructions, contrary to the round-3 note below.)
#if defined(OW_RANK_SELECT) // (the form of rounds 3-5, kept for A/B builds: tools/order_two_builds.py)
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const bool bit = (dig[j] >> b) & 1;
      const uint64_t bm = __ballot(bit);
      peers &= bit ? bm : ~bm;
    }
#else
    uint32_t dlo = 0, dhi = 0;
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const int32_t mask = ((int32_t)(dig[j] << (31 - b))) >> 31; // 0 or -1
      const uint64_t bm = __ballot(mask != 0);
      dlo |= (uint32_t)bm ^ (uint32_t)mask;
      dhi |= (uint32_t)(bm >> 32) ^ (uint32_t)mask;
    }
    peers &= ~(((uint64_t)dhi << 32) | dlo);
#endif
    const uint32_t r = (uint32_t)mbcnt(peers);
    uint32_t old = 0;
    if (valid[j] && r == 0) {
      old = wcnt_w[dig[j]];
      wcnt_w[dig[j]] = old + (uint32_t)__popcll(peers);
    }
    old = (uint32_t)__shfl((int)old, valid[j] ? __builtin_ctzll(peers) : 0, 64);
    rnk[j] = old + r;
  }
}

// (A persistent, software-pipelined form of this kernel — contiguous tile range per workgroup, the next tile's rows
// loaded into a second register set, unconditional loads / stores as in rp_scatter_kernel — measured SLOWER: 2.49
// against 2.35 ms for the two passes of 1e8 rows.  The pass is co-limited by the stable ranking: ~50 VALU
// instructions per row slot and wave for the 8 ballots, i.e. ~70 % of the SIMD time two resident workgroups have per
// tile at the HBM rate, so a second register set (128 VGPRs, spills) buys nothing that the second resident
// workgroup does not already provide.)
// REC (NPAY == 1): the pass writes {word, carried value} records into `words_out` (16 B per row) — what the
// in-LDS finish reads; a (tile, digit) run of 16 rows is one 256-byte piece instead of 128 B in each of two columns
// REC_IN: the previous pass wrote records (both HBM passes of the usual two then move one 16-byte piece per row)
// LB (round 5): no count matrix — the pass takes its digit bases from ONE histogram of the whole column (ow_ghist_kernel,
// `ghist` = the 256 counts of this pass's digit) and a tile's offset inside a digit from a chained scan over the tiles
// in launch order: digit d of tile t publishes {AGG | count}, walks back over the descriptors of tiles t-1, t-2, ... until
// one carries an inclusive prefix, publishes {PFX | prefix + count} (one u32 per (tile, digit): flag and value in one
// word, agent-scope relaxed accesses — nothing to order).  The walk is issued before the rows are staged in LDS and
// consumed behind it.  Workgroups start in index order, so every predecessor of a running tile is running or done; the
// spin is bounded all the same and raises `lb_fail`, on which the host takes the counting form.  The first tile of
// segment `lo` of the tiled pass also leaves bound[lo][d] = where the segment's rows of digit d begin: the group table.
constexpr uint32_t OLB_AGG = 1u << 30, OLB_PFX = 2u << 30, OLB_VAL = (1u << 30) - 1u;
__device__ __forceinline__ uint32_t olb_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void olb_store(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// TWO (records out): word and carried value take turns in ONE LDS tile instead of two — 43 KB and <= 80 VGPRs: three
// workgroups per CU instead of two, for two more barriers per tile
// SLIM (round 6, with TWO): the records between the passes and into the finish are 12 bytes — {key offset (u32), carried value} —
// instead of 16.  The low half of the word, the row id, is dead weight on this route whenever the caller does not ask for the
// permutation: both passes and the finish's LDS passes are STABLE (stable_wave_ranks, tiles chained in launch order), so equal
// keys keep their input order without it.  1.6 GB less per 1e8 rows over the three passes (10.4 -> 8.8).
struct __attribute__((aligned(4))) OwRec12 {
  uint32_t off, vlo, vhi;
};
__device__ __forceinline__ void ow_rec12_load(const void *__restrict__ recs, int64_t i, uint64_t &word, uint64_t &val) {
  const uint32_t *p = (const uint32_t *)recs + 3 * i;
  const uint32_t o = __builtin_nontemporal_load(p), lo = __builtin_nontemporal_load(p + 1), hi = __builtin_nontemporal_load(p + 2);
  word = (uint64_t)o << 32;
  val = (uint64_t)lo | ((uint64_t)hi << 32);
}
__device__ __forceinline__ void ow_rec12_store(void *__restrict__ recs, int64_t i, uint64_t word, uint64_t val) {
  OwRec12 r;
  r.off = (uint32_t)(word >> 32);
  r.vlo = (uint32_t)val;
  r.vhi = (uint32_t)(val >> 32);
  ((OwRec12 *)recs)[i] = r;
}
template <int KIND, bool RAW, int NPAY, bool TILED = false, bool REC = false, bool REC_IN = false, bool LB = false, bool TWO = false,
          bool SLIM = false>
__global__ __launch_bounds__(OW_WG, TWO ? OW_TWO_WGS : 1) void ow_scatter_kernel(const void *__restrict__ src, const uint64_t *__restrict__ pay,
                                                           int64_t n, int desc, uint64_t imin, int shift, int64_t nblocks,
                                                           const uint32_t *__restrict__ offsets,
                                                           uint64_t *__restrict__ words_out, uint64_t *__restrict__ pay_out,
                                                           const OwTile *__restrict__ tiles,
                                                           const unsigned int *__restrict__ abort_flag = nullptr,
                                                           const uint32_t *__restrict__ ghist = nullptr,
                                                           uint32_t *__restrict__ lbdesc = nullptr,
                                                           uint32_t *__restrict__ bound = nullptr,
                                                           unsigned int *__restrict__ lb_fail = nullptr) {
  // (optimistic key range: the raw pass's histogram kernel has already seen every key; once it raised the flag, nothing
  //  this attempt produces is used — a miss then costs that histogram pass, not the two split passes behind it)
  if (abort_flag && *abort_flag) return;
  // (advisor r05, medium) a look-back spin that ran out in the FIRST split pass leaves overlapped and never-written records
  // behind; the tiled pass would count its digits from those stale words against the bases of the TRUE histogram and store
  // past its output — it must not run at all.  (Inside one pass a failed spin only shortens a prefix: positions stay in range.)
  if (LB && lb_fail && *lb_fail) return;
  static_assert(!TWO || (REC && NPAY == 1), "TWO: the record form");
  static_assert(!SLIM || (TWO && LB), "SLIM: 12-byte records of the look-back form");
  __shared__ uint64_t sword[OW_TILE];
  __shared__ uint64_t spay[NPAY && !TWO ? OW_TILE : 1];
  __shared__ uint32_t wcnt[OW_WAVES][256];
  __shared__ uint32_t dstart[256];
  __shared__ int64_t gbase[256];
  __shared__ uint32_t s_wsum[4];
  __shared__ uint32_t s_gsum[4];
  const int w = wave_id(), lane = lane_id();
  int64_t tbase;
  uint32_t len;
  ow_tile_of<TILED>(tiles, n, tbase, len);
  if (TILED && len == 0) return; // (spare slots: behind every tile that brings rows)
  const uint32_t wrow = (uint32_t)w * (OW_ITEMS * 64) + lane; // element of the tile
  uint64_t k[OW_ITEMS], v[NPAY ? OW_ITEMS : 1];
#pragma unroll
  for (int j = 0; j < OW_ITEMS; j++) {
    const int64_t i = tbase + min(wrow + j * 64, len - 1);
    if (REC_IN) {
      if constexpr (SLIM) {
        ow_rec12_load(src, i, k[j], v[NPAY ? j : 0]);
        continue;
      }
      const u64x2 rec = __builtin_nontemporal_load((const u64x2 *)src + i);
      k[j] = rec.x;
      v[NPAY ? j : 0] = rec.y;
      continue;
    }
    k[j] = ow_word<KIND, RAW>(src, i, desc, imin);
    if (NPAY) v[j] = __builtin_nontemporal_load(pay + i);
  }
  const uint32_t goff = !LB && threadIdx.x < 256 ? offsets[(int64_t)threadIdx.x * nblocks + blockIdx.x] : 0;
  const uint32_t gcnt = LB && threadIdx.x < 256 ? ghist[threadIdx.x] : 0;
#pragma unroll
  for (int q = 0; q < 4; q++) wcnt[w][lane + 64 * q] = 0;
  uint32_t dig[OW_ITEMS], rnk[OW_ITEMS];
  bool valid[OW_ITEMS];
#pragma unroll
  for (int j = 0; j < OW_ITEMS; j++) {
    valid[j] = wrow + j * 64 < len;
    dig[j] = (uint32_t)(k[j] >> shift) & 255u;
  }
  stable_wave_ranks<OW_ITEMS>(dig, valid, wcnt[w], rnk);
  __syncthreads();
  uint32_t my_cnt = 0, my_ds = 0, gex = 0, lb_first = 0; // (threads < 256: digit threadIdx.x of this tile)
  uint32_t *my_desc = LB ? lbdesc + (size_t)blockIdx.x * 256 + min(threadIdx.x, 255u) : nullptr;
  if (threadIdx.x < 256) { // wave counters -> exclusive prefix over the waves; scan over the digits
    uint32_t acc = 0;
#pragma unroll
    for (int q = 0; q < OW_WAVES; q++) {
      uint32_t c = wcnt[q][threadIdx.x];
      wcnt[q][threadIdx.x] = acc;
      acc += c;
    }
    if (LB) { // publish the count, ask for the predecessor's word (consumed behind the staging loop)
      olb_store(my_desc, (blockIdx.x == 0 ? OLB_PFX : OLB_AGG) | acc);
      if (blockIdx.x > 0) lb_first = olb_load(my_desc - 256);
      my_cnt = acc;
      const uint32_t ginc = wave_iscan_u32(gcnt);
      if (lane == 63) s_gsum[w] = ginc;
      gex = ginc - gcnt;
    }
    uint32_t inc = wave_iscan_u32(acc);
    if (lane == 63) s_wsum[w] = inc;
    dstart[threadIdx.x] = inc - acc;
  }
  __syncthreads();
  if (threadIdx.x < 256) {
    uint32_t wb = 0;
    for (int q = 0; q < w; q++) wb += s_wsum[q];
    const uint32_t ds = dstart[threadIdx.x] + wb;
    dstart[threadIdx.x] = ds;
    if (LB) {
      my_ds = ds;
      for (int q = 0; q < w; q++) gex += s_gsum[q];
    } else
      gbase[threadIdx.x] = (int64_t)goff - (int64_t)ds;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < OW_ITEMS; j++) {
    if (!valid[j]) continue;
    const uint32_t p = dstart[dig[j]] + wcnt[w][dig[j]] + rnk[j];
    sword[p] = k[j];
    if (TWO) rnk[j] = p; // (kept for the value's turn)
    else if (NPAY) spay[p] = v[j];
  }
  if (LB && threadIdx.x < 256) {
    uint32_t excl = 0;
    if (blockIdx.x > 0) {
      const uint32_t *p = my_desc - 256;
      uint32_t st = lb_first;
      for (;;) {
        unsigned spins = 0;
        while ((st >> 30) == 0) { // not published yet
          if (++spins > LB_SPIN_LIMIT) {
            *lb_fail = 1u;
            st = OLB_PFX;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
          st = olb_load(p);
        }
        excl += st & OLB_VAL;
        if ((st >> 30) == 2 || p == lbdesc + threadIdx.x) break;
        p -= 256;
        st = olb_load(p);
      }
      olb_store(my_desc, OLB_PFX | ((excl + my_cnt) & OLB_VAL));
    }
    gbase[threadIdx.x] = (int64_t)gex + (int64_t)excl - (int64_t)my_ds;
    if (TILED && bound && tiles[blockIdx.x].pad) bound[(size_t)(tiles[blockIdx.x].pad - 1) * 256 + threadIdx.x] = gex + excl;
  }
  __syncthreads();
  if constexpr (TWO) {
#pragma unroll
    for (int j = 0; j < OW_ITEMS; j++) k[j] = sword[min((uint32_t)(j * OW_WG) + threadIdx.x, (uint32_t)(OW_TILE - 1))]; // the words in output order
    __syncthreads();
#pragma unroll
    for (int j = 0; j < OW_ITEMS; j++)
      if (valid[j]) sword[rnk[j]] = v[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < OW_ITEMS; j++) {
      const uint32_t p = j * OW_WG + threadIdx.x;
      if (p < len) {
        const int64_t g = gbase[(uint32_t)(k[j] >> shift) & 255u] + p;
        if constexpr (SLIM) {
          ow_rec12_store(words_out, g, k[j], sword[p]);
          continue;
        }
        u64x2 rec;
        rec.x = k[j];
        rec.y = sword[p];
        ((u64x2 *)words_out)[g] = rec;
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < OW_ITEMS; j++) {
    const uint32_t p = j * OW_WG + threadIdx.x;
    if (p < len) {
      const uint64_t kk = sword[p];
      const int64_t g = gbase[(uint32_t)(kk >> shift) & 255u] + p;
      if (REC) {
        u64x2 rec;
        rec.x = kk;
        rec.y = spay[NPAY ? p : 0];
        ((u64x2 *)words_out)[g] = rec;
        continue;
      }
      words_out[g] = kk;
      if (NPAY) pay_out[g] = spay[p];
    }
  }
}

// ---- segment-aligned tiles for the last HBM pass ---------------------------------------------------------
// After the first pass the rows are grouped by its digit (segment d = rows [S[d], S[d+1]), S read from the
// scanned count matrix of that pass).  The last pass runs over tiles that never cross a segment boundary, so
// the rows of group (hi, lo) — digit `hi` of the last pass, digit `lo` of the first — are exactly the rows the
// tiles of segment `lo` send to digit `hi`: their position range falls out of the last pass's own scanned
// count matrix (ow_group_table_kernel) and the sorted rows need not be read again to find the boundaries.
__global__ __launch_bounds__(256) void ow_tile_plan_kernel(const uint32_t *__restrict__ offs1, int64_t nblocks1, int64_t n,
                                                           uint32_t *__restrict__ firsttile /* [257] */,
                                                           int64_t *__restrict__ segstart /* [257] */) {
  __shared__ uint32_t s_w[4];
  const uint32_t d = threadIdx.x;
  const int64_t lo = offs1[(int64_t)d * nblocks1], hi = d == 255 ? n : (int64_t)offs1[(int64_t)(d + 1) * nblocks1];
  const uint32_t nt = (uint32_t)((hi - lo + OW_TILE - 1) / OW_TILE);
  const uint32_t inc = wave_iscan_u32(nt);
  if (lane_id() == 63) s_w[wave_id()] = inc;
  __syncthreads();
  uint32_t wb = 0;
  for (int q = 0; q < wave_id(); q++) wb += s_w[q];
  firsttile[d] = wb + inc - nt;
  segstart[d] = lo;
  if (d == 255) {
    firsttile[256] = wb + inc;
    segstart[256] = n;
  }
}
__global__ void ow_tile_fill_kernel(const uint32_t *__restrict__ firsttile, const int64_t *__restrict__ segstart, uint32_t ntmax,
                                    OwTile *__restrict__ tiles) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntmax) return;
  OwTile o;
  o.start = 0;
  o.len = 0;
  o.pad = 0;
  if (t < firsttile[256]) {
    uint32_t lo = 0, hi = 256; // last segment whose first tile is <= t (segments without tiles share their successor's)
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (firsttile[mid] <= t) lo = mid; else hi = mid;
    }
    o.start = segstart[lo] + (int64_t)(t - firsttile[lo]) * OW_TILE;
    o.len = (uint32_t)min<int64_t>(OW_TILE, segstart[lo + 1] - o.start);
    o.pad = t == firsttile[lo] ? lo + 1 : 0; // (segment + 1 on the segment's first tile: the look-back form's group bounds)
  }
  tiles[t] = o;
}
// group g = hi << 8 | lo: rows [gstart[g], gend[g]) of the last pass's output; one block per value of `hi`;
// gend[number of groups] = largest group
__global__ __launch_bounds__(256) void ow_group_table_kernel(const uint32_t *__restrict__ offs2, int64_t ntmax,
                                                             const uint32_t *__restrict__ firsttile, int64_t n,
                                                             uint32_t *__restrict__ gstart, uint32_t *__restrict__ gend) {
  const uint32_t hi = blockIdx.x, lo = threadIdx.x;
  // position of the first row that the tiles from `firsttile[lo]` on send to digit hi
  auto at = [&](uint32_t h, uint32_t t) -> uint32_t {
    if (t >= (uint32_t)ntmax) { h++; t = 0; }
    return h >= 256 ? (uint32_t)n : offs2[(int64_t)h * ntmax + t];
  };
  const uint32_t a = at(hi, firsttile[lo]);
  const uint32_t b = lo == 255 ? at(hi + 1, 0) : at(hi, firsttile[lo + 1]);
  gstart[hi * 256 + lo] = a;
  gend[hi * 256 + lo] = b;
  uint32_t sz = b - a;
  for (int k = 32; k >= 1; k >>= 1) sz = max(sz, (uint32_t)__shfl_xor((int)sz, k, 64));
  if (lane_id() == 0 && sz) atomicMax(gend + (size_t)gridDim.x * 256, sz);
}

// ---- look-back form of the two split passes: one histogram, no count matrices ---------------------------------------
// ghist[0..255] = rows per digit of the first pass (bits [shift_lo, +8) of the word), ghist[256..511] = of the second;
// one read of the raw column by a persistent grid (a block per tile would put 2.5e4 x 512 atomics on 512 words), which also
// tests every key against an optimistic range (`oob`, see ow_hist_kernel)
template <int KIND>
__global__ __launch_bounds__(OW_WG) void ow_ghist_kernel(const void *__restrict__ src, int64_t n, int desc, uint64_t imin,
                                                         int shift_lo, int shift_hi, int64_t nblocks, uint32_t *__restrict__ ghist,
                                                         unsigned int *__restrict__ oob, int kbits) {
  __shared__ uint32_t h[512];
  static_assert(OW_WG == 512, "one counter per thread");
  h[threadIdx.x] = 0;
  __syncthreads();
  bool bad = false;
  for (int64_t t = blockIdx.x; t < nblocks; t += gridDim.x) {
    const int64_t t0 = t * OW_TILE;
    const uint32_t tl = (uint32_t)min<int64_t>(OW_TILE, n - t0);
    uint64_t off[OW_ITEMS];
#pragma unroll
    for (int r = 0; r < OW_ITEMS; r++) off[r] = order_image<KIND>(src, t0 + min((uint32_t)(threadIdx.x + r * OW_WG), tl - 1), desc) - imin;
#pragma unroll
    for (int r = 0; r < OW_ITEMS; r++) {
      if ((uint32_t)(threadIdx.x + r * OW_WG) >= tl) continue;
      if (oob) bad |= (off[r] >> kbits) != 0;
      const uint64_t wd = off[r] << 32;
      atomicAdd(&h[(wd >> shift_lo) & 255], 1u);
      atomicAdd(&h[256 + ((wd >> shift_hi) & 255)], 1u);
    }
  }
  if (oob && __ballot(bad) && lane_id() == 0) atomicOr(oob, 1u);
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&ghist[threadIdx.x], h[threadIdx.x]);
}
// ow_tile_plan_kernel from the first pass's 256 digit counts instead of its scanned count matrix
__global__ __launch_bounds__(256) void ow_tile_plan_gh_kernel(const uint32_t *__restrict__ ghist, int64_t n,
                                                              uint32_t *__restrict__ firsttile /* [257] */,
                                                              int64_t *__restrict__ segstart /* [257] */) {
  __shared__ uint32_t s_w[4], s_c[4];
  const uint32_t d = threadIdx.x, c = ghist[d];
  const uint32_t nt = (c + OW_TILE - 1) / OW_TILE;
  const uint32_t inc = wave_iscan_u32(nt), cinc = wave_iscan_u32(c);
  if (lane_id() == 63) {
    s_w[wave_id()] = inc;
    s_c[wave_id()] = cinc;
  }
  __syncthreads();
  uint32_t wb = 0, cb = 0;
  for (int q = 0; q < wave_id(); q++) {
    wb += s_w[q];
    cb += s_c[q];
  }
  firsttile[d] = wb + inc - nt;
  segstart[d] = (int64_t)(cb + cinc - c);
  if (d == 255) {
    firsttile[256] = wb + inc;
    segstart[256] = n;
  }
}
// ow_group_table_kernel from the bounds the first tile of every segment left: group (hi, lo) = [bound[lo][hi], bound[lo'][hi])
// with lo' the next segment that has rows, or the end of digit hi; out[1] = largest group.  One block per value of `hi`.
__global__ __launch_bounds__(256) void ow_group_table_lb_kernel(const uint32_t *__restrict__ bound, const uint32_t *__restrict__ ghist,
                                                                uint32_t *__restrict__ gstart, uint32_t *__restrict__ gend,
                                                                unsigned int *__restrict__ out) {
  if (out[0]) return; // (a look-back spin ran out: `bound` was never written; the host discards the attempt)
  __shared__ uint32_t s_seg[256], s_w[4];
  __shared__ uint32_t s_end;
  const uint32_t hi = blockIdx.x, lo = threadIdx.x;
  s_seg[lo] = ghist[lo];
  const uint32_t c = ghist[256 + lo];
  const uint32_t inc = wave_iscan_u32(c);
  if (lane_id() == 63) s_w[wave_id()] = inc;
  __syncthreads();
  uint32_t wb = 0;
  for (int q = 0; q < wave_id(); q++) wb += s_w[q];
  if (lo == hi) s_end = wb + inc; // end of digit hi
  __syncthreads();
  uint32_t nx = lo + 1;
  while (nx < 256 && s_seg[nx] == 0) nx++;
  const uint32_t b = nx < 256 ? bound[(size_t)nx * 256 + hi] : s_end;
  const uint32_t a = s_seg[lo] ? bound[(size_t)lo * 256 + hi] : b;
  gstart[hi * 256 + lo] = a;
  gend[hi * 256 + lo] = b;
  uint32_t sz = b - a;
  for (int k = 32; k >= 1; k >>= 1) sz = max(sz, (uint32_t)__shfl_xor((int)sz, k, 64));
  if (lane_id() == 0 && sz) atomicMax(out + 1, sz);
}

// ---- group boundaries ---------------------------------------------------------------------------------
// rows are grouped by their top bits: a row whose predecessor belongs to another group opens its group and
// closes the predecessor's (gstart / gend start as ~0 / 0: an absent group keeps gstart = ~0)
__global__ __launch_bounds__(256) void ow_group_bounds_kernel(const uint64_t *__restrict__ words, int64_t n, int gshift,
                                                              uint32_t *__restrict__ gstart, uint32_t *__restrict__ gend) {
  constexpr int KU = 8; // independent loads in flight per lane
  for (int64_t base = blockIdx.x * (int64_t)(256 * KU) + threadIdx.x; base < n; base += (int64_t)gridDim.x * (256 * KU)) {
    uint64_t cur[KU], prev[KU];
#pragma unroll
    for (int u = 0; u < KU; u++) {
      const int64_t i = min(base + u * 256, n - 1);
      cur[u] = __builtin_nontemporal_load(words + i);
      prev[u] = words[i > 0 ? i - 1 : 0]; // (the neighbouring lane's element: served by the same lines)
    }
#pragma unroll
    for (int u = 0; u < KU; u++) {
      const int64_t i = base + u * 256;
      if (i >= n) continue;
      const uint32_t g = (uint32_t)(cur[u] >> gshift), gp = (uint32_t)(prev[u] >> gshift);
      if (i == 0 || gp != g) {
        gstart[g] = (uint32_t)i;
        if (i) gend[gp] = (uint32_t)i;
      }
      if (i == n - 1) gend[g] = (uint32_t)n;
    }
  }
}
__global__ void ow_group_max_kernel(const uint32_t *__restrict__ gstart, const uint32_t *__restrict__ gend, uint32_t G,
                                    uint32_t *__restrict__ max_group) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t sz = (g < G && gstart[g] != 0xffffffffu) ? gend[g] - gstart[g] : 0;
  for (int k = 32; k >= 1; k >>= 1) sz = max(sz, (uint32_t)__shfl_xor((int)sz, k, 64));
  if (lane_id() == 0 && sz) atomicMax(max_group, sz);
}

// ---- finish: sort every group on its low bits inside LDS, write the final columns ----------------------
constexpr int FIN_WG = 256, FIN_WAVES = FIN_WG / 64;
constexpr uint32_t FIN_BUCKET_CAP = 24; // rows of the largest bucket the bucket + count form of the finish ranks by counting
template <int KIND, int NPAY, int R, bool REC = false, bool SLIM = false>
__global__ __launch_bounds__(FIN_WG) void ow_finish_kernel(const uint64_t *__restrict__ words, const uint64_t *__restrict__ pay,
                                                           const uint32_t *__restrict__ gstart,
                                                           const uint32_t *__restrict__ gend, int rbits, int desc,
                                                           uint64_t imin, void *__restrict__ key_out,
                                                           uint64_t *__restrict__ pay_out, uint32_t *__restrict__ perm_out,
                                                           int count_form = 1) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lo = gstart[blockIdx.x], hi = gend[blockIdx.x];
  if (lo == 0xffffffffu || lo >= hi) return; // no row carries these top bits
  const uint32_t m = hi - lo;
  uint64_t *sword = (uint64_t *)smem;                       // [R * FIN_WG]
  uint64_t *spay = sword + (size_t)R * FIN_WG;              // [NPAY ? R * FIN_WG : 0]
  uint32_t *wcnt = (uint32_t *)(spay + (NPAY ? (size_t)R * FIN_WG : 0)); // [FIN_WAVES][256]
  uint32_t *dstart = wcnt + FIN_WAVES * 256;                // [256]
  __shared__ uint32_t s_wsum[FIN_WAVES];
  const int w = wave_id(), lane = lane_id();
  // element e = (w * cpw + j) * 64 + lane: wave w owns cpw <= R chunks of 64 consecutive rows — as many as give all four
  // waves the same share of THIS group (with R per wave, a group of 1500 rows kept three waves busy and one idle)
  const uint32_t cpw = ((m + 63) / 64 + FIN_WAVES - 1) / FIN_WAVES;
  uint64_t k[R], v[NPAY ? R : 1];
  bool valid[R];
#pragma unroll
  for (int j = 0; j < R; j++) {
    const uint32_t e = (uint32_t)(w * cpw + j) * 64 + lane;
    valid[j] = (uint32_t)j < cpw && e < m;
    const uint32_t i = lo + min(e, m - 1);
    if (REC && SLIM) { // 12-byte records {key offset, value}: no row id (perm_out == nullptr)
      ow_rec12_load(words, i, k[j], v[NPAY ? j : 0]);
    } else if (REC) {
      const u64x2 rec = __builtin_nontemporal_load((const u64x2 *)words + i);
      k[j] = rec.x;
      v[NPAY ? j : 0] = rec.y;
    } else {
      k[j] = __builtin_nontemporal_load(words + i);
      if (NPAY) v[j] = __builtin_nontemporal_load(pay + i);
    }
  }
  // Round 6 — more than 8 low bits (two LSD passes below, ~60 VALU instructions per row and pass for the 8 ballots of the stable
  // ranking: the finish was bound by them, not by its bytes): BUCKET + COUNT instead.  The group's ~1.5 K rows fall into up to
  // 1024 buckets by the top bits of what is left of the key (one LDS atomic per row, one scan of the counts), take any free slot of
  // their bucket (a second atomic: the order inside a bucket is whatever the LDS unit makes it), and every row then counts the
  // entries of its bucket that are smaller than its own {low bits | position in the group} — one to a handful of 4-byte LDS reads.
  // The position makes the entries distinct, so the ranks are a permutation and equal keys keep their input order whatever
  // order the atomics ran in.  A bucket of more than FIN_BUCKET_CAP rows (keys that repeat a lot) sends the group through
  // the LSD passes as before (workgroup-uniform).  SQLRS_ORDER_FINISH_COUNT=0 (host, read per call): the LSD passes always.
  bool placed = false;
  if (count_form && rbits > 8) {
    const int nbb = rbits < 10 ? rbits : 10, lowb = rbits - nbb; // (rbits <= 16: lowb <= 6, entries of <= 19 bits)
    const uint32_t NB = 1u << nbb, per = NB / FIN_WG;           // (NB = 512 or 1024: 2 or 4 counters per thread)
    uint32_t *A = wcnt + 1; // A[b] (A[-1] = 0): count -> start -> end of bucket b; 1025 words of the 1280 wcnt + dstart hold
    uint32_t *sbuf = (uint32_t *)sword; // the buckets' entries (the words / values take the space over afterwards)
    __shared__ uint32_t s_maxb;
    for (uint32_t q = threadIdx.x; q <= NB; q += FIN_WG) wcnt[q] = 0;
    if (threadIdx.x == 0) s_maxb = 0;
    __syncthreads();
    uint32_t bk[R], ent[R];
#pragma unroll
    for (int j = 0; j < R; j++) {
      const uint32_t rel = (uint32_t)(k[j] >> 32) & ((1u << rbits) - 1u);
      bk[j] = rel >> lowb;
      ent[j] = ((rel & ((1u << lowb) - 1u)) << 13) | ((uint32_t)(w * cpw + j) * 64 + lane); // (position < 6144 < 2^13)
      if (valid[j]) atomicAdd(&A[bk[j]], 1u);
    }
    __syncthreads();
    {
      uint32_t c[4], sum = 0, mx = 0;
#pragma unroll
      for (uint32_t i = 0; i < 4; i++) {
        c[i] = i < per ? A[threadIdx.x * per + i] : 0u;
        sum += c[i];
        mx = max(mx, c[i]);
      }
      const uint32_t inc = wave_iscan_u32(sum);
      if (lane == 63) s_wsum[w] = inc;
      for (int q = 32; q >= 1; q >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, q, 64));
      if (lane == 0 && mx > FIN_BUCKET_CAP) s_maxb = mx; // (any writer will do)
      __syncthreads();
      uint32_t run = inc - sum;
      for (int q = 0; q < w; q++) run += s_wsum[q];
#pragma unroll
      for (uint32_t i = 0; i < 4; i++) {
        if (i < per) A[threadIdx.x * per + i] = run;
        run += c[i];
      }
    }
    __syncthreads();
    if (s_maxb == 0) {
#pragma unroll
      for (int j = 0; j < R; j++)
        if (valid[j]) sbuf[atomicAdd(&A[bk[j]], 1u)] = ent[j];
      __syncthreads();
#pragma unroll
      for (int j = 0; j < R; j++) {
        if (!valid[j]) continue;
        const uint32_t s0 = A[(int)bk[j] - 1], s1 = A[bk[j]]; // (after the placement A[b] is the END of bucket b)
        uint32_t r = s0;
        for (uint32_t q = s0; q < s1; q++) r += sbuf[q] < ent[j];
        bk[j] = r;
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < R; j++) {
        if (!valid[j]) continue;
        sword[bk[j]] = k[j];
        if (NPAY) spay[bk[j]] = v[j];
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < R; j++) {
        const uint32_t e = min((uint32_t)(w * cpw + j) * 64 + lane, m - 1);
        k[j] = sword[e];
        if (NPAY) v[j] = spay[e];
      }
      placed = true;
    }
  }
  for (int shift = 32; !placed && shift < 32 + rbits; shift += 8) { // stable LSD passes over the low key bits, all in LDS
    for (int q = lane; q < 256; q += 64) wcnt[w * 256 + q] = 0;
    uint32_t dig[R], rnk[R];
#pragma unroll
    for (int j = 0; j < R; j++) {
      // the last pass may cover fewer than 8 bits: bits >= 32 + rbits are equal inside the group
      dig[j] = (uint32_t)(k[j] >> shift) & 255u & ((shift + 8 > 32 + rbits) ? ((1u << (32 + rbits - shift)) - 1) : 255u);
    }
    stable_wave_ranks<R, true>(dig, valid, wcnt + w * 256, rnk);
    __syncthreads();
    { // FIN_WG == 256: one thread per digit
      uint32_t acc = 0;
#pragma unroll
      for (int q = 0; q < FIN_WAVES; q++) {
        const uint32_t c = wcnt[q * 256 + threadIdx.x];
        wcnt[q * 256 + threadIdx.x] = acc;
        acc += c;
      }
      const uint32_t inc = wave_iscan_u32(acc);
      if (lane == 63) s_wsum[w] = inc;
      dstart[threadIdx.x] = inc - acc;
    }
    __syncthreads();
    {
      uint32_t wb = 0;
      for (int q = 0; q < w; q++) wb += s_wsum[q];
      dstart[threadIdx.x] += wb;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < R; j++) {
      if (!valid[j]) continue;
      const uint32_t p = dstart[dig[j]] + wcnt[w * 256 + dig[j]] + rnk[j];
      sword[p] = k[j];
      if (NPAY) spay[p] = v[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < R; j++) {
      const uint32_t e = min((uint32_t)(w * cpw + j) * 64 + lane, m - 1);
      k[j] = sword[e];
      if (NPAY) v[j] = spay[e];
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < R; j++) {
    if (!valid[j]) continue;
    const uint32_t i = lo + (uint32_t)(w * cpw + j) * 64 + lane;
    const uint64_t val = order_unimage<KIND>((k[j] >> 32) + imin, desc);
    if (KIND == OKIND_I32) ((int32_t *)key_out)[i] = (int32_t)(int64_t)val;
    else ((uint64_t *)key_out)[i] = val;
    if (NPAY) pay_out[i] = v[j];
    if (perm_out) perm_out[i] = (uint32_t)k[j];
  }
}

// the key value of an output row: 4 bytes for an int32 key column, 8 otherwise
template <int KIND> __device__ __forceinline__ void order_store_key(void *__restrict__ key_out, int64_t i, uint64_t image, int desc) {
  const uint64_t val = order_unimage<KIND>(image, desc);
  if (KIND == OKIND_I32) ((int32_t *)key_out)[i] = (int32_t)(int64_t)val;
  else ((uint64_t *)key_out)[i] = val;
}

// rbits == 0: everything was sorted in HBM, the finish is a streaming unpack
template <int KIND, int NPAY>
__global__ void ow_unpack_kernel(const uint64_t *__restrict__ words, int64_t n, int desc, uint64_t imin,
                                 void *__restrict__ key_out, uint32_t *__restrict__ perm_out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t kw = words[i];
  const uint64_t val = order_unimage<KIND>((kw >> 32) + imin, desc);
  if (KIND == OKIND_I32) ((int32_t *)key_out)[i] = (int32_t)(int64_t)val;
  else ((uint64_t *)key_out)[i] = val;
  if (perm_out) perm_out[i] = (uint32_t)kw;
}

constexpr uint32_t FIN_CAP = 6144; // rows of the largest group the in-LDS finish takes (R = 24)

// ==== keys with more than 32 varying bits ==========================================================================
// (random 64-bit ids, nanosecond timestamps over months, float64 measurements: the word `off << 32 | row` cannot hold
// them, and fixed top bits make groups of wildly different sizes — the image of a float is its exponent first.)
// Splitters taken from a sorted SAMPLE replace the bits: G = 2^top groups (the same count the narrow route would use),
// 16 samples per group, sub[g] = sample 16 g (sub[0] = 0).  Group of a row = the last g with sub[g] <= off, found in two
// levels so that a level fits LDS and the passes stay 256-way multi-splits:
//   pass 1 (raw column -> word = off, 8 B, + payload): digit k = last TOP splitter (sub[k << 8]) <= off — a binary search
//          over <= 256 values in LDS;
//   pass 2 (tiles aligned to the segments of pass 1, so k is uniform per tile): digit d = last of sub[k << 8 | 0..255]
//          <= off.  MSD order: the counts of this pass are laid out segment after segment, digit-major inside the
//          segment, so that the one exclusive scan yields the position of every (tile, digit) run AND the group table;
//   finish (one workgroup per group g = k << 8 | d, <= FIN_CAP rows, balanced by construction whatever the distribution):
//          rel = off - sub[g] < the group's width; two stable 8-bit passes in LDS on the TOP 16 bits of rel, then the
//          rows with the same top bits — runs of one or two rows, the group's ~1.5 K rows fall onto 65 536 values — are
//          put in order by counting: position = run start + #(smaller-or-earlier words in the run).  A run longer than
//          OWK_WALK rows (heavy duplicates of nearly-equal keys) sends the group through LSD passes over all bits of rel.
//   heavy values (a run of equal splitters): a group of their own that is copied, not sorted — owk_topfirst_kernel.
// The payload is the carried column, or the row id when the caller needs the permutation (more columns than two).
// Stability: passes 1 and 2 are stable and the counting step ranks equal words by position.
struct OwkTile {
  int64_t start;
  uint32_t len, k;   // rows, segment (digit of pass 1)
  uint32_t nt, tin;  // tiles of the segment, index of this tile among them
};
constexpr int OWK_SAMPLES = 16; // per group
constexpr uint32_t OWK_WALK = 192;

template <int KIND>
__global__ void owk_sample_kernel(const void *__restrict__ vals, int64_t n, int desc, uint64_t imin, int64_t S, int64_t stride,
                                  uint64_t *__restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= S) return;
  // (a jittered grid: an even stride would alias with periodic data)
  const int64_t row = min(n - 1, i * stride + (int64_t)(mix64((uint64_t)i) % (uint64_t)stride));
  out[i] = order_image<KIND>(vals, row, desc) - imin;
}
__global__ void owk_knots_kernel(const uint64_t *__restrict__ ss, uint32_t G, uint32_t per_group, uint64_t *__restrict__ sub) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < G) sub[g] = g ? ss[(size_t)g * per_group] : 0ull;
}
// A value that holds a large share of the rows (zeros, a default, a sentinel) shows up as a RUN of equal splitters.  Rows
// equal to such a value all go to the FIRST group of the run, which then holds nothing else ("pure": sub[g] == sub[g + 1]) and
// needs no sorting whatever its size — the stable passes have left its rows in input order; the values between the run and
// the next splitter go to the run's last group as before, the groups in between stay empty.
// topfirst[k] = first group whose splitter equals top-level splitter k (pass 1 sends the rows equal to it there).
__global__ void owk_topfirst_kernel(const uint64_t *__restrict__ sub, uint32_t G, uint32_t nk1, uint32_t *__restrict__ topfirst) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nk1) return;
  const uint64_t v = sub[(size_t)k << 8];
  uint32_t lo = 0, hi = k << 8; // first index with sub[index] >= v (sub[hi] == v)
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (sub[mid] < v) lo = mid + 1; else hi = mid;
  }
  topfirst[k] = lo;
}
// first t with sk[t] >= off, given that some sk[d] == off (a branch-free lower bound over the 256 entries of a level)
__device__ __forceinline__ uint32_t knot_first_equal(const uint64_t *__restrict__ sk, uint64_t off) {
  uint32_t pos = 0; // = number of entries < off
#pragma unroll
  for (uint32_t step = 128; step; step >>= 1)
    if (sk[pos + step - 1] < off) pos += step;
  return pos;
}

// last t < nk with sk[t] <= off (sk[0] <= off by construction), for ITEMS rows at once (independent chains of LDS reads)
template <int ITEMS>
__device__ __forceinline__ void knot_digits(const uint64_t *__restrict__ sk, uint32_t nk, const uint64_t (&off)[ITEMS], uint32_t (&dig)[ITEMS]) {
#pragma unroll
  for (int j = 0; j < ITEMS; j++) dig[j] = 0;
#pragma unroll
  for (uint32_t step = 128; step; step >>= 1) {
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
      const uint32_t t = dig[j] + step;
      if (t < nk && sk[t] <= off[j]) dig[j] = t;
    }
  }
}

// LEVEL 1: fixed blocks over the raw column, count matrix digit-major [d * nblocks + block];
// LEVEL 2: the tile list, count matrix segment-major [(first tile of the segment) * 256 + d * nt + tin]
template <int KIND, int LEVEL, bool REC_IN = false>
__global__ __launch_bounds__(OW_WG) void owk_hist_kernel(const void *__restrict__ src, int64_t n, int desc, uint64_t imin,
                                                         int64_t nblocks, uint32_t *__restrict__ hist,
                                                         const OwkTile *__restrict__ tiles, const uint64_t *__restrict__ sub, uint32_t nk1,
                                                         const uint32_t *__restrict__ topfirst) {
  __shared__ uint32_t h[256];
  __shared__ uint64_t sk[256];
  __shared__ uint32_t sfirst[256]; // LEVEL 1: segment of the rows EQUAL to top-level splitter k
  int64_t t0;
  uint32_t tl, nk = nk1;
  size_t hbase = 0, hstride = (size_t)nblocks, hcol = blockIdx.x;
  if (LEVEL == 2) {
    const OwkTile t = tiles[blockIdx.x];
    t0 = t.start;
    tl = t.len;
    nk = 256;
    hbase = (size_t)(blockIdx.x - t.tin) * 256;
    hstride = t.nt;
    hcol = t.tin;
    if (tl == 0) {
      if (threadIdx.x < 256) hist[hbase + threadIdx.x * hstride + hcol] = 0;
      return;
    }
    if (threadIdx.x < 256) sk[threadIdx.x] = sub[(size_t)t.k * 256 + threadIdx.x];
  } else {
    t0 = (int64_t)blockIdx.x * OW_TILE;
    tl = (uint32_t)min<int64_t>(OW_TILE, n - t0);
    if (threadIdx.x < 256) {
      sk[threadIdx.x] = threadIdx.x < nk1 ? sub[(size_t)threadIdx.x << 8] : ~0ull;
      sfirst[threadIdx.x] = threadIdx.x < nk1 ? topfirst[threadIdx.x] >> 8 : 0;
    }
  }
  if (threadIdx.x < 256) h[threadIdx.x] = 0;
  uint64_t k[OW_ITEMS];
#pragma unroll
  for (int r = 0; r < OW_ITEMS; r++) {
    const int64_t i = t0 + min((uint32_t)(threadIdx.x + r * OW_WG), tl - 1);
    k[r] = LEVEL == 1 ? order_image<KIND>(src, i, desc) - imin : __builtin_nontemporal_load((const uint64_t *)src + (REC_IN ? 2 * i : i));
  }
  __syncthreads();
  uint32_t dig[OW_ITEMS];
  knot_digits<OW_ITEMS>(sk, nk, k, dig);
#pragma unroll
  for (int r = 0; r < OW_ITEMS; r++) // a row equal to its splitter: the first group of the run of equal splitters
    if (sk[dig[r]] == k[r]) dig[r] = LEVEL == 1 ? sfirst[dig[r]] : knot_first_equal(sk, k[r]);
#pragma unroll
  for (int r = 0; r < OW_ITEMS; r++)
    if ((uint32_t)(threadIdx.x + r * OW_WG) < tl) atomicAdd(&h[dig[r]], 1u);
  __syncthreads();
  if (threadIdx.x < 256) hist[hbase + threadIdx.x * hstride + hcol] = h[threadIdx.x];
}

// The first pass's 256 segment sizes in ONE persistent launch (the look-back form of that pass: ow_scatter_kernel's LB):
// the search of owk_hist_kernel<KIND, 1>, counted into 256 global words instead of a (tile, digit) matrix
template <int KIND>
__global__ __launch_bounds__(OW_WG) void owk_ghist_kernel(const void *__restrict__ src, int64_t n, int desc, uint64_t imin,
                                                          int64_t nblocks, uint32_t *__restrict__ ghist,
                                                          const uint64_t *__restrict__ sub, uint32_t nk1,
                                                          const uint32_t *__restrict__ topfirst) {
  __shared__ uint32_t h[256];
  __shared__ uint64_t sk[256];
  __shared__ uint32_t sfirst[256];
  if (threadIdx.x < 256) {
    sk[threadIdx.x] = threadIdx.x < nk1 ? sub[(size_t)threadIdx.x << 8] : ~0ull;
    sfirst[threadIdx.x] = threadIdx.x < nk1 ? topfirst[threadIdx.x] >> 8 : 0;
    h[threadIdx.x] = 0;
  }
  __syncthreads();
  for (int64_t t = blockIdx.x; t < nblocks; t += gridDim.x) {
    const int64_t t0 = t * OW_TILE;
    const uint32_t tl = (uint32_t)min<int64_t>(OW_TILE, n - t0);
    uint64_t k[OW_ITEMS];
#pragma unroll
    for (int r = 0; r < OW_ITEMS; r++) k[r] = order_image<KIND>(src, t0 + min((uint32_t)(threadIdx.x + r * OW_WG), tl - 1), desc) - imin;
    uint32_t dig[OW_ITEMS];
    knot_digits<OW_ITEMS>(sk, nk1, k, dig);
#pragma unroll
    for (int r = 0; r < OW_ITEMS; r++)
      if (sk[dig[r]] == k[r]) dig[r] = sfirst[dig[r]];
#pragma unroll
    for (int r = 0; r < OW_ITEMS; r++)
      if ((uint32_t)(threadIdx.x + r * OW_WG) < tl) atomicAdd(&h[dig[r]], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 256 && h[threadIdx.x]) atomicAdd(&ghist[threadIdx.x], h[threadIdx.x]);
}

// pay == nullptr with NPAY: the payload is the row id (LEVEL 1 only).  REC: {word, payload} records out (LEVEL 2, NPAY)
// TWO: as in ow_scatter_kernel — word and payload take turns in one LDS tile, three workgroups per CU
// LB (LEVEL 1): the chained look-back of ow_scatter_kernel instead of the scanned count matrix — `ghist` = the 256 segment
// sizes (owk_ghist_kernel), `lbdesc` [tile][256] zeroed, `lb_fail` raised when a spin runs out
template <int KIND, int LEVEL, int NPAY, bool REC, bool REC_IN = false, bool TWO = false, bool LB = false>
__global__ __launch_bounds__(OW_WG, TWO ? OW_TWO_WGS : 1) void owk_scatter_kernel(const void *__restrict__ src, const uint64_t *__restrict__ pay, int64_t n,
                                                            int desc, uint64_t imin, int64_t nblocks,
                                                            const uint32_t *__restrict__ offsets, uint64_t *__restrict__ words_out,
                                                            uint64_t *__restrict__ pay_out, const OwkTile *__restrict__ tiles,
                                                            const uint64_t *__restrict__ sub, uint32_t nk1,
                                                            const uint32_t *__restrict__ topfirst,
                                                            const uint32_t *__restrict__ ghist = nullptr,
                                                            uint32_t *__restrict__ lbdesc = nullptr,
                                                            unsigned int *__restrict__ lb_fail = nullptr) {
  static_assert(!LB || LEVEL == 1, "look-back: the first pass");
  static_assert(!TWO || (REC && NPAY == 1), "TWO: the record form");
  __shared__ uint64_t sword[OW_TILE];
  __shared__ uint64_t spay[NPAY && !TWO ? OW_TILE : 1];
  __shared__ uint32_t wcnt[OW_WAVES][256]; // (its first 2 KB hold the splitters until the digits are known)
  __shared__ uint32_t dstart[256];
  __shared__ uint32_t gbase[256]; // position of the digit's run in the output minus its start in the tile (mod 2^32)
  __shared__ uint8_t sdig[OW_TILE];
  __shared__ uint32_t s_wsum[4];
  __shared__ uint32_t s_gsum[4];
  uint64_t *sk = (uint64_t *)&wcnt[0][0];
  uint32_t *sfirst = &wcnt[2][0]; // (behind the 2 KB of splitters; LEVEL 1 only)
  const int w = wave_id(), lane = lane_id();
  int64_t tbase;
  uint32_t len, nk = nk1;
  size_t hbase = 0, hstride = (size_t)nblocks, hcol = blockIdx.x;
  if (LEVEL == 2) {
    const OwkTile t = tiles[blockIdx.x];
    tbase = t.start;
    len = t.len;
    nk = 256;
    hbase = (size_t)(blockIdx.x - t.tin) * 256;
    hstride = t.nt;
    hcol = t.tin;
    if (len == 0) return;
    if (threadIdx.x < 256) sk[threadIdx.x] = sub[(size_t)t.k * 256 + threadIdx.x];
  } else {
    tbase = (int64_t)blockIdx.x * OW_TILE;
    len = (uint32_t)min<int64_t>(OW_TILE, n - tbase);
    if (threadIdx.x < 256) {
      sk[threadIdx.x] = threadIdx.x < nk1 ? sub[(size_t)threadIdx.x << 8] : ~0ull;
      sfirst[threadIdx.x] = threadIdx.x < nk1 ? topfirst[threadIdx.x] >> 8 : 0;
    }
  }
  const uint32_t wrow = (uint32_t)w * (OW_ITEMS * 64) + lane; // element of the tile
  uint64_t k[OW_ITEMS], v[NPAY ? OW_ITEMS : 1];
#pragma unroll
  for (int j = 0; j < OW_ITEMS; j++) {
    const int64_t i = tbase + min(wrow + j * 64, len - 1);
    if (REC_IN) { // (the previous pass wrote {word, payload} records)
      const u64x2 rec = __builtin_nontemporal_load((const u64x2 *)src + i);
      k[j] = rec.x;
      v[NPAY ? j : 0] = rec.y;
      continue;
    }
    k[j] = LEVEL == 1 ? order_image<KIND>(src, i, desc) - imin : __builtin_nontemporal_load((const uint64_t *)src + i);
    if (NPAY) v[j] = pay ? __builtin_nontemporal_load(pay + i) : (uint64_t)i;
  }
  const uint32_t goff = !LB && threadIdx.x < 256 ? offsets[hbase + threadIdx.x * hstride + hcol] : 0;
  const uint32_t gcnt = LB && threadIdx.x < 256 ? ghist[threadIdx.x] : 0;
  __syncthreads();
  uint32_t dig[OW_ITEMS], rnk[OW_ITEMS];
  knot_digits<OW_ITEMS>(sk, nk, k, dig);
#pragma unroll
  for (int j = 0; j < OW_ITEMS; j++) // a row equal to its splitter: the first group of the run of equal splitters
    if (sk[dig[j]] == k[j]) dig[j] = LEVEL == 1 ? sfirst[dig[j]] : knot_first_equal(sk, k[j]);
  __syncthreads(); // (the splitters are read: their bytes become the wave counters)
#pragma unroll
  for (int q = 0; q < 4; q++) wcnt[w][lane + 64 * q] = 0;
  bool valid[OW_ITEMS];
#pragma unroll
  for (int j = 0; j < OW_ITEMS; j++) valid[j] = wrow + j * 64 < len;
  stable_wave_ranks<OW_ITEMS>(dig, valid, wcnt[w], rnk);
  __syncthreads();
  uint32_t my_cnt = 0, my_ds = 0, gex = 0, lb_first = 0; // (LB, threads < 256: digit threadIdx.x of this tile)
  uint32_t *my_desc = LB ? lbdesc + (size_t)blockIdx.x * 256 + min(threadIdx.x, 255u) : nullptr;
  if (threadIdx.x < 256) {
    uint32_t acc = 0;
#pragma unroll
    for (int q = 0; q < OW_WAVES; q++) {
      uint32_t c = wcnt[q][threadIdx.x];
      wcnt[q][threadIdx.x] = acc;
      acc += c;
    }
    if (LB) { // publish the count, ask for the predecessor's word (consumed behind the staging loop)
      olb_store(my_desc, (blockIdx.x == 0 ? OLB_PFX : OLB_AGG) | acc);
      if (blockIdx.x > 0) lb_first = olb_load(my_desc - 256);
      my_cnt = acc;
      const uint32_t ginc = wave_iscan_u32(gcnt);
      if (lane == 63) s_gsum[w] = ginc;
      gex = ginc - gcnt;
    }
    uint32_t inc = wave_iscan_u32(acc);
    if (lane == 63) s_wsum[w] = inc;
    dstart[threadIdx.x] = inc - acc;
  }
  __syncthreads();
  if (threadIdx.x < 256) {
    uint32_t wb = 0;
    for (int q = 0; q < w; q++) wb += s_wsum[q];
    const uint32_t ds = dstart[threadIdx.x] + wb;
    dstart[threadIdx.x] = ds;
    if (LB) {
      my_ds = ds;
      for (int q = 0; q < w; q++) gex += s_gsum[q];
    } else
      gbase[threadIdx.x] = goff - ds;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < OW_ITEMS; j++) {
    if (!valid[j]) continue;
    const uint32_t p = dstart[dig[j]] + wcnt[w][dig[j]] + rnk[j];
    sword[p] = k[j];
    sdig[p] = (uint8_t)dig[j];
    if (TWO) rnk[j] = p; // (kept for the payload's turn)
    else if (NPAY) spay[p] = v[j];
  }
  if (LB && threadIdx.x < 256) { // (the walk of ow_scatter_kernel)
    uint32_t excl = 0;
    if (blockIdx.x > 0) {
      const uint32_t *p = my_desc - 256;
      uint32_t st = lb_first;
      for (;;) {
        unsigned spins = 0;
        while ((st >> 30) == 0) {
          if (++spins > LB_SPIN_LIMIT) {
            *lb_fail = 1u;
            st = OLB_PFX;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
          st = olb_load(p);
        }
        excl += st & OLB_VAL;
        if ((st >> 30) == 2 || p == lbdesc + threadIdx.x) break;
        p -= 256;
        st = olb_load(p);
      }
      olb_store(my_desc, OLB_PFX | ((excl + my_cnt) & OLB_VAL));
    }
    gbase[threadIdx.x] = gex + excl - my_ds;
  }
  __syncthreads();
  if constexpr (TWO) {
#pragma unroll
    for (int j = 0; j < OW_ITEMS; j++) k[j] = sword[min((uint32_t)(j * OW_WG) + threadIdx.x, (uint32_t)(OW_TILE - 1))]; // the words in output order
    __syncthreads();
#pragma unroll
    for (int j = 0; j < OW_ITEMS; j++)
      if (valid[j]) sword[rnk[j]] = v[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < OW_ITEMS; j++) {
      const uint32_t p = j * OW_WG + threadIdx.x;
      if (p < len) {
        u64x2 rec;
        rec.x = k[j];
        rec.y = sword[p];
        ((u64x2 *)words_out)[(size_t)(uint32_t)(gbase[sdig[p]] + p)] = rec;
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < OW_ITEMS; j++) {
    const uint32_t p = j * OW_WG + threadIdx.x;
    if (p < len) {
      const uint64_t kk = sword[p];
      const size_t g = (size_t)(uint32_t)(gbase[sdig[p]] + p);
      if (REC) {
        u64x2 rec;
        rec.x = kk;
        rec.y = spay[NPAY ? p : 0];
        ((u64x2 *)words_out)[g] = rec;
        continue;
      }
      words_out[g] = kk;
      if (NPAY) pay_out[g] = spay[p];
    }
  }
}

__global__ void owk_tile_fill_kernel(const uint32_t *__restrict__ firsttile, const int64_t *__restrict__ segstart, uint32_t ntmax,
                                     OwkTile *__restrict__ tiles) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntmax) return;
  OwkTile o;
  const uint32_t used = firsttile[256];
  if (t < used) {
    uint32_t lo = 0, hi = 256; // last segment whose first tile is <= t (segments without tiles share their successor's)
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (firsttile[mid] <= t) lo = mid; else hi = mid;
    }
    o.start = segstart[lo] + (int64_t)(t - firsttile[lo]) * OW_TILE;
    o.len = (uint32_t)min<int64_t>(OW_TILE, segstart[lo + 1] - o.start);
    o.k = lo;
    o.nt = firsttile[lo + 1] - firsttile[lo];
    o.tin = t - firsttile[lo];
  } else { // spare slots: one more "segment" of empty tiles behind the last
    o.start = 0;
    o.len = 0;
    o.k = 256;
    o.nt = ntmax - used;
    o.tin = t - used;
  }
  tiles[t] = o;
}
// group g = k << 8 | d: rows [gstart[g], gend[g]); one block per segment k; gend[G] = largest group that needs sorting,
// gend[G + 1] = entries of the work list `pure_items` — {group, chunk of 4096 rows} for every pure group (owk_topfirst_kernel)
constexpr uint32_t OWK_PURE_CHUNK = 4096;
__device__ __forceinline__ bool owk_pure(const uint64_t *__restrict__ sub, uint32_t G, uint32_t g) {
  return g + 1 < G && sub[g] == sub[g + 1];
}
__global__ __launch_bounds__(256) void owk_group_table_kernel(const uint32_t *__restrict__ offs2, const uint32_t *__restrict__ firsttile,
                                                              const int64_t *__restrict__ segstart, const uint64_t *__restrict__ sub,
                                                              uint32_t G, uint32_t *__restrict__ gstart, uint32_t *__restrict__ gend,
                                                              uint2 *__restrict__ pure_items) {
  const uint32_t k = blockIdx.x, d = threadIdx.x, g = k * 256 + d;
  const uint32_t ft = firsttile[k], nt = firsttile[k + 1] - ft;
  uint32_t a = 0xffffffffu, b = 0;
  if (nt) {
    a = offs2[(size_t)ft * 256 + (size_t)d * nt];
    b = d == 255 ? (uint32_t)segstart[k + 1] : offs2[(size_t)ft * 256 + (size_t)(d + 1) * nt];
  }
  gstart[g] = a;
  gend[g] = b;
  uint32_t sz = nt ? b - a : 0;
  if (sz && owk_pure(sub, G, g)) { // nothing to sort: its rows are copied out chunk by chunk (owk_pure_copy_kernel)
    const uint32_t items = (sz + OWK_PURE_CHUNK - 1) / OWK_PURE_CHUNK;
    const uint32_t at = atomicAdd(gend + G + 1, items);
    for (uint32_t q = 0; q < items; q++) pure_items[at + q] = make_uint2(g, q);
    sz = 0;
  }
  for (int s = 32; s >= 1; s >>= 1) sz = max(sz, (uint32_t)__shfl_xor((int)sz, s, 64));
  if (lane_id() == 0 && sz) atomicMax(gend + G, sz);
}
// rows of the pure groups: every row of the group carries the same word, and the stable passes kept them in input order
template <int KIND, int NPAY, bool REC>
__global__ __launch_bounds__(256) void owk_pure_copy_kernel(const uint64_t *__restrict__ words, const uint2 *__restrict__ items,
                                                            const uint32_t *__restrict__ gstart, const uint32_t *__restrict__ gend,
                                                            int desc, uint64_t imin, void *__restrict__ key_out,
                                                            uint64_t *__restrict__ pay_out, uint32_t *__restrict__ perm_out) {
  const uint2 it = items[blockIdx.x];
  const uint32_t lo = gstart[it.x] + it.y * OWK_PURE_CHUNK, hi = min(gend[it.x], lo + OWK_PURE_CHUNK);
  for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) {
    uint64_t kw, v = 0;
    if (REC) {
      const u64x2 rec = __builtin_nontemporal_load((const u64x2 *)words + i);
      kw = rec.x;
      v = rec.y;
    } else
      kw = __builtin_nontemporal_load(words + i);
    order_store_key<KIND>(key_out, i, kw + imin, desc);
    if (perm_out) perm_out[i] = (uint32_t)v;
    else if (NPAY) pay_out[i] = v;
  }
}

// perm_out != nullptr: the payload is the row id — it leaves as the permutation and there is no carried column
template <int KIND, int NPAY, int R, bool REC>
__global__ __launch_bounds__(FIN_WG, R == 8 ? 4 : 1) void owk_finish_kernel(const uint64_t *__restrict__ words, const uint64_t *__restrict__ pay,
                                                            const uint32_t *__restrict__ gstart, const uint32_t *__restrict__ gend,
                                                            const uint64_t *__restrict__ sub, uint32_t G, int desc, uint64_t imin,
                                                            void *__restrict__ key_out, uint64_t *__restrict__ pay_out,
                                                            uint32_t *__restrict__ perm_out, uint32_t m_above, uint32_t m_upto,
                                                            int count_form) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lo = gstart[blockIdx.x], hi = gend[blockIdx.x];
  if (lo == 0xffffffffu || lo >= hi) return;
  const uint32_t m = hi - lo;
  if (m <= m_above || m > m_upto) return; // (a group of another size class: the launch with the LDS room for it takes it)
  if (owk_pure(sub, G, blockIdx.x)) return; // (one value, nothing to sort: owk_pure_copy_kernel)
  uint64_t *sword = (uint64_t *)smem;                       // [R * FIN_WG]
  uint64_t *spay = sword + (size_t)R * FIN_WG;              // [NPAY ? R * FIN_WG : 0]
  uint32_t *wcnt = (uint32_t *)(spay + (NPAY ? (size_t)R * FIN_WG : 0)); // [FIN_WAVES][256]
  uint32_t *dstart = wcnt + FIN_WAVES * 256;                // [256]
  __shared__ uint32_t s_wsum[FIN_WAVES];
  __shared__ uint32_t s_heavy;
  __shared__ uint64_t s_mn[FIN_WAVES], s_mx[FIN_WAVES];
  const int w = wave_id(), lane = lane_id();
  // wave w owns chunks [w * cpw, (w + 1) * cpw) of 64 consecutive rows (cpw <= R): all four waves work whatever the group's size
  const uint32_t cpw = ((m + 63) / 64 + FIN_WAVES - 1) / FIN_WAVES;
  // rel = word - base < 2^tb.  Inner groups: base = the group's splitter, width = the distance to the next one.  The first
  // and the last group reach down to 0 / up to the end of the 64-bit range, far beyond the values their rows have: they take
  // their own extremes (a workgroup reduction over the rows just loaded)
  const bool own_extremes = blockIdx.x == 0 || blockIdx.x + 1 == G;
  uint64_t base = own_extremes ? 0 : sub[blockIdx.x];
  uint64_t relmax = own_extremes ? 0 : sub[blockIdx.x + 1] - 1 - base; // (next > base: the group has rows)
  int sh = 0, top = 16; // in-LDS passes on bits [sh, top) of rel, counting below sh
  if (threadIdx.x == 0) s_heavy = 0;
  // (before the attempts below and with its own loads, so that the two forms do not hold each other's registers: inside the
  //  attempt loop the bucket + count form spilled 312 bytes per lane at the 128 VGPRs four workgroups per CU allow and ran the
  //  finish at 2.9 ms instead of 1.9)
  if (count_form && !own_extremes) {
    const int tbits = relmax ? 64 - __builtin_clzll(relmax) : 0; // rel < 2^tbits
    uint64_t k[R], v[NPAY ? R : 1];
    bool valid[R];
    if (tbits > 8) {
#pragma unroll
      for (int j = 0; j < R; j++) {
        const uint32_t e = (uint32_t)(w * cpw + j) * 64 + lane;
        valid[j] = (uint32_t)j < cpw && e < m;
        const uint32_t i = lo + min(e, m - 1);
        if (REC) {
          const u64x2 rec = __builtin_nontemporal_load((const u64x2 *)words + i);
          k[j] = rec.x - base;
          v[NPAY ? j : 0] = rec.y;
        } else {
          k[j] = __builtin_nontemporal_load(words + i) - base;
          if (NPAY) v[j] = __builtin_nontemporal_load(pay + i);
        }
      }
    }
    // Round 6 — BUCKET + COUNT (see ow_finish_kernel): the rows go into up to 1024 buckets by the top bits of rel, take any free slot
    // of their bucket, and every row counts the entries of its bucket below its own (rel, then position in the group — with no
    // payload equal words are interchangeable and the slot breaks the tie).  Replaces the two LSD passes AND the neighbour walk
    // when no bucket holds more than FIN_BUCKET_CAP rows; heavy values and dense clusters take the passes below as before.
    if (tbits > 8) {
      const int nbb = tbits < 10 ? tbits : 10, lowb = tbits - nbb;
      const uint32_t NB = 1u << nbb, per = NB / FIN_WG;
      uint32_t *A = wcnt + 1;               // A[b] (A[-1] = 0): count -> start -> end of bucket b
      uint32_t *spos = (uint32_t *)spay;    // positions of the entries (NPAY only)
      __shared__ uint32_t s_maxb;
      for (uint32_t q = threadIdx.x; q <= NB; q += FIN_WG) wcnt[q] = 0;
      if (threadIdx.x == 0) s_maxb = 0;
      __syncthreads();
      uint32_t bk[R];
#pragma unroll
      for (int j = 0; j < R; j++) {
        bk[j] = (uint32_t)min(k[j] >> lowb, (uint64_t)(NB - 1)); // (rel <= relmax < 2^tbits; clamped all the same: a counter index)
        if (valid[j]) atomicAdd(&A[bk[j]], 1u);
      }
      __syncthreads();
      {
        uint32_t c[4], sum = 0, mx = 0;
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) {
          c[i] = i < per ? A[threadIdx.x * per + i] : 0u;
          sum += c[i];
          mx = max(mx, c[i]);
        }
        const uint32_t inc = wave_iscan_u32(sum);
        if (lane == 63) s_wsum[w] = inc;
        for (int q = 32; q >= 1; q >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, q, 64));
        if (lane == 0 && mx > FIN_BUCKET_CAP) s_maxb = mx;
        __syncthreads();
        uint32_t run = inc - sum;
        for (int q = 0; q < w; q++) run += s_wsum[q];
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) {
          if (i < per) A[threadIdx.x * per + i] = run;
          run += c[i];
        }
      }
      __syncthreads();
      if (s_maxb == 0) {
        uint32_t slot[NPAY ? 1 : R]; // (no payload: the slot is the tie-break)
#pragma unroll
        for (int j = 0; j < R; j++) {
          if (!valid[j]) continue;
          const uint32_t sl = atomicAdd(&A[bk[j]], 1u);
          sword[sl] = k[j];
          if (NPAY) spos[sl] = (uint32_t)(w * cpw + j) * 64 + lane;
          else slot[j] = sl;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < R; j++) {
          if (!valid[j]) continue;
          const uint32_t s0 = A[(int)bk[j] - 1], s1 = A[bk[j]], mypos = NPAY ? (uint32_t)(w * cpw + j) * 64 + lane : slot[NPAY ? 0 : j];
          uint32_t r = s0;
          for (uint32_t q = s0; q < s1; q++) {
            const uint64_t o = sword[q];
            const uint32_t op = NPAY ? spos[q] : q;
            r += (o < k[j]) || (o == k[j] && op < mypos);
          }
          bk[j] = r;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < R; j++) {
          if (!valid[j]) continue;
          sword[bk[j]] = k[j];
          if (NPAY) spay[bk[j]] = v[j];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < R; j++) {
          if (!valid[j]) continue;
          const uint32_t e = (uint32_t)(w * cpw + j) * 64 + lane;
          const uint64_t pv = spay[NPAY ? e : 0];
          order_store_key<KIND>(key_out, lo + e, sword[e] + base + imin, desc);
          if (perm_out) perm_out[lo + e] = (uint32_t)pv;
          else if (NPAY) pay_out[lo + e] = pv;
        }
        return;
      }
      __syncthreads(); // (the passes below start from the registers; wcnt is theirs again)
    }
  }
  for (int attempt = 0; attempt < 2; attempt++) {
    uint64_t k[R], v[NPAY ? R : 1];
    bool valid[R];
#pragma unroll
    for (int j = 0; j < R; j++) {
      const uint32_t e = (uint32_t)(w * cpw + j) * 64 + lane;
      valid[j] = (uint32_t)j < cpw && e < m;
      const uint32_t i = lo + min(e, m - 1);
      if (REC) {
        const u64x2 rec = __builtin_nontemporal_load((const u64x2 *)words + i);
        k[j] = rec.x;
        v[NPAY ? j : 0] = rec.y;
      } else {
        k[j] = __builtin_nontemporal_load(words + i);
        if (NPAY) v[j] = __builtin_nontemporal_load(pay + i);
      }
    }
    if (attempt == 0) {
      if (own_extremes) { // (rows past the group's end repeat its last row: they do not move the extremes)
        uint64_t mn = ~0ull, mx = 0;
#pragma unroll
        for (int j = 0; j < R; j++) {
          mn = min(mn, k[j]);
          mx = max(mx, k[j]);
        }
        mn = wave_min_u64(mn);
        mx = wave_max_u64(mx);
        if (lane == 0) {
          s_mn[w] = mn;
          s_mx[w] = mx;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < FIN_WAVES; q++) {
          mn = min(mn, s_mn[q]);
          mx = max(mx, s_mx[q]);
        }
        base = mn;
        relmax = mx - mn;
      }
      const int tb = relmax ? 64 - __builtin_clzll(relmax) : 0;
      sh = tb > 16 ? tb - 16 : 0;
      top = sh + 16;
    }
#pragma unroll
    for (int j = 0; j < R; j++) k[j] -= base;
    for (int shift = attempt ? 0 : sh; shift < top; shift += 8) { // stable LSD passes, all in LDS
      for (int q = lane; q < 256; q += 64) wcnt[w * 256 + q] = 0;
      uint32_t dig[R], rnk[R];
#pragma unroll
      for (int j = 0; j < R; j++) dig[j] = (uint32_t)(k[j] >> shift) & ((shift + 8 > top) ? ((1u << (top - shift)) - 1) : 255u);
      stable_wave_ranks<R, true>(dig, valid, wcnt + w * 256, rnk);
      __syncthreads();
      {
        uint32_t acc = 0;
#pragma unroll
        for (int q = 0; q < FIN_WAVES; q++) {
          const uint32_t c = wcnt[q * 256 + threadIdx.x];
          wcnt[q * 256 + threadIdx.x] = acc;
          acc += c;
        }
        const uint32_t inc = wave_iscan_u32(acc);
        if (lane == 63) s_wsum[w] = inc;
        dstart[threadIdx.x] = inc - acc;
      }
      __syncthreads();
      {
        uint32_t wb = 0;
        for (int q = 0; q < w; q++) wb += s_wsum[q];
        dstart[threadIdx.x] += wb;
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < R; j++) {
        if (!valid[j]) continue;
        const uint32_t p = dstart[dig[j]] + wcnt[w * 256 + dig[j]] + rnk[j];
        sword[p] = k[j];
        if (NPAY) spay[p] = v[j];
      }
      __syncthreads();
      if (!attempt && sh != 0 && shift + 8 >= top) break; // (the counting step below reads the LDS copy, not the registers)
#pragma unroll
      for (int j = 0; j < R; j++) {
        const uint32_t e = min((uint32_t)(w * cpw + j) * 64 + lane, m - 1);
        k[j] = sword[e];
        if (NPAY) v[j] = spay[e];
      }
      __syncthreads();
    }
    if (attempt || sh == 0) { // every bit of rel has been sorted on: the rows are in place
#pragma unroll
      for (int j = 0; j < R; j++) {
        if (!valid[j]) continue;
        const uint32_t i = lo + (uint32_t)(w * cpw + j) * 64 + lane;
        order_store_key<KIND>(key_out, i, k[j] + base + imin, desc);
        if (perm_out) perm_out[i] = (uint32_t)v[NPAY ? j : 0];
        else if (NPAY) pay_out[i] = v[j];
      }
      return;
    }
    // Rows with equal bits [sh, ..) form runs — of one row, mostly: the group's ~1.5 K rows fall onto 65 536 values.  A row
    // whose two neighbours have other bits is in place; a row in a run is ranked inside it by counting.  All from the LDS
    // copy the last pass left (sword / spay), four independent reads per row.
    for (uint32_t e = threadIdx.x; e < m; e += FIN_WG) {
      const uint64_t me = sword[e], pw = sword[e ? e - 1 : 0], nw = sword[min(e + 1, m - 1)];
      const uint64_t pv = spay[NPAY ? e : 0];
      const uint64_t pf = me >> sh;
      uint32_t pos = e;
      if ((e && (pw >> sh) == pf) || (e + 1 < m && (nw >> sh) == pf)) {
        uint32_t before = 0, steps = 0;
        int64_t q = (int64_t)e - 1;
        for (; q >= 0 && steps <= OWK_WALK; q--, steps++) {
          const uint64_t o = sword[q];
          if ((o >> sh) != pf) break;
          before += o <= me; // (an equal word further up stays in front: ties in input order)
        }
        bool heavy = steps > OWK_WALK;
        steps = 0;
        for (uint32_t f = e + 1; f < m && steps <= OWK_WALK; f++, steps++) {
          const uint64_t o = sword[f];
          if ((o >> sh) != pf) break;
          before += o < me;
        }
        heavy |= steps > OWK_WALK;
        if (heavy) {
          s_heavy = 1; // (the whole group is redone below: what has been written meanwhile is overwritten)
          continue;
        }
        pos = (uint32_t)(q + 1) + before;
      }
      const uint32_t i = lo + pos;
      order_store_key<KIND>(key_out, i, me + base + imin, desc);
      if (perm_out) perm_out[i] = (uint32_t)pv;
      else if (NPAY) pay_out[i] = pv;
    }
    __syncthreads();
    if (!s_heavy) return;
  }
}

// the wide route; false = not taken (nothing produced that the caller may use).  `imin`, `range`: EXACT extremes of the
// key image.  want_perm: the row ids travel as the payload and `carry_out` stays empty (the caller gathers that column
// like the others)
static std::atomic<bool> g_order_lb_off{false}; // a look-back spin ran out once: the counting forms for the rest of the process
template <int KIND>
static bool order_wide(Ctx *ctx, const DCol &key, int desc, const DCol *carry, int64_t n, uint64_t imin, uint64_t range,
                       DCol *key_out, DCol *carry_out, BufP *perm_out, bool want_perm) {
  if (const char *e = hook("SQLRS_ORDER_WIDE")) // (A/B hook, read per call: 0 = the general path)
    if (e[0] == '0') return false;
  int top = 9;
  while (top < 16 && (n >> top) > 2048) top++;
  const uint32_t G = 1u << top, nk1 = G >> 8;
  int per_group = OWK_SAMPLES;
  if (const char *e = hook("SQLRS_ORDER_SAMPLES")) per_group = std::max(1, std::min(256, std::atoi(e))); // (A/B hook, read per call)
  const int64_t S = (int64_t)G * per_group;
  if (n < S) return false;
  const int kb = range ? 64 - __builtin_clzll(range) : 1;
  const bool pay_rows = want_perm, has_pay = pay_rows || carry != nullptr;
  // 0. splitters
  BufP ss = ctx->alloc(8 * (size_t)S), ssv = ctx->alloc(4 * (size_t)S), sub = ctx->alloc(8 * ((size_t)G + 1));
  BufP topfirst = ctx->alloc(4 * 256);
  {
    ProfScope ps(ctx, "order_knots");
    owk_sample_kernel<KIND><<<dim3((unsigned)ceil_div(S, 256)), dim3(256), 0, ctx->stream>>>(key.values, n, desc, imin, S, n / S, ss->as<uint64_t>());
    SQ_HIP(hipGetLastError());
    radix_sort_pairs(ctx, ss->as<uint64_t>(), ssv->as<uint32_t>(), S, 0, kb, true);
    owk_knots_kernel<<<dim3((unsigned)ceil_div((int64_t)G, 256)), dim3(256), 0, ctx->stream>>>(ss->as<uint64_t>(), G, (uint32_t)per_group, sub->as<uint64_t>());
    owk_topfirst_kernel<<<dim3(1), dim3(256), 0, ctx->stream>>>(sub->as<uint64_t>(), G, nk1, topfirst->as<uint32_t>());
    SQ_HIP(hipGetLastError());
  }
  const uint64_t *subp = sub->as<uint64_t>();
  const uint32_t *tfp = topfirst->as<uint32_t>();
  // 1. pass 1: top-level splitters, raw column -> (word, payload) columns
  const int64_t nblocks = ceil_div(n, OW_TILE), ntmax = nblocks + 256;
  // (with a payload both passes write {word, payload} records: a (tile, digit) run is one piece instead of one per column.
  //  SQLRS_ORDER_WIDE_REC1=0, read per call: pass 1 writes two columns)
  const char *rec1_e = hook("SQLRS_ORDER_WIDE_REC1");
  const bool rec1 = has_pay && !(rec1_e && rec1_e[0] == '0');
  BufP w1 = ctx->alloc((rec1 ? 16 : 8) * (size_t)n), p1 = has_pay && !rec1 ? ctx->alloc(8 * (size_t)n) : nullptr;
  BufP hist = ctx->alloc(4 * (size_t)(256 * nblocks)), offs = ctx->alloc(4 * (size_t)(256 * nblocks)), total = ctx->alloc(8);
  const uint64_t *psrc = (carry && !pay_rows) ? carry->v<uint64_t>() : nullptr;
  dim3 g1((unsigned)nblocks), g2((unsigned)ntmax), b(OW_WG);
  const char *two_e = hook("SQLRS_ORDER_TWO"); // (read per call: 0 = word and payload side by side in LDS, two workgroups per CU)
  const bool two = !(two_e && two_e[0] == '0');
  // The first pass in its look-back form (the narrow route's, see ow_scatter_kernel): its 256 segment sizes from one persistent
  // launch, the tiles chained — no count matrix, no scan.  The second pass keeps its counting form: its digit is a search in the
  // row's own segment's splitters, so nothing ahead of the first pass can count it.  SQLRS_ORDER_LB=0 (read per call) / a spin
  // that ran out: the counting form.
  static thread_local bool wide_lb_skip = false;
  const char *lb_e = hook("SQLRS_ORDER_LB"), *lbf_e = hook("SQLRS_ORDER_LB_TEST_FAIL");
  const bool lb1 = rec1 && two && n < (1ll << 30) && !g_order_lb_off.load() && !wide_lb_skip && !(lb_e && lb_e[0] == '0');
  BufP ghb1, lbdesc1;
  if (lb1) {
    ProfScope ps(ctx, "order_split");
    ghb1 = ctx->alloc(4 * 260);
    lbdesc1 = ctx->alloc(4 * 256 * (size_t)nblocks);
    SQ_HIP(hipMemsetAsync(ghb1->p, 0, 4 * 260, ctx->stream));
    SQ_HIP(hipMemsetAsync(lbdesc1->p, 0, 4 * 256 * (size_t)nblocks, ctx->stream));
    const unsigned gblocks = (unsigned)std::min<int64_t>(nblocks, 4 * (int64_t)ctx->num_cus);
    owk_ghist_kernel<KIND><<<dim3(gblocks), b, 0, ctx->stream>>>(key.values, n, desc, imin, nblocks, ghb1->as<uint32_t>(), subp, nk1, tfp);
    owk_scatter_kernel<KIND, 1, 1, true, false, true, true><<<g1, b, 0, ctx->stream>>>(key.values, psrc, n, desc, imin, nblocks, nullptr,
                                                                                        w1->as<uint64_t>(), nullptr, nullptr, subp, nk1, tfp,
                                                                                        ghb1->as<uint32_t>(), lbdesc1->as<uint32_t>(),
                                                                                        ghb1->as<uint32_t>() + 256);
    SQ_HIP(hipGetLastError());
  } else {
    ProfScope ps(ctx, "order_split");
    owk_hist_kernel<KIND, 1><<<g1, b, 0, ctx->stream>>>(key.values, n, desc, imin, nblocks, hist->as<uint32_t>(), nullptr, subp, nk1, tfp);
    SQ_HIP(hipGetLastError());
    exclusive_scan_u32(ctx, hist->as<uint32_t>(), 256 * nblocks, nullptr, offs->as<uint32_t>(), total->as<uint64_t>());
    if (rec1 && two)
      owk_scatter_kernel<KIND, 1, 1, true, false, true><<<g1, b, 0, ctx->stream>>>(key.values, psrc, n, desc, imin, nblocks, offs->as<uint32_t>(),
                                                                                  w1->as<uint64_t>(), nullptr, nullptr, subp, nk1, tfp);
    else if (rec1)
      owk_scatter_kernel<KIND, 1, 1, true><<<g1, b, 0, ctx->stream>>>(key.values, psrc, n, desc, imin, nblocks, offs->as<uint32_t>(),
                                                                     w1->as<uint64_t>(), nullptr, nullptr, subp, nk1, tfp);
    else if (has_pay)
      owk_scatter_kernel<KIND, 1, 1, false><<<g1, b, 0, ctx->stream>>>(key.values, psrc, n, desc, imin, nblocks, offs->as<uint32_t>(),
                                                                      w1->as<uint64_t>(), p1->as<uint64_t>(), nullptr, subp, nk1, tfp);
    else
      owk_scatter_kernel<KIND, 1, 0, false><<<g1, b, 0, ctx->stream>>>(key.values, nullptr, n, desc, imin, nblocks, offs->as<uint32_t>(),
                                                                      w1->as<uint64_t>(), nullptr, nullptr, subp, nk1, tfp);
    SQ_HIP(hipGetLastError());
  }
  // 2. pass 2 over segment-aligned tiles: the segment's own 256 splitters; records out when there is a payload
  BufP firsttile = ctx->alloc(4 * 257), segstart = ctx->alloc(8 * 257), tiles2 = ctx->alloc(sizeof(OwkTile) * (size_t)ntmax);
  BufP hist2 = ctx->alloc(4 * (size_t)(256 * ntmax)), offs2 = ctx->alloc(4 * (size_t)(256 * ntmax));
  BufP out2 = ctx->alloc((has_pay ? 16 : 8) * (size_t)n);
  {
    ProfScope ps(ctx, "order_split");
    if (lb1) ow_tile_plan_gh_kernel<<<dim3(1), dim3(256), 0, ctx->stream>>>(ghb1->as<uint32_t>(), n, firsttile->as<uint32_t>(), segstart->as<int64_t>());
    else ow_tile_plan_kernel<<<dim3(1), dim3(256), 0, ctx->stream>>>(offs->as<uint32_t>(), nblocks, n, firsttile->as<uint32_t>(), segstart->as<int64_t>());
    owk_tile_fill_kernel<<<dim3((unsigned)ceil_div(ntmax, 256)), dim3(256), 0, ctx->stream>>>(firsttile->as<uint32_t>(), segstart->as<int64_t>(),
                                                                                             (uint32_t)ntmax, (OwkTile *)tiles2->p);
    const OwkTile *tp = (const OwkTile *)tiles2->p;
    if (rec1) owk_hist_kernel<KIND, 2, true><<<g2, b, 0, ctx->stream>>>(w1->p, n, desc, imin, ntmax, hist2->as<uint32_t>(), tp, subp, nk1, tfp);
    else owk_hist_kernel<KIND, 2><<<g2, b, 0, ctx->stream>>>(w1->p, n, desc, imin, ntmax, hist2->as<uint32_t>(), tp, subp, nk1, tfp);
    SQ_HIP(hipGetLastError());
    exclusive_scan_u32(ctx, hist2->as<uint32_t>(), 256 * ntmax, nullptr, offs2->as<uint32_t>(), total->as<uint64_t>());
    if (rec1 && two)
      owk_scatter_kernel<KIND, 2, 1, true, true, true><<<g2, b, 0, ctx->stream>>>(w1->p, nullptr, n, desc, imin, ntmax, offs2->as<uint32_t>(),
                                                                                 out2->as<uint64_t>(), nullptr, tp, subp, nk1, tfp);
    else if (rec1)
      owk_scatter_kernel<KIND, 2, 1, true, true><<<g2, b, 0, ctx->stream>>>(w1->p, nullptr, n, desc, imin, ntmax, offs2->as<uint32_t>(),
                                                                           out2->as<uint64_t>(), nullptr, tp, subp, nk1, tfp);
    else if (has_pay && two)
      owk_scatter_kernel<KIND, 2, 1, true, false, true><<<g2, b, 0, ctx->stream>>>(w1->p, p1->as<uint64_t>(), n, desc, imin, ntmax, offs2->as<uint32_t>(),
                                                                                  out2->as<uint64_t>(), nullptr, tp, subp, nk1, tfp);
    else if (has_pay)
      owk_scatter_kernel<KIND, 2, 1, true><<<g2, b, 0, ctx->stream>>>(w1->p, p1->as<uint64_t>(), n, desc, imin, ntmax, offs2->as<uint32_t>(),
                                                                     out2->as<uint64_t>(), nullptr, tp, subp, nk1, tfp);
    else
      owk_scatter_kernel<KIND, 2, 0, false><<<g2, b, 0, ctx->stream>>>(w1->p, nullptr, n, desc, imin, ntmax, offs2->as<uint32_t>(),
                                                                      out2->as<uint64_t>(), nullptr, tp, subp, nk1, tfp);
    SQ_HIP(hipGetLastError());
  }
  // 3. groups
  BufP gstart = ctx->alloc(4 * (size_t)65536), gend = ctx->alloc(4 * ((size_t)65536 + 4)); // [G]: largest group, [G + 1]: pure chunks, [G + 2]: look-back spin ran out
  BufP pure_items = ctx->alloc(8 * ((size_t)ceil_div(n, (int64_t)OWK_PURE_CHUNK) + G + 1));
  SQ_HIP(hipMemsetAsync(gend->as<uint32_t>() + G, 0, 12, ctx->stream));
  if (lb1) SQ_HIP(hipMemcpyAsync(gend->as<uint32_t>() + G + 2, ghb1->as<uint32_t>() + 256, 4, hipMemcpyDeviceToDevice, ctx->stream));
  {
    ProfScope ps(ctx, "order_groups");
    owk_group_table_kernel<<<dim3(nk1), dim3(256), 0, ctx->stream>>>(offs2->as<uint32_t>(), firsttile->as<uint32_t>(), segstart->as<int64_t>(), subp, G,
                                                                   gstart->as<uint32_t>(), gend->as<uint32_t>(), (uint2 *)pure_items->p);
    SQ_HIP(hipGetLastError());
  }
  const uint32_t *gh = (const uint32_t *)ctx->fetch(gend->as<uint32_t>() + G, 12);
  const uint32_t max_group = gh[0], pure_chunks = gh[1];
  if (lb1 && (gh[2] || (lbf_e && lbf_e[0] == '1'))) { // nothing of this attempt is valid: once more in the counting form
    if (gh[2]) {
      g_order_lb_off.store(true);
      ctx->order_lb_fallbacks++;
    }
    struct Skip {
      Skip() { wide_lb_skip = true; }
      ~Skip() { wide_lb_skip = false; }
    } skip;
    return order_wide<KIND>(ctx, key, desc, carry, n, imin, range, key_out, carry_out, perm_out, want_perm);
  }
  if (hook("SQLRS_ORDER_TRACE"))
    std::fprintf(stderr, "[order_wide] n=%lld key bits=%d groups=%u largest group to sort=%u rows, %u chunks of single-value groups\n",
                 (long long)n, kb, G, max_group, pure_chunks);
  if (max_group > FIN_CAP) return false; // thousands of distinct keys between two neighbouring samples: general path
  // 4. finish
  key_out->dtype = key.dtype;
  key_out->length = n;
  key_out->null_count = 0;
  key_out->own_values = ctx->alloc((KIND == OKIND_I32 ? 4 : 8) * (size_t)n + 16);
  key_out->values = key_out->own_values->p;
  uint32_t *perm = nullptr;
  uint64_t *po = nullptr;
  if (pay_rows) {
    *perm_out = ctx->alloc(4 * (size_t)n);
    perm = (*perm_out)->as<uint32_t>();
  } else if (carry) {
    carry_out->dtype = carry->dtype;
    carry_out->length = n;
    carry_out->null_count = 0;
    carry_out->own_values = ctx->alloc(8 * (size_t)n + 16);
    carry_out->values = carry_out->own_values->p;
    po = carry_out->own_values->as<uint64_t>();
  }
  const char *fc_e = hook("SQLRS_ORDER_FINISH_COUNT"); // (A/B hook, read per call: 0 = LSD passes + neighbour walk only)
  const int fin_count = !(fc_e && fc_e[0] == '0');
  {
    ProfScope ps(ctx, "order_finish");
    // The splitters balance the groups only statistically (16 samples per group: sizes spread like a Gamma(16), the largest of
    // 65 536 is ~2.5x the mean), and the LDS a workgroup asks for decides how many are resident: the groups of up to
    // 2048 rows (9 in 10) go through a launch of their own with 37 KB each, those of up to 4096 through a second, the few
    // beyond (if any) through a third.
#define SQ_WFIN(NP, RR, ABOVE, UPTO)                                                                                 \
  do {                                                                                                               \
    auto kfn = owk_finish_kernel<KIND, NP, RR, NP == 1>;                                                             \
    const size_t lds = (size_t)RR * FIN_WG * 8 * (1 + NP) + 4 * (FIN_WAVES * 256 + 256);                             \
    if (lds > 64 * 1024) allow_big_lds(ctx, kfn);                                                                    \
    kfn<<<dim3(G), dim3(FIN_WG), lds, ctx->stream>>>(out2->as<uint64_t>(), nullptr, gstart->as<uint32_t>(), gend->as<uint32_t>(), subp, G, \
                                                     desc, imin, key_out->own_values->p, po, perm,         \
                                                     (uint32_t)(ABOVE), (uint32_t)(UPTO), fin_count);                \
  } while (0)
#define SQ_WFIN_R(NP)                                                                                                \
  do {                                                                                                               \
    SQ_WFIN(NP, 8, 0, 8 * FIN_WG);                                                                                   \
    if (max_group > 8 * FIN_WG) SQ_WFIN(NP, 16, 8 * FIN_WG, 16 * FIN_WG);                                            \
    if (max_group > 16 * FIN_WG) SQ_WFIN(NP, 24, 16 * FIN_WG, FIN_CAP);                                              \
  } while (0)
    if (has_pay) SQ_WFIN_R(1);
    else SQ_WFIN_R(0);
#undef SQ_WFIN_R
#undef SQ_WFIN
    if (pure_chunks) { // groups of ONE value (heavy hitters): copied out as they are
      const uint2 *items = (const uint2 *)pure_items->p;
      if (has_pay)
        owk_pure_copy_kernel<KIND, 1, true><<<dim3(pure_chunks), dim3(256), 0, ctx->stream>>>(
            out2->as<uint64_t>(), items, gstart->as<uint32_t>(), gend->as<uint32_t>(), desc, imin, key_out->own_values->p, po, perm);
      else
        owk_pure_copy_kernel<KIND, 0, false><<<dim3(pure_chunks), dim3(256), 0, ctx->stream>>>(
            out2->as<uint64_t>(), items, gstart->as<uint32_t>(), gend->as<uint32_t>(), desc, imin, key_out->own_values->p, po, perm);
    }
    SQ_HIP(hipGetLastError());
  }
  return true;
}

// `optimistic`: the key range comes from a SAMPLE (every 16th chunk of 2048 rows: 0.19 -> 0.03 ms for 1e8 rows), widened
// as far as the same number of key bits allows; the first split pass tests every key against it and *retry_exact is set
// (nothing produced, return false) when one lies outside — the caller runs the exact form once.
template <int KIND, int NPAY>
static bool order_fast_impl(Ctx *ctx, const DCol &key, int desc, const DCol *carry, int64_t n, DCol *key_out, DCol *carry_out,
                            BufP *perm_out, bool want_perm, bool optimistic, bool *retry_exact, bool *in_order,
                            bool hbm_only = false, bool *retry_hbm_only = nullptr) {
  // 0. key range
  BufP mm = ctx->alloc(16 * OW_MM_SLOTS + 16); // {min = ~0, max = 0} x OW_MM_SLOTS | out-of-range flag (u32), largest group (u32) | inversion seen (u32)
  constexpr int FLAG_W = 2 * OW_MM_SLOTS;       // index of the flag word (u64)
  unsigned int *inv = (unsigned int *)(mm->as<uint64_t>() + FLAG_W + 1); // (its upper half: the heavy-value probe's count)
  const char *hp_e = hook("SQLRS_ORDER_HEAVY_PROBE"); // (A/B hook, read per call: 0 = no probe)
  const bool heavy_probe = !hbm_only && !(hp_e && hp_e[0] == '0');
  {
    ProfScope ps(ctx, "order_minmax");
    order_minmax_init_kernel<<<dim3(1), dim3(128), 0, ctx->stream>>>(mm->as<unsigned long long>());
    if (in_order) order_inversion_kernel<KIND, true><<<dim3(256), dim3(256), 0, ctx->stream>>>(key.values, n, desc, inv);
    if (heavy_probe) order_heavy_probe_kernel<KIND><<<dim3(1), dim3(1024), 0, ctx->stream>>>(key.values, n, desc, inv + 1);
    const int every = optimistic ? 16 : 1;
    unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n, (int64_t)256 * 8 * every), 8 * (int64_t)ctx->num_cus));
    order_minmax_kernel<KIND><<<dim3(blocks), dim3(256), 0, ctx->stream>>>(key.values, n, desc, mm->as<unsigned long long>(), every);
    SQ_HIP(hipGetLastError());
  }
  const uint64_t *h = (const uint64_t *)ctx->fetch(mm->p, 16 * OW_MM_SLOTS + 16);
  const uint32_t heavy_cnt = (uint32_t)(h[FLAG_W + 1] >> 32);
  if (in_order && (uint32_t)h[FLAG_W + 1] == 0) { // no inversion among the sampled pairs: look at every pair
    ProfScope ps(ctx, "order_minmax");
    const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n, 256 * 8), 8 * (int64_t)ctx->num_cus));
    order_inversion_kernel<KIND, false><<<dim3(blocks), dim3(256), 0, ctx->stream>>>(key.values, n, desc, inv);
    SQ_HIP(hipGetLastError());
    if (ctx->fetch_value(inv) == 0) {
      *in_order = true; // the rows are in the requested order already: nothing to do
      return false;
    }
    h = (const uint64_t *)ctx->fetch(mm->p, 16 * OW_MM_SLOTS); // (the pinned staging buffer was reused)
  }
  uint64_t imin = ~0ull, imax = 0;
  for (int q = 0; q < OW_MM_SLOTS; q++) {
    imin = std::min(imin, h[2 * q]);
    imax = std::max(imax, h[2 * q + 1]);
  }
  uint64_t range = imax - imin;
  if (imin > imax) return false;
  if (range > 0xffffffffull) { // more than 32 varying key bits: splitters instead of bits (order_wide), or the general path
    if constexpr (KIND == OKIND_I32) return false;
    else {
      // (sampled extremes are not the extremes, and this route does not need them: offsets from 0 over the whole 64-bit
      //  range do — the first and the last group then span far more values than their rows use, which their workgroups
      //  notice as one long run of equal top bits and sort on all bits.  SQLRS_ORDER_WIDE_EXACT=1: the exact pass first)
      if (optimistic) {
        const char *ex = hook("SQLRS_ORDER_WIDE_EXACT");
        if (ex && ex[0] == '1') {
          *retry_exact = true;
          return false;
        }
        imin = 0;
        range = ~0ull;
      }
      return order_wide<KIND>(ctx, key, desc, carry, n, imin, range, key_out, carry_out, perm_out, want_perm);
    }
  }
  int kbits = 1;
  while (kbits < 32 && (1ull << kbits) <= range) kbits++;
  unsigned int *oob = nullptr;
  if (optimistic) {
    // The sample's extremes lie INSIDE the true range.  (a) When the sampled keys fit the 2^kbits-aligned window they start
    // in, that window is the guess (keys `x mod 2^k`, ids counted from 0: the true range is the window, and splitting the
    // few values the sample leaves free evenly between both ends misses one of them every other time); (b) otherwise one
    // more key bit, the sampled range in the middle of the window (the split plan changes by one bit, not the cost);
    // (c) no bit left: the exact pass.
    const uint64_t win = kbits < 64 ? (1ull << kbits) : 0, base = imin & ~(win - 1);
    if (imax - base < win) {
      imin = base;
    } else if (kbits < 32) {
      kbits++;
      const uint64_t slack = ((1ull << kbits) - 1) - range;
      imin -= std::min<uint64_t>(slack / 2, imin);
    } else {
      *retry_exact = true; // (nothing was launched beyond the sample)
      return false;
    }
    oob = (unsigned int *)(mm->as<uint64_t>() + FLAG_W);
  }
  // top <= 16 bits go through HBM — as many as leave groups of ~1-2 K rows for the in-LDS finish (2e6 rows: 10 bits;
  // with 16 the finish ran 65 536 workgroups of 30 rows each: 0.64 ms of its 1.27 ms)
  int want = 1;
  while (want < 16 && (n >> want) > 2048) want++;
  // (hbm_only: every key bit goes through the HBM passes, <= 4 of them, and the finish is a streaming unpack — the second
  //  try after a group turned out larger than the in-LDS finish takes: few distinct keys spread over many bits)
  const int top = hbm_only ? kbits : std::min(kbits, want), rbits = kbits - top;
  {
    // a value with a visible share of the rows: the splitter route gives it a group of its own that is copied, not sorted
    // (order_wide works on any range; if it declines, the plan below runs as before)
    // (only when low bits are left for the in-LDS finish: with every bit sorted in HBM no group is too large)
    if (heavy_probe && rbits > 0 && heavy_cnt >= OW_HEAVY_MIN &&
        order_wide<KIND>(ctx, key, desc, carry, n, optimistic ? 0 : imin, optimistic ? ~0ull : range, key_out, carry_out, perm_out, want_perm))
      return true;
  }
  // 1. stable multi-split passes on bits [32 + rbits, 32 + kbits) of the word, LSD order
  const int64_t nblocks = ceil_div(n, OW_TILE);
  // (the usual plan — two passes, one carried column, an in-LDS finish — moves {word, value} records through BOTH passes and
  //  needs none of the four column buffers; SQLRS_ORDER_REC1=0, read per call: records out of the last pass only)
  const char *tl_e0 = hook("SQLRS_ORDER_TILED"), *rec_e0 = hook("SQLRS_ORDER_REC"), *rec1_e = hook("SQLRS_ORDER_REC1");
  const bool rec1 = NPAY == 1 && rbits > 0 && top > 8 && top <= 16 && !(tl_e0 && std::atoi(tl_e0) == 0) && !(rec_e0 && std::atoi(rec_e0) == 0) &&
                    !(rec1_e && rec1_e[0] == '0');
  BufP wa = rec1 ? nullptr : ctx->alloc(8 * (size_t)n), wb = rec1 ? nullptr : ctx->alloc(8 * (size_t)n);
  BufP pa = NPAY && !rec1 ? ctx->alloc(8 * (size_t)n) : nullptr, pb = NPAY && !rec1 ? ctx->alloc(8 * (size_t)n) : nullptr;
  BufP recbuf1 = rec1 ? ctx->alloc(16 * (size_t)n) : nullptr;
  BufP hist = ctx->alloc(4 * (size_t)(256 * nblocks)), offs = ctx->alloc(4 * (size_t)(256 * nblocks)), total = ctx->alloc(8);
  const void *src = key.values;
  const uint64_t *psrc = NPAY ? carry->v<uint64_t>() : nullptr;
  uint64_t *wdst = rec1 ? nullptr : wa->as<uint64_t>(), *walt = rec1 ? nullptr : wb->as<uint64_t>();
  uint64_t *pdst = NPAY && !rec1 ? pa->as<uint64_t>() : nullptr, *palt = NPAY && !rec1 ? pb->as<uint64_t>() : nullptr;
  bool raw = true, rec_in = false;
  dim3 g((unsigned)nblocks), b(OW_WG);
  // The last of two HBM passes (rbits > 0: an in-LDS finish follows) runs over segment-aligned tiles, which makes
  // the group boundaries a by-product of its count matrix, and (one carried column) writes 16-byte records.
  // SQLRS_ORDER_TILED=0 / SQLRS_ORDER_REC=0 (read per call) keep the plain blocks / the column form for A/B runs.
  const char *tl_e = hook("SQLRS_ORDER_TILED"), *rec_e = hook("SQLRS_ORDER_REC");
  const bool use_tiled = rbits > 0 && !(tl_e && std::atoi(tl_e) == 0);
  const bool use_rec = use_tiled && NPAY == 1 && !(rec_e && std::atoi(rec_e) == 0);
  const int64_t ntmax = nblocks + 256; // tiles of the segment-aligned pass: at most one ragged tile per segment more
  BufP recbuf = use_rec ? ctx->alloc(16 * (size_t)n) : nullptr;
  BufP firsttile, segstart, tiles2, hist2, offs2;
  bool rec_form = false, tiled_done = false;
  auto one_pass = [&](int shift) {
    ProfScope ps(ctx, "order_split");
    const bool last = shift + 8 >= 32 + kbits;
    if (last && use_tiled && !raw) {
      firsttile = ctx->alloc(4 * 257);
      segstart = ctx->alloc(8 * 257);
      tiles2 = ctx->alloc(sizeof(OwTile) * (size_t)ntmax);
      hist2 = ctx->alloc(4 * (size_t)(256 * ntmax));
      offs2 = ctx->alloc(4 * (size_t)(256 * ntmax));
      ow_tile_plan_kernel<<<dim3(1), dim3(256), 0, ctx->stream>>>(offs->as<uint32_t>(), nblocks, n, firsttile->as<uint32_t>(),
                                                                segstart->as<int64_t>());
      ow_tile_fill_kernel<<<dim3((unsigned)ceil_div(ntmax, 256)), dim3(256), 0, ctx->stream>>>(
          firsttile->as<uint32_t>(), segstart->as<int64_t>(), (uint32_t)ntmax, (OwTile *)tiles2->p);
      dim3 g2((unsigned)ntmax);
      const OwTile *tp = (const OwTile *)tiles2->p;
      if (rec_in) ow_hist_kernel<KIND, false, true, true><<<g2, b, 0, ctx->stream>>>(src, n, desc, imin, shift, ntmax, hist2->as<uint32_t>(), tp);
      else ow_hist_kernel<KIND, false, true><<<g2, b, 0, ctx->stream>>>(src, n, desc, imin, shift, ntmax, hist2->as<uint32_t>(), tp);
      SQ_HIP(hipGetLastError());
      exclusive_scan_u32(ctx, hist2->as<uint32_t>(), 256 * ntmax, nullptr, offs2->as<uint32_t>(), total->as<uint64_t>());
      if (use_rec && rec_in) {
        ow_scatter_kernel<KIND, false, NPAY, true, NPAY == 1, NPAY == 1><<<g2, b, 0, ctx->stream>>>(
            src, nullptr, n, desc, imin, shift, ntmax, offs2->as<uint32_t>(), recbuf->as<uint64_t>(), nullptr, tp, oob);
        src = recbuf->p;
        psrc = nullptr;
        rec_form = true;
      } else if (use_rec) {
        ow_scatter_kernel<KIND, false, NPAY, true, NPAY == 1><<<g2, b, 0, ctx->stream>>>(
            src, psrc, n, desc, imin, shift, ntmax, offs2->as<uint32_t>(), recbuf->as<uint64_t>(), nullptr, tp, oob);
        src = recbuf->p;
        psrc = nullptr;
        rec_form = true;
      } else {
        ow_scatter_kernel<KIND, false, NPAY, true, false><<<g2, b, 0, ctx->stream>>>(src, psrc, n, desc, imin, shift, ntmax,
                                                                                    offs2->as<uint32_t>(), wdst, pdst, tp, oob);
        src = wdst;
        psrc = pdst;
      }
      SQ_HIP(hipGetLastError());
      tiled_done = true;
      return;
    }
    if (raw) ow_hist_kernel<KIND, true><<<g, b, 0, ctx->stream>>>(src, n, desc, imin, shift, nblocks, hist->as<uint32_t>(), nullptr, oob, kbits);
    else ow_hist_kernel<KIND, false><<<g, b, 0, ctx->stream>>>(src, n, desc, imin, shift, nblocks, hist->as<uint32_t>(), nullptr);
    SQ_HIP(hipGetLastError());
    exclusive_scan_u32(ctx, hist->as<uint32_t>(), 256 * nblocks, nullptr, offs->as<uint32_t>(), total->as<uint64_t>());
    if (raw && rec1) { // records out of the raw pass: the tiled pass behind it reads one 16-byte piece per row
      ow_scatter_kernel<KIND, true, NPAY, false, NPAY == 1><<<g, b, 0, ctx->stream>>>(src, psrc, n, desc, imin, shift, nblocks, offs->as<uint32_t>(),
                                                                                     recbuf1->as<uint64_t>(), nullptr, nullptr, oob);
      SQ_HIP(hipGetLastError());
      src = recbuf1->p;
      psrc = nullptr;
      raw = false;
      rec_in = true;
      return;
    }
    if (raw) ow_scatter_kernel<KIND, true, NPAY><<<g, b, 0, ctx->stream>>>(src, psrc, n, desc, imin, shift, nblocks, offs->as<uint32_t>(), wdst, pdst, nullptr, oob);
    else ow_scatter_kernel<KIND, false, NPAY><<<g, b, 0, ctx->stream>>>(src, psrc, n, desc, imin, shift, nblocks, offs->as<uint32_t>(), wdst, pdst, nullptr, oob);
    SQ_HIP(hipGetLastError());
    src = wdst;
    psrc = pdst;
    std::swap(wdst, walt);
    std::swap(pdst, palt);
    raw = false;
  };
  // The usual plan in its look-back form (round 5): ONE histogram of the column for both passes, the passes themselves
  // chained over their tiles — no count matrices, no scans (1e8 rows: 0.24 + 0.33 ms of histograms and 0.08 of scans
  // against 0.2 for the one histogram).  SQLRS_ORDER_LB=0 (read per call): the counting form; also taken for the rest of
  // the process once a look-back spin ran out (a predecessor tile that never showed up: see ow_scatter_kernel).
  const char *lb_e = hook("SQLRS_ORDER_LB");
  static thread_local bool lb_skip = false; // (set around the one re-run after a failed attempt)
  const char *lbf_e = hook("SQLRS_ORDER_LB_TEST_FAIL"); // (test hook, read per call: treat the attempt as failed)
  const bool two_pass = rbits > 0 && top > 8 && top <= 16 && use_tiled;
  const bool lb = two_pass && (NPAY == 1 ? (rec1 && use_rec) : true) && n < (1ll << 30) && !g_order_lb_off.load() && !lb_skip && !(lb_e && lb_e[0] == '0');
  BufP ghb, lbdesc, boundb;
  bool slim = false; // the look-back form moved 12-byte records (ow_scatter_kernel<.., SLIM>): the finish reads those
  unsigned int *lbw = nullptr; // {look-back spin ran out, largest group, key outside the optimistic range}
  bool lb_done = false;
  if (lb) {
    ProfScope ps(ctx, "order_split");
    ghb = ctx->alloc(4 * 520);
    lbdesc = ctx->alloc(4 * 256 * (size_t)(nblocks + ntmax));
    boundb = ctx->alloc(4 * 256 * 256);
    SQ_HIP(hipMemsetAsync(ghb->p, 0, 4 * 520, ctx->stream));
    SQ_HIP(hipMemsetAsync(lbdesc->p, 0, 4 * 256 * (size_t)(nblocks + ntmax), ctx->stream));
    uint32_t *gh = ghb->as<uint32_t>();
    lbw = gh + 512;
    unsigned int *oob_lb = oob ? lbw + 2 : nullptr;
    const int s1 = 32 + rbits, s2 = s1 + 8;
    const unsigned gblocks = (unsigned)std::min<int64_t>(nblocks, 4 * (int64_t)ctx->num_cus);
    ow_ghist_kernel<KIND><<<dim3(gblocks), b, 0, ctx->stream>>>(src, n, desc, imin, s1, s2, nblocks, gh, oob_lb, kbits);
    uint64_t *out1 = NPAY == 1 ? recbuf1->as<uint64_t>() : wdst, *out2 = NPAY == 1 ? recbuf->as<uint64_t>() : walt; // (records / words)
    const char *two_e = hook("SQLRS_ORDER_TWO"); // (read per call: 0 = word and value side by side in LDS, two workgroups per CU)
    const bool two = NPAY == 1 && !(two_e && two_e[0] == '0');
    const char *slim_e = hook("SQLRS_ORDER_SLIM"); // (A/B hook, read per call: 0 = 16-byte records {word, value} between the passes)
    slim = two && !want_perm && !(slim_e && slim_e[0] == '0'); // 12-byte records {key offset, value}: nobody reads the row id
    if (slim)
      ow_scatter_kernel<KIND, true, NPAY, false, NPAY == 1, false, true, NPAY == 1, NPAY == 1><<<g, b, 0, ctx->stream>>>(
          src, psrc, n, desc, imin, s1, nblocks, nullptr, out1, nullptr, nullptr, oob_lb, gh, lbdesc->as<uint32_t>(), nullptr, lbw);
    else if (two)
      ow_scatter_kernel<KIND, true, NPAY, false, NPAY == 1, false, true, NPAY == 1><<<g, b, 0, ctx->stream>>>(
          src, psrc, n, desc, imin, s1, nblocks, nullptr, out1, nullptr, nullptr, oob_lb, gh, lbdesc->as<uint32_t>(), nullptr, lbw);
    else
    ow_scatter_kernel<KIND, true, NPAY, false, NPAY == 1, false, true><<<g, b, 0, ctx->stream>>>(
        src, psrc, n, desc, imin, s1, nblocks, nullptr, out1, nullptr, nullptr, oob_lb, gh, lbdesc->as<uint32_t>(), nullptr, lbw);
    if (lbf_e && lbf_e[0] == '2') { // test hook: as if a spin had run out in the first pass — its output is garbage, the flag is up
      SQ_HIP(hipMemsetAsync(out1, 0xff, (NPAY == 1 ? 16 : 8) * (size_t)n, ctx->stream));
      SQ_HIP(hipMemsetAsync(lbw, 1, 4, ctx->stream));
    }
    firsttile = ctx->alloc(4 * 257);
    segstart = ctx->alloc(8 * 257);
    tiles2 = ctx->alloc(sizeof(OwTile) * (size_t)ntmax);
    ow_tile_plan_gh_kernel<<<dim3(1), dim3(256), 0, ctx->stream>>>(gh, n, firsttile->as<uint32_t>(), segstart->as<int64_t>());
    ow_tile_fill_kernel<<<dim3((unsigned)ceil_div(ntmax, 256)), dim3(256), 0, ctx->stream>>>(
        firsttile->as<uint32_t>(), segstart->as<int64_t>(), (uint32_t)ntmax, (OwTile *)tiles2->p);
    if (slim)
      ow_scatter_kernel<KIND, false, NPAY, true, NPAY == 1, NPAY == 1, true, NPAY == 1, NPAY == 1><<<dim3((unsigned)ntmax), b, 0, ctx->stream>>>(
          out1, nullptr, n, desc, imin, s2, ntmax, nullptr, out2, nullptr, (const OwTile *)tiles2->p, oob_lb, gh + 256,
          lbdesc->as<uint32_t>() + 256 * (size_t)nblocks, boundb->as<uint32_t>(), lbw);
    else if (two)
      ow_scatter_kernel<KIND, false, NPAY, true, NPAY == 1, NPAY == 1, true, NPAY == 1><<<dim3((unsigned)ntmax), b, 0, ctx->stream>>>(
          out1, nullptr, n, desc, imin, s2, ntmax, nullptr, out2, nullptr, (const OwTile *)tiles2->p, oob_lb, gh + 256,
          lbdesc->as<uint32_t>() + 256 * (size_t)nblocks, boundb->as<uint32_t>(), lbw);
    else
    ow_scatter_kernel<KIND, false, NPAY, true, NPAY == 1, NPAY == 1, true><<<dim3((unsigned)ntmax), b, 0, ctx->stream>>>(
        out1, nullptr, n, desc, imin, s2, ntmax, nullptr, out2, nullptr, (const OwTile *)tiles2->p, oob_lb, gh + 256,
        lbdesc->as<uint32_t>() + 256 * (size_t)nblocks, boundb->as<uint32_t>(), lbw);
    SQ_HIP(hipGetLastError());
    src = out2;
    psrc = nullptr;
    rec_form = NPAY == 1;
    lb_done = true;
  } else
    for (int shift = 32 + rbits; shift < 32 + kbits || raw; shift += 8) one_pass(shift); // (>= 1 pass: the words must exist)
  const uint64_t *words = (const uint64_t *)src;
  const uint64_t *pays = psrc;
  // outputs
  key_out->dtype = key.dtype;
  key_out->length = n;
  key_out->null_count = 0;
  key_out->own_values = ctx->alloc((KIND == OKIND_I32 ? 4 : 8) * (size_t)n + 16);
  key_out->values = key_out->own_values->p;
  if (want_perm) *perm_out = ctx->alloc(4 * (size_t)n);
  uint32_t *perm = want_perm ? (*perm_out)->as<uint32_t>() : nullptr;
  if (rbits == 0) {
    if (oob && ctx->fetch_value(oob)) {
      *retry_exact = true;
      return false;
    }
    ProfScope ps(ctx, "order_finish");
    ow_unpack_kernel<KIND, NPAY><<<dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream>>>(words, n, desc, imin,
                                                                                               key_out->own_values->p, perm);
    SQ_HIP(hipGetLastError());
    if (NPAY) { // the carried column is already in final order: adopt the buffer it sits in
      carry_out->dtype = carry->dtype;
      carry_out->length = n;
      carry_out->null_count = 0;
      carry_out->own_values = (pays == pa->as<uint64_t>()) ? pa : pb;
      carry_out->values = carry_out->own_values->p;
    }
    return true;
  }
  // 2. groups = distinct values of the top bits
  const uint32_t G = 1u << top;
  BufP gstart = ctx->alloc(4 * (size_t)G), gend = ctx->alloc(4 * ((size_t)G + 1)); // gend[G] = largest group
  if (!lb_done) { // (the look-back form's table kernel writes every entry, its largest group goes to lbw[1])
    SQ_HIP(hipMemsetAsync(gstart->p, 0xff, 4 * (size_t)G, ctx->stream));
    SQ_HIP(hipMemsetAsync(gend->p, 0, 4 * ((size_t)G + 1), ctx->stream));
  }
  {
    ProfScope ps(ctx, "order_groups");
    if (lb_done) {
      ow_group_table_lb_kernel<<<dim3(G >> 8), dim3(256), 0, ctx->stream>>>(boundb->as<uint32_t>(), ghb->as<uint32_t>(), gstart->as<uint32_t>(),
                                                                       gend->as<uint32_t>(), lbw);
    } else if (tiled_done) { // (two passes: 8 bits, then top - 8)
      ow_group_table_kernel<<<dim3(G >> 8), dim3(256), 0, ctx->stream>>>(offs2->as<uint32_t>(), ntmax, firsttile->as<uint32_t>(), n,
                                                                    gstart->as<uint32_t>(), gend->as<uint32_t>());
    } else {
      ow_group_bounds_kernel<<<dim3((unsigned)std::min<int64_t>(ceil_div(n, 256 * 8), 16 * (int64_t)ctx->num_cus)), dim3(256), 0, ctx->stream>>>(
          words, n, 32 + rbits, gstart->as<uint32_t>(), gend->as<uint32_t>());
      ow_group_max_kernel<<<dim3((unsigned)ceil_div(G, 256)), dim3(256), 0, ctx->stream>>>(gstart->as<uint32_t>(), gend->as<uint32_t>(), G,
                                                                                        gend->as<uint32_t>() + G);
    }
    SQ_HIP(hipGetLastError());
  }
  uint32_t max_group;
  if (lb_done) { // one round trip for the three
    const uint32_t *hv = (const uint32_t *)ctx->fetch(lbw, 12);
    if (hv[0] || (lbf_e && lbf_e[0] == '1')) { // a look-back spin ran out: nothing of this attempt is valid; the counting form from here on
      if (hv[0] && !(lbf_e && lbf_e[0] == '2')) { // (not for the test hook's forced failure)
        g_order_lb_off.store(true);
        ctx->order_lb_fallbacks++;
      }
      struct Skip {
        Skip() { lb_skip = true; }
        ~Skip() { lb_skip = false; }
      } skip;
      return order_fast_impl<KIND, NPAY>(ctx, key, desc, carry, n, key_out, carry_out, perm_out, want_perm, optimistic, retry_exact, nullptr,
                                         hbm_only, retry_hbm_only);
    }
    if (oob && hv[2]) {
      *retry_exact = true;
      return false;
    }
    max_group = hv[1];
  } else if (oob) { // one round trip for both: the largest group and the verdict on the optimistic key range
    SQ_HIP(hipMemcpyAsync(mm->as<uint32_t>() + 2 * FLAG_W + 1, gend->as<uint32_t>() + G, 4, hipMemcpyDeviceToDevice, ctx->stream));
    const uint32_t *hv = (const uint32_t *)ctx->fetch(mm->as<uint64_t>() + FLAG_W, 8);
    if (hv[0]) {
      *retry_exact = true;
      return false;
    }
    max_group = hv[1];
  } else
    max_group = ctx->fetch_value(gend->as<uint32_t>() + G);
  if (max_group > FIN_CAP) { // heavily repeated top bits: all bits through HBM passes instead (no limit on a group there)
    if (retry_hbm_only) *retry_hbm_only = true;
    return false;
  }
  if (NPAY) {
    carry_out->dtype = carry->dtype;
    carry_out->length = n;
    carry_out->null_count = 0;
    carry_out->own_values = ctx->alloc(8 * (size_t)n + 16);
    carry_out->values = carry_out->own_values->p;
  }
  {
    ProfScope ps(ctx, "order_finish");
    uint64_t *po = NPAY ? carry_out->own_values->as<uint64_t>() : nullptr;
    const char *fc_e = hook("SQLRS_ORDER_FINISH_COUNT"); // (A/B hook, read per call: 0 = the finish sorts on its low bits with LSD passes only)
    const int fin_count = !(fc_e && fc_e[0] == '0');
#define SQ_FIN(RR)                                                                                                   \
  do {                                                                                                               \
    auto kfn = ow_finish_kernel<KIND, NPAY, RR>;                                                                     \
    if (NPAY == 1 && rec_form) kfn = ow_finish_kernel<KIND, NPAY, RR, NPAY == 1>;                                    \
    if (NPAY == 1 && rec_form && slim) kfn = ow_finish_kernel<KIND, NPAY, RR, NPAY == 1, NPAY == 1>;                 \
    const size_t lds = (size_t)RR * FIN_WG * 8 * (1 + NPAY) + 4 * (FIN_WAVES * 256 + 256);                           \
    if (lds > 64 * 1024) allow_big_lds(ctx, kfn);                                                                    \
    kfn<<<dim3(G), dim3(FIN_WG), lds, ctx->stream>>>(words, pays, gstart->as<uint32_t>(), gend->as<uint32_t>(), rbits, desc, imin, \
                                                     key_out->own_values->p, po, perm, fin_count);                   \
  } while (0)
    if (max_group <= 8 * FIN_WG) SQ_FIN(8);
    else if (max_group <= 16 * FIN_WG) SQ_FIN(16);
    else SQ_FIN(24);
#undef SQ_FIN
    SQ_HIP(hipGetLastError());
  }
  return true;
}

// ORDER BY one key column (int64 / float64 / int32, no NULLs) of >= 2^20 rows, optionally carrying one
// 8-byte column without NULLs; `perm` (row ids in output order) is produced when asked for.  Returns false
// when the shape or the data do not fit (nothing has been produced then).
// `in_order` (optional, out): set when the rows are in the requested order already — the call then returns false having
// produced nothing, and the caller emits its input as it is
bool order_fast(Ctx *ctx, const DCol &key, int desc, const DCol *carry, int64_t n, DCol *key_out, DCol *carry_out,
                BufP *perm, bool want_perm, bool *in_order) {
  if (in_order) *in_order = false;
  if (n < (1 << 20) || n > 0xffffffffll || key.stride == 0 || (key.validity && key.null_count != 0)) return false;
  if (carry && (width_of(carry->dtype) != 8 || carry->stride == 0 || (carry->validity && carry->null_count != 0))) return false;
  // optimistic key range for large columns (SQLRS_ORDER_SAMPLE, read per call: 0 = always the exact pass, 1 = always sampled)
  const char *smp_e = hook("SQLRS_ORDER_SAMPLE");
  const bool optimistic = smp_e ? std::atoi(smp_e) != 0 : n >= (1ll << 24); // (1 = whatever the size: tests)
#define SQ_OF(K)                                                                                                     \
  do {                                                                                                               \
    bool retry = false, hbm = false;                                                                                 \
    bool ok = carry ? order_fast_impl<K, 1>(ctx, key, desc, carry, n, key_out, carry_out, perm, want_perm, optimistic, &retry, in_order, false, &hbm) \
                    : order_fast_impl<K, 0>(ctx, key, desc, nullptr, n, key_out, carry_out, perm, want_perm, optimistic, &retry, in_order, false, &hbm); \
    if (ok || (!retry && !hbm)) return ok;                                                                           \
    if (retry) {                                                                                                     \
      retry = false;                                                                                                 \
      ok = carry ? order_fast_impl<K, 1>(ctx, key, desc, carry, n, key_out, carry_out, perm, want_perm, false, &retry, nullptr, false, &hbm) \
                 : order_fast_impl<K, 0>(ctx, key, desc, nullptr, n, key_out, carry_out, perm, want_perm, false, &retry, nullptr, false, &hbm); \
      if (ok || !hbm) return ok;                                                                                     \
    }                                                                                                                \
    return carry ? order_fast_impl<K, 1>(ctx, key, desc, carry, n, key_out, carry_out, perm, want_perm, false, &retry, nullptr, true) \
                 : order_fast_impl<K, 0>(ctx, key, desc, nullptr, n, key_out, carry_out, perm, want_perm, false, &retry, nullptr, true); \
  } while (0)
  switch (key.dtype) {
  case SQLRS_INT64: SQ_OF(OKIND_I64);
  case SQLRS_FLOAT64: SQ_OF(OKIND_F64);
  case SQLRS_INT32: SQ_OF(OKIND_I32);
  default: return false;
  }
#undef SQ_OF
}

// ==== several integer keys, nullable integer keys ====================================================================
// ORDER BY a, b [, c, d] over plain int64 / int32 columns (order.rs:27-66: lexsort over the sort columns, NULLs first
// whatever the direction): when the keys' ranges together need <= 64 bits, the rows are ordered by ONE composite key
//     field_c = [valid bit, only for a column with NULLs :] asc ? value - min_c : max_c - value      (a NULL: all zero)
//     composite = field_0 : field_1 : ...   (most significant first)
// through the single-key routes above (<= 32 bits: the narrow one; more: the splitter route), and the key columns of the
// result — values and validity — are decoded from the sorted composite instead of being gathered.  The general path runs
// one stable radix sort of (key, row id) pairs per key, last key first, and gathers every column.
struct CompKeys {
  const void *vals[4];
  const uint64_t *valid[4]; // nullptr = no NULLs in this key
  uint64_t imin[4], imax[4];
  int kind[4], bits[4], desc[4]; // bits: of the value part (the valid bit sits above it)
  int nk;
};
__device__ __forceinline__ uint64_t comp_image(const CompKeys &ck, int c, int64_t i) {
  return ck.kind[c] == OKIND_I64 ? order_image<OKIND_I64>(ck.vals[c], i, 0) : order_image<OKIND_I32>(ck.vals[c], i, 0);
}
__global__ __launch_bounds__(256) void comp_build_kernel(CompKeys ck, int64_t n, int64_t *__restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t comp = 0;
  for (int c = 0; c < ck.nk; c++) {
    const int fb = ck.bits[c] + (ck.valid[c] ? 1 : 0);
    if (fb == 0) continue; // (a constant column; a shift by 64 would also be undefined)
    uint64_t field = 0;
    if (!ck.valid[c] || ((ck.valid[c][i >> 6] >> (i & 63)) & 1ull)) {
      const uint64_t img = comp_image(ck, c, i);
      field = (ck.desc[c] ? ck.imax[c] - img : img - ck.imin[c]) | (ck.valid[c] ? 1ull << ck.bits[c] : 0ull);
    }
    comp = (fb < 64 ? comp << fb : 0) | field;
  }
  out[i] = (int64_t)(comp ^ (1ull << 63)); // (as int64 whose order-preserving image is the composite itself)
}
// one key column out of the sorted composite; out_valid (nullable keys): one word per wave
__global__ __launch_bounds__(256) void comp_decode_kernel(const int64_t *__restrict__ comp, int64_t n, int shift, int bits, int nullable,
                                                          int desc, uint64_t imin, uint64_t imax, int kind, void *__restrict__ out,
                                                          uint64_t *__restrict__ out_valid) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int fb = bits + nullable;
  bool valid = false;
  int64_t v = 0;
  if (i < n) {
    const uint64_t c = (uint64_t)comp[i] ^ (1ull << 63);
    const uint64_t field = fb == 0 ? 0 : ((c >> shift) & (fb < 64 ? (1ull << fb) - 1 : ~0ull));
    valid = !nullable || ((field >> bits) & 1ull);
    const uint64_t f = bits == 0 ? 0 : (field & (bits < 64 ? (1ull << bits) - 1 : ~0ull));
    if (valid) v = ordered_to_i64(desc ? imax - f : imin + f);
    if (kind == OKIND_I32) ((int32_t *)out)[i] = (int32_t)v;
    else ((int64_t *)out)[i] = v;
  }
  if (out_valid) {
    const uint64_t m = __ballot(valid);
    if (lane_id() == 0 && (i >> 6) < (n + 63) / 64) out_valid[i >> 6] = m;
  }
}

bool order_composite(Ctx *ctx, const std::vector<const DCol *> &keys, const std::vector<int> &desc, const DCol *carry, int64_t n,
                     std::vector<DCol> *keys_out, DCol *carry_out, BufP *perm, bool want_perm, bool *in_order) {
  if (in_order) *in_order = false;
  const int nk = (int)keys.size();
  if (nk < 1 || nk > 4 || n < (1 << 20) || n > 0xffffffffll) return false;
  if (const char *e = hook("SQLRS_ORDER_COMPOSITE")) // (A/B hook, read per call: 0 = the general path)
    if (e[0] == '0') return false;
  CompKeys ck{};
  ck.nk = nk;
  bool any_nullable = false;
  for (int c = 0; c < nk; c++) {
    const DCol &k = *keys[(size_t)c];
    if ((k.dtype != SQLRS_INT64 && k.dtype != SQLRS_INT32) || k.stride == 0) return false;
    ck.vals[c] = k.values;
    ck.valid[c] = (k.validity && k.null_count != 0) ? k.validity : nullptr;
    any_nullable |= ck.valid[c] != nullptr;
    ck.kind[c] = k.dtype == SQLRS_INT64 ? OKIND_I64 : OKIND_I32;
    ck.desc[c] = desc[(size_t)c];
  }
  if (nk == 1 && !any_nullable) return false; // (the single-key route has had its say)
  // ranges of the keys (over all rows: what sits under a NULL can only widen them): one pass per column, one round trip for all
  constexpr size_t MM_WORDS = 2 * OW_MM_SLOTS + 2;
  BufP mm = ctx->alloc(8 * MM_WORDS * (size_t)nk);
  {
    ProfScope ps(ctx, "order_minmax");
    const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n, (int64_t)256 * 8), 8 * (int64_t)ctx->num_cus));
    for (int c = 0; c < nk; c++) {
      unsigned long long *m = mm->as<unsigned long long>() + MM_WORDS * (size_t)c;
      order_minmax_init_kernel<<<dim3(1), dim3(128), 0, ctx->stream>>>(m);
      if (ck.kind[c] == OKIND_I64) order_minmax_kernel<OKIND_I64><<<dim3(blocks), dim3(256), 0, ctx->stream>>>(ck.vals[c], n, 0, m, 1);
      else order_minmax_kernel<OKIND_I32><<<dim3(blocks), dim3(256), 0, ctx->stream>>>(ck.vals[c], n, 0, m, 1);
    }
    SQ_HIP(hipGetLastError());
  }
  const uint64_t *h = (const uint64_t *)ctx->fetch(mm->p, 8 * MM_WORDS * (size_t)nk);
  int total = 0;
  for (int c = 0; c < nk; c++) {
    uint64_t lo = ~0ull, hi = 0;
    for (int q = 0; q < OW_MM_SLOTS; q++) {
      lo = std::min(lo, h[MM_WORDS * (size_t)c + 2 * q]);
      hi = std::max(hi, h[MM_WORDS * (size_t)c + 2 * q + 1]);
    }
    if (lo > hi) return false;
    ck.imin[c] = lo;
    ck.imax[c] = hi;
    ck.bits[c] = hi == lo ? 0 : 64 - __builtin_clzll(hi - lo);
    total += ck.bits[c] + (ck.valid[c] ? 1 : 0);
  }
  if (total > 64) return false; // the composite does not fit one word: general path
  std::vector<int64_t> nulls((size_t)nk, 0);
  for (int c = 0; c < nk; c++)
    if (ck.valid[c]) nulls[(size_t)c] = count_nulls(ctx, *keys[(size_t)c]);
  DCol comp;
  comp.dtype = SQLRS_INT64;
  comp.length = n;
  comp.null_count = 0;
  comp.own_values = ctx->alloc(8 * (size_t)n + 16);
  comp.values = comp.own_values->p;
  {
    ProfScope ps(ctx, "order_keys");
    comp_build_kernel<<<dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream>>>(ck, n, comp.own_values->as<int64_t>());
    SQ_HIP(hipGetLastError());
  }
  DCol sorted;
  if (!order_fast(ctx, comp, 0, carry, n, &sorted, carry_out, perm, want_perm, in_order)) return false;
  comp = DCol(); // (its buffer goes back to the pool before the outputs are allocated)
  {
    ProfScope ps(ctx, "order_keys");
    keys_out->clear();
    int shift = total;
    for (int c = 0; c < nk; c++) {
      const int nullable = ck.valid[c] ? 1 : 0;
      shift -= ck.bits[c] + nullable;
      DCol o;
      o.dtype = keys[(size_t)c]->dtype;
      o.length = n;
      o.null_count = nulls[(size_t)c];
      o.own_values = ctx->alloc((ck.kind[c] == OKIND_I32 ? 4 : 8) * (size_t)n + 16);
      o.values = o.own_values->p;
      if (nullable) {
        o.own_validity = ctx->alloc(8 * (size_t)ceil_div(n, 64) + 16);
        o.validity = o.own_validity->as<uint64_t>();
      }
      comp_decode_kernel<<<dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream>>>(
          (const int64_t *)sorted.values, n, shift, ck.bits[c], nullable, ck.desc[c], ck.imin[c], ck.imax[c], ck.kind[c], o.own_values->p,
          nullable ? o.own_validity->as<uint64_t>() : nullptr);
      keys_out->push_back(o);
    }
    SQ_HIP(hipGetLastError());
  }
  return true;
}

} // namespace sq
