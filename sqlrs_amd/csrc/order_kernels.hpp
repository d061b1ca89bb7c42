// order_kernels.hpp — the device side of order_fast.hip (round 6: moved out of the route file, which keeps the host logic; one
// translation unit as before): order-preserving images, key range / inversion / heavy-value probes, the stable 8-bit multi-split
// passes over HBM in their counting and look-back forms (`ow_*`), the in-LDS finish (LSD passes or bucket + count), and the same for
// the splitter route of keys with more than 32 varying bits (`owk_*`).  order.rs:15-67; the route itself is described at the top of
// order_fast.hip.
#pragma once
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
#ifndef OW_WG_N
#define OW_WG_N 512 // (threads of a split pass's workgroup; 1024 x 4 rows = the same tile on sixteen waves: tools/order_two_builds.py)
#endif
constexpr int OW_WG = OW_WG_N, OW_WAVES = OW_WG / 64, OW_ITEMS = OW_ITEMS_N, OW_TILE = OW_WG * OW_ITEMS;
constexpr int OW_TWO_WGS = OW_WG > 512 ? 2 : (OW_ITEMS <= 8 ? 3 : 2); // workgroups per CU the TWO form is compiled for (its one LDS tile: 8 bytes x OW_TILE)

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
  static_assert(OW_WG >= 512, "one counter per thread");
  if (threadIdx.x < 512) h[threadIdx.x] = 0;
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
  if (threadIdx.x < 512 && h[threadIdx.x]) atomicAdd(&ghist[threadIdx.x], h[threadIdx.x]);
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

} // namespace sq
