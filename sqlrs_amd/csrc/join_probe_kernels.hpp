// join_probe_kernels.hpp — the device side of join.hip (round 6: moved out of the operator file, which keeps the host logic; one
// translation unit as before).  The open-addressing table (insert / count / fill), the direct-address table (build without a host
// round trip, the bit-packed copy, the all-hit and the compacting probe kernels), the existence bitmap of a key-only build side and
// the outer-join marks.  hash_join.rs:146-323; what each kernel replaces is cited at the kernel.
#pragma once
#include "common.hpp"
#include "device_utils.hpp"
#include "prims.hpp"
#include "join_kernels.hpp"

namespace sq {


__global__ void table_init_kernel(Slot *t, int64_t n) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) {
    t[i].key = EMPTY_KEY;
    t[i].head = 0;
    t[i].count = 0;
  }
}

// one build row per lane
__global__ __launch_bounds__(BLOCK) void join_insert_kernel(const uint64_t *__restrict__ keys,
                                                            const uint64_t *__restrict__ validity,
                                                            int64_t n, Slot *table, uint64_t mask,
                                                            uint32_t *__restrict__ row_slot,
                                                            int *dup_flag) {
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= n) return;
  const uint64_t cap = mask + 1;
  uint64_t key = keys[r];
  uint64_t s;
  if (validity && !((validity[r >> 6] >> (r & 63)) & 1))
    s = cap; // all NULL keys share one slot (hash_utils.rs:91-104)
  else if (key == EMPTY_KEY)
    s = cap + 1;
  else {
    s = mix64(key) & mask;
    while (true) {
      unsigned long long cur = __hip_atomic_load(&table[s].key, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
      if (cur == key) break;
      if (cur == EMPTY_KEY) {
        unsigned long long prev = atomicCAS(&table[s].key, EMPTY_KEY, (unsigned long long)key);
        if (prev == EMPTY_KEY || prev == key) break;
      }
      s = (s + 1) & mask;
    }
  }
  uint32_t old = atomicAdd(&table[s].count, 1u);
  if (old) *dup_flag = 1;
  table[s].head = (uint32_t)r; // final only when every key is unique
  row_slot[r] = (uint32_t)s;
}

__global__ void slot_counts_kernel(const Slot *__restrict__ t, int64_t n, uint32_t *__restrict__ c) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) c[i] = t[i].count;
}
__global__ void slot_heads_kernel(Slot *__restrict__ t, int64_t n, const uint32_t *__restrict__ h) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) t[i].head = h[i];
}
__global__ void u32_to_u64_kernel(const uint32_t *__restrict__ in, int64_t n, uint64_t *__restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}


// returns the slot of `key` (count may be 0 for the two reserved slots) or count==0 on a miss
__device__ __forceinline__ Slot probe_slot(const Slot *__restrict__ table, uint64_t mask, uint64_t key,
                                           bool is_null) {
  const uint64_t cap = mask + 1;
  if (is_null) return load_slot(&table[cap]);
  if (key == EMPTY_KEY) return load_slot(&table[cap + 1]);
  uint64_t s = mix64(key) & mask;
  while (true) {
    Slot sl = load_slot(&table[s]);
    if (sl.key == key) return sl;
    if (sl.key == EMPTY_KEY) {
      sl.count = 0;
      return sl;
    }
    s = (s + 1) & mask;
  }
}

// pass 1: pairs emitted by each probe row (Right/Full: an unmatched row emits one pair)
__global__ __launch_bounds__(BLOCK) void join_count_kernel(const uint64_t *__restrict__ keys,
                                                           const uint64_t *__restrict__ validity,
                                                           int64_t n, const Slot *__restrict__ table,
                                                           uint64_t mask, int outer_right,
                                                           uint32_t *__restrict__ counts, uint2 *__restrict__ match, int grouped) {
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  uint32_t c = 0;
  if (r < n) {
    bool is_null = validity && !((validity[r >> 6] >> (r & 63)) & 1);
    Slot s = probe_slot(table, mask, keys[r], is_null);
    c = s.count;
    match[r] = make_uint2(s.head, c); // what the fill pass needs: no second probe
    if (outer_right && c == 0) c = 1;
    if (!grouped) counts[r] = c;
  }
  if (grouped) { // the fill pass scans inside its 64-row group itself: only the groups' sums are scanned globally
    const uint32_t wsum = wave_sum_u32(c);
    if (lane_id() == 0 && (r & ~63ll) < n) counts[r >> 6] = wsum;
  }
}

// pass 2, wave-cooperative: a wave owns 64 consecutive probe rows and writes THEIR pairs as one contiguous
// range of output positions, 64 at a time — lane t of a step finds the row that owns output t by a 6-step
// search over the wave's inclusive scan of the per-row counts (shuffles), so consecutive lanes store
// consecutive pairs.  (One lane per probe row, each walking its own run of `count` pairs, stored at the
// random-store rate: 1.8 ms for 8e7 pairs; this form: see DESIGN.md.)
// (Round 6 measured FOUR 64-row groups per wave, their match words and offsets in flight together: 1.74 -> 2.00 ms for 4e8 pairs —
//  the pass lives on the number of waves that have stores in flight, not on the latency in front of them.  Four 64-output steps of ONE
//  group per trip, their owner searches and rows_by_slot loads issued before the stores: 1.60 -> 1.67 ms.)
__global__ __launch_bounds__(BLOCK) void join_fill_expand_kernel(
    const uint2 *__restrict__ match, int64_t n, int unique, int outer_right, const uint32_t *__restrict__ rows_by_slot,
    const uint64_t *__restrict__ offsets, uint64_t *__restrict__ left_idx, uint32_t *__restrict__ right_idx,
    uint8_t *__restrict__ left_valid_bytes, int grouped) {
  const int lane = lane_id();
  const int64_t wbase = (blockIdx.x * (int64_t)WAVES_PER_BLOCK + wave_id()) * 64;
  if (wbase >= n) return;
  const int64_t r = wbase + lane;
  uint2 m = r < n ? match[r] : make_uint2(0u, 0u);
  const bool hit = m.y != 0;
  uint32_t cnt = m.y;
  if (outer_right && r < n && cnt == 0) cnt = 1; // (NULL, row)  hash_join.rs:241-246
  const uint32_t incl = wave_iscan_u32(cnt), excl = incl - cnt;
  const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
  const uint64_t obase = offsets[grouped ? wbase >> 6 : wbase]; // (grouped: one scanned offset per 64-row group)
  for (uint32_t t0 = 0; t0 < total; t0 += 64) {
    const uint32_t t = t0 + lane;
    // owner of output t = number of rows of the wave whose inclusive scan is <= t
    int pos = 0;
#pragma unroll
    for (int sstep = 32; sstep >= 1; sstep >>= 1) {
      const uint32_t v = (uint32_t)__shfl((int)incl, pos + sstep - 1, 64);
      if (v <= t) pos += sstep;
    }
    pos = min(pos, 63);
    const uint32_t j = t - (uint32_t)__shfl((int)excl, pos, 64);
    const uint32_t head = (uint32_t)__shfl((int)m.x, pos, 64);
    const bool phit = __shfl((int)hit, pos, 64) != 0;
    if (t < total) {
      const uint64_t o = obase + t;
      uint64_t l = 0;
      if (phit) l = unique ? head : rows_by_slot[head + j];
      __builtin_nontemporal_store(l, &left_idx[o]);
      __builtin_nontemporal_store((uint32_t)(wbase + pos), &right_idx[o]);
      if (left_valid_bytes) left_valid_bytes[o] = phit ? 1 : 0;
    }
  }
}

// Single-pass probe for UNIQUE build keys (the PK-FK case), Inner/Left: a probe row emits at
// most one pair, so the probe is an order-preserving compaction: one table lookup per row,
// ranks from ballots, global offset from the decoupled look-back.  Tile = 2048 probe rows;
// each lane has 8 independent lookups in flight.
constexpr int JP_ITEMS = 8;
constexpr int JP_TILE = BLOCK * JP_ITEMS;

// Direct-address table for build keys that are unique and cover a small integer range (the
// dense surrogate keys of a dimension table): heads[key - kmin] = build row.  4 bytes per
// possible key instead of a 16-byte hash slot at load factor <= 2/3: a 1e6-key dimension needs
// 4 MiB, which one XCD's L2 holds (265 G lookups/s instead of 66 G/s, profiles/r01_ubench).
// Round 5: the probe kernels read a BIT-PACKED copy of the table when there is one — `bits` = ceil(log2(rows + 1)) bits per
// possible key (all ones = empty), entry e at bit e * bits, fetched with ONE unaligned 4-byte load (bits <= 25).  20 bits
// instead of 32 for 1e6 build rows: 2.4 MiB instead of 3.8, which is what lets the table stay in its XCD's 4 MiB L2 NEXT TO
// the key stream and the pair stores (the probe's time is the sum of its L1 miss latencies over 64 miss slots per CU,
// profiles/r02_probe_pmc_ta.txt, and a lookup that has left L2 holds its slot 3-5x longer).  tools/ubench2.hip, same memory
// work and nothing else, 1e8 keys against 1e6: 0.654 ms with 4-byte entries, 0.516 with 3-byte, 0.520 with 20-bit ones
// (profiles/r05a_ubench2.txt, r05b_ubench2.txt); 2e6 build rows: 1.08 -> 0.78 ms.
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
struct DenseTable {
  const uint32_t *heads;
  uint64_t kmin, range;
  uint32_t null_head; // build row whose key is NULL (NULL = NULL matches) or DENSE_EMPTY
  const uint8_t *packed = nullptr; // bit-packed copy of heads[0 .. range + 2) or null
  uint32_t bits = 0, pmask = 0;    // bits per entry, (1 << bits) - 1 = the packed form of DENSE_EMPTY
  // the build's verdict is still on the device (sqlrs_hash_join::dense_pending): kmin / range are read from `st` by the
  // kernel (dense_table_from_device), which raises bit 1 of its miss flag when the build keys are not a unique dense set
  const unsigned long long *st = nullptr;
  uint64_t st_max_range = 0, st_rows = 0;
};
__device__ __forceinline__ uint32_t dense_packed_raw(const uint8_t *__restrict__ packed, uint32_t bits, uint32_t pmask, uint32_t e) {
  const uint32_t bit = e * bits; // (the host packs only tables of less than 2^32 bits)
  return (*(const u32_unaligned *)(packed + (bit >> 3)) >> (bit & 7)) & pmask;
}
// entry d (d <= range + 1) of the table: the build row or DENSE_EMPTY
__device__ __forceinline__ uint32_t dense_get(const DenseTable &dt, uint64_t d) {
  if (dt.packed) {
    const uint32_t v = dense_packed_raw(dt.packed, dt.bits, dt.pmask, (uint32_t)d);
    return v == dt.pmask ? DENSE_EMPTY : v;
  }
  return dt.heads[d];
}

// (defined with the build kernels below)
__device__ __forceinline__ bool dense_table_from_device(DenseTable &dt);

template <bool DENSE>
__global__ __launch_bounds__(BLOCK) void join_probe_unique_kernel(
    const uint64_t *__restrict__ keys, const uint64_t *__restrict__ validity, int64_t n,
    int64_t num_tiles, const Slot *__restrict__ table, uint64_t mask, DenseTable dt,
    uint64_t *__restrict__ left_idx, uint32_t *__restrict__ right_idx, uint64_t *desc, unsigned *ticket,
    uint64_t *total, int use_ticket) {
  unsigned *timeout = use_ticket ? nullptr : ticket + 1;
  __shared__ int64_t s_tile;
  __shared__ uint32_t s_wave[WAVES_PER_BLOCK];
  __shared__ uint64_t s_excl;
  // one ticket buys LB_TILES_PER_TICKET consecutive tiles (a single atomic counter sustains only
  // ~88 tickets/us: a ticket per 2048-row tile would cost >= 0.55 ms per 1e8 probe rows)
  if (threadIdx.x == 0) s_tile = use_ticket ? (int64_t)atomicAdd(ticket, 1u) : (int64_t)blockIdx.x;
  __syncthreads();
  const int64_t tile0 = s_tile;
  const int lane = lane_id(), w = wave_id();
  const uint64_t cap = mask + 1;
  for (int sub = 0; sub < LB_TILES_PER_TICKET; sub++) {
    const int64_t tile = tile0 + sub;
    if (tile >= num_tiles) break;
    const int64_t wrow = tile * JP_TILE + (int64_t)w * (64 * JP_ITEMS);
    uint64_t k[JP_ITEMS];
    bool isnull[JP_ITEMS];
#pragma unroll
    for (int j = 0; j < JP_ITEMS; j++) {
      int64_t r = wrow + j * 64 + lane;
      k[j] = (r < n) ? __builtin_nontemporal_load(&keys[r]) : 0; // streamed once: keep the table cached
      isnull[j] = (r < n) && validity && !((validity[r >> 6] >> (r & 63)) & 1);
    }
    uint32_t head[JP_ITEMS];
    uint64_t m[JP_ITEMS];
    uint32_t wave_cnt = 0;
    if (DENSE) {
#pragma unroll
      for (int j = 0; j < JP_ITEMS; j++) { // 8 independent 4-byte loads in flight per lane
        int64_t r = wrow + j * 64 + lane;
        uint64_t d = k[j] - dt.kmin;
        head[j] = DENSE_EMPTY;
        if (r < n) head[j] = isnull[j] ? dt.null_head : (d < dt.range ? dense_get(dt, d) : DENSE_EMPTY);
      }
#pragma unroll
      for (int j = 0; j < JP_ITEMS; j++) {
        m[j] = __ballot(head[j] != DENSE_EMPTY);
        wave_cnt += (uint32_t)__popcll(m[j]);
      }
    } else {
      // first probe of all 8 rows issued back to back (8 independent 16-byte loads in flight per
      // lane); only the rare collision chains continue one at a time
      uint64_t slot[JP_ITEMS];
      Slot sl[JP_ITEMS];
#pragma unroll
      for (int j = 0; j < JP_ITEMS; j++) {
        slot[j] = isnull[j] ? cap : (k[j] == EMPTY_KEY ? cap + 1 : (mix64(k[j]) & mask));
        sl[j] = load_slot(&table[slot[j]]);
      }
#pragma unroll
      for (int j = 0; j < JP_ITEMS; j++) {
        int64_t r = wrow + j * 64 + lane;
        bool hit = false;
        if (r < n) {
          if (slot[j] >= cap) {
            hit = sl[j].count != 0; // reserved slots: NULL keys / key == EMPTY_KEY
          } else {
            while (sl[j].key != k[j] && sl[j].key != EMPTY_KEY) {
              slot[j] = (slot[j] + 1) & mask;
              sl[j] = load_slot(&table[slot[j]]);
            }
            hit = sl[j].key == k[j];
          }
        }
        head[j] = sl[j].head;
        m[j] = __ballot(hit);
        wave_cnt += (uint32_t)__popcll(m[j]);
      }
    }
    if (lane == 0) s_wave[w] = wave_cnt;
    __syncthreads();
    if (w == 0) {
      uint64_t agg = (uint64_t)s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
      uint64_t excl = lookback_wave(desc, tile, agg, timeout);
      if (lane == 0) {
        s_excl = excl;
        if (tile == num_tiles - 1) *total = excl + agg;
      }
    }
    __syncthreads();
    uint64_t pos = s_excl;
    for (int q = 0; q < w; q++) pos += s_wave[q];
#pragma unroll
    for (int j = 0; j < JP_ITEMS; j++) {
      if ((m[j] >> lane) & 1) {
        uint64_t o = pos + mbcnt(m[j]);
        __builtin_nontemporal_store((uint64_t)head[j], &left_idx[o]);
        __builtin_nontemporal_store((uint32_t)(wrow + j * 64 + lane), &right_idx[o]);
      }
      pos += (uint32_t)__popcll(m[j]);
    }
    __syncthreads(); // s_wave / s_excl are reused by the next tile
  }
}

// Direct-address probe with large tiles: 8 worker waves x 32 rows per lane = 16384 probe rows per
// block, plus a scan wave that owns the decoupled look-back (same structure and same reason as
// filter_cmp_const_kernel, select.hip: with 2048-row tiles the probe ran at the look-back's pace,
// ~47 tiles/us = 97 Grows/s, not at the memory system's).
// (Round 6, where the compacting probe's time goes — C3 half-hit, 1e8 probe rows, probe alone, tools/build_obj_variant.sh with
//  -DJD_DBG / -DJD_WAVES_N / -DJD_ITEMS_N / -DJD_OCC: as shipped 0.56-0.59 ms; without the table lookups 0.40; without the
//  look-back (wrong offsets) 0.50; without both 0.34 = the 1.4 GB of keys and pairs at 4.1 TB/s.  8 waves x 16 rows per lane at
//  four workgroups per CU: 0.43 without the look-back — twice the waves hide the lookups — but 0.71 with it (twice the tiles on
//  the chain); 8 x 16 / 8 x 24 at two per CU: 0.70 / 0.54.  More worker waves on the SAME number of tiles — 15 x 16, 11 x 24,
//  12 x 20 at two per CU, 15 x 24 at one: 0.54-0.55, 15 x 32: 0.60.  The look-back's cost is the cross-XCD latency of a
//  predecessor's word times the rounds a tile waits, and only long tiles amortise it; nothing here is worth a changed default.)
#ifndef JD_WAVES_N
#define JD_WAVES_N 8
#endif
constexpr int JD_WAVES = JD_WAVES_N;
#ifndef JD_ITEMS_N
#define JD_ITEMS_N 32
#endif
#ifndef JD_OCC
#define JD_OCC 2
#endif
#ifndef JD_DBG
#define JD_DBG 0
#endif
constexpr int JD_ITEMS = JD_ITEMS_N;
constexpr int JD_TILE = JD_WAVES * JD_ITEMS * 64;
constexpr int JD_BLOCK = (JD_WAVES + 1) * 64;

// ---- every probe row has a partner (the PK-FK join): no compaction ---------------------------------------------
// Pair i of an Inner join whose probe rows ALL match unique build keys is (heads[key[i] - kmin], i): the output
// position is the row number, so the tile counts, the look-back chain and the two barriers of the kernel below
// have nothing to decide.  This kernel does exactly the memory work of the probe — the key stream, one table
// lookup per key, the pair stores (what tools/ubench.hip's composite measures: 0.689 ms per 1e8 rows on a 3.8 MiB
// table, the floor of §4.2) — OPTIMISTICALLY: a row without partner raises `miss`, and join_probe_dense_kernel
// (launched right behind, a no-op while the flag is clear) redoes the batch with compaction.  A sample of ~16 K rows
// is tested first, so a probe with many misses costs two empty launches, not an attempt; one with a rare miss pays
// for the attempt (0.7 of the compacting kernel's time) once.
constexpr int JA_ILP = 16; // independent table loads in flight per lane
// rows `every` apart (a sample of the batch): a probe with many misses is recognised before the attempt starts
// (Round 6 measured the sample INSIDE the all-hit kernel — its first 64 workgroups test the rows, every wave looks at the flag
//  behind its first trip: one dispatch less, but an attempt that has to be given up then costs every wave a trip, C3 half-hit
//  build + probe 0.65 -> 0.68 ms for ~5 us on the all-hit side; the separate launch stays.)
// (Round 6 also measured the compacting kernel QUEUED BEHIND the attempt of a first probe — a no-op while the flag is clear, its
//  descriptors cleared by the sample launch, its pair count fetched with the verdict: C3 half-hit build + probe 0.651 -> 0.632 ms,
//  but the empty 6104-workgroup launch costs the all-hit side 18 us, 0.626 -> 0.644 ms.  The host decides between the two.)
__global__ void join_probe_dense_sample_kernel(const uint64_t *__restrict__ keys, int64_t n, int64_t every, DenseTable dt,
                                               unsigned int *__restrict__ miss) {
  if (dt.st && !dense_table_from_device(dt)) {
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(miss, 2u);
    return;
  }
  const int64_t r = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * every;
  if (r >= n) return;
  const uint64_t d = keys[r] - dt.kmin;
  // (heads[range] is the NULL build row's slot — a non-NULL probe key never matches it — heads[range + 1] the always-empty padding)
  if (dense_get(dt, d < dt.range ? d : dt.range + 1) == DENSE_EMPTY) atomicOr(miss, 1u);
}
// thread t of the grid takes rows t, t + S, t + 2 S, ... (S = threads of the grid), JA_ILP of them per trip: the shape of
// the composite micro-benchmark (per-wave contiguous chunks with clamped tails measured 9 % slower, 0.755 vs 0.69 ms)
template <bool SC1>
__global__ __launch_bounds__(256) void join_probe_dense_allhit_kernel(const uint64_t *__restrict__ keys, int64_t n, DenseTable dt,
                                                                      uint64_t *__restrict__ left_idx,
                                                                      uint32_t *__restrict__ right_idx,
                                                                      unsigned int *__restrict__ miss) {
  if (*(volatile unsigned int *)miss) return; // (the sample met a row without partner: no attempt)
  const int64_t S = (int64_t)gridDim.x * blockDim.x;
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  bool bad = false;
  for (; i + (JA_ILP - 1) * S < n; i += JA_ILP * S) {
    uint64_t k[JA_ILP];
#pragma unroll
    for (int u = 0; u < JA_ILP; u++) k[u] = __builtin_nontemporal_load(keys + i + u * S);
    uint32_t h[JA_ILP];
#pragma unroll
    for (int u = 0; u < JA_ILP; u++) { // (unconditional: out-of-range keys read heads[range + 1], always empty; SC1: agent-scope loads bypass the L1)
      const uint64_t d = k[u] - dt.kmin;
      const uint32_t *hp = dt.heads + (d < dt.range ? d : dt.range + 1); // (never heads[range]: the NULL build row's slot)
      h[u] = SC1 ? __hip_atomic_load(hp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *hp;
    }
#pragma unroll
    for (int u = 0; u < JA_ILP; u++) {
      bad |= h[u] == DENSE_EMPTY;
      __builtin_nontemporal_store((uint64_t)h[u], &left_idx[i + u * S]);
      __builtin_nontemporal_store((uint32_t)(i + u * S), &right_idx[i + u * S]);
    }
  }
  for (; i < n; i += S) { // the last, partial trip
    const uint64_t d = keys[i] - dt.kmin;
    const uint32_t h = dt.heads[d < dt.range ? d : dt.range + 1];
    bad |= h == DENSE_EMPTY;
    left_idx[i] = h;
    right_idx[i] = (uint32_t)i;
  }
  if (__ballot(bad) && lane_id() == 0) atomicOr(miss, 1u);
}

// The same attempt over the BIT-PACKED table (round 5).  A wave takes 512 CONSECUTIVE rows per trip — VEC2: four 16-byte key
// loads per lane (lane l: rows 2l, 2l + 1 of each 128-row piece), the build rows of a piece stored with one 16-byte and
// the probe rows with one 8-byte store per lane; !VEC2 (a key column that is not 16-byte aligned): eight 8-byte loads,
// 8 + 4-byte stores.  Wave-contiguous chunks beat the strided shape above once the table is packed (tools/ubench2.hip,
// 1e8 x 1e6: 0.520 ms against 0.65-0.76 strided; 4 pieces: 3 are 8 % and 6 are 50 % slower, 16-byte plain stores instead
// of non-temporal ones 12 % slower, LDS-DMA keys 12 % slower, the grid makes no difference from 1024 blocks on).
typedef unsigned long long u64x2_vec __attribute__((ext_vector_type(2)));
constexpr int JAP_ROWS = 512; // rows per wave and trip
#ifndef JAP_DBG
#define JAP_DBG 0
#endif
#ifndef JAP_GRID
#define JAP_GRID 32
#endif
template <bool VEC2>
__global__ __launch_bounds__(256) void join_probe_dense_allhit_packed_kernel(const uint64_t *__restrict__ keys, int64_t n, DenseTable dt,
                                                                             uint64_t *__restrict__ left_idx,
                                                                             uint32_t *__restrict__ right_idx,
                                                                             unsigned int *__restrict__ miss) {
  if (*(volatile unsigned int *)miss) return; // (the sample met a row without partner / the build is not dense: no attempt)
  if (dt.st) dense_table_from_device(dt);     // (true: the sample kernel has checked it)
  const int lane = lane_id();
  const int64_t nchunks = n / JAP_ROWS, gw = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)),
                nw = (int64_t)gridDim.x * 4;
  const uint8_t *__restrict__ tab = dt.packed;
  const uint32_t bits = dt.bits, pmask = dt.pmask, range = (uint32_t)dt.range, pad = range + 1; // (pad: always empty; never `range`, the NULL row's)
  const uint64_t kmin = dt.kmin;
  bool bad = false;
  uint32_t hmax = 0; // (an empty entry is all ones = the largest value an entry takes: one max per entry, one compare at the end)
  for (int64_t c = gw; c < nchunks; c += nw) {
    if (VEC2) {
      const int64_t r0 = c * JAP_ROWS + 2 * lane;
      u64x2_vec k[4];
      uint32_t h[8];
#pragma unroll
      for (int g = 0; g < 4; g++) k[g] = __builtin_nontemporal_load((const u64x2_vec *)(keys + r0 + g * 128));
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const uint64_t d0 = k[g].x - kmin, d1 = k[g].y - kmin;
#if JAP_DBG & 1
        h[2 * g] = dense_packed_raw(tab, bits, pmask, (uint32_t)d0);
        h[2 * g + 1] = dense_packed_raw(tab, bits, pmask, (uint32_t)d1);
#else
        h[2 * g] = dense_packed_raw(tab, bits, pmask, d0 < range ? (uint32_t)d0 : pad);
        h[2 * g + 1] = dense_packed_raw(tab, bits, pmask, d1 < range ? (uint32_t)d1 : pad);
#endif
      }
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int64_t r = r0 + g * 128;
#if !(JAP_DBG & 2)
        hmax = max(hmax, max(h[2 * g], h[2 * g + 1]));
#endif
        u64x2_vec lv;
        lv.x = h[2 * g];
        lv.y = h[2 * g + 1];
        __builtin_nontemporal_store(lv, (u64x2_vec *)(left_idx + r));
        __builtin_nontemporal_store(((uint64_t)(uint32_t)(r + 1) << 32) | (uint32_t)r, (uint64_t *)(right_idx + r));
      }
    } else {
      const int64_t r0 = c * JAP_ROWS + lane;
      uint64_t k[8];
      uint32_t h[8];
#pragma unroll
      for (int g = 0; g < 8; g++) k[g] = __builtin_nontemporal_load(keys + r0 + g * 64);
#pragma unroll
      for (int g = 0; g < 8; g++) {
        const uint64_t d = k[g] - kmin;
        h[g] = dense_packed_raw(tab, bits, pmask, d < range ? (uint32_t)d : pad);
      }
#pragma unroll
      for (int g = 0; g < 8; g++) {
        hmax = max(hmax, h[g]);
        __builtin_nontemporal_store((uint64_t)h[g], left_idx + r0 + g * 64);
        __builtin_nontemporal_store((uint32_t)(r0 + g * 64), right_idx + r0 + g * 64);
      }
    }
  }
  if (gw == nchunks % nw) // the rows behind the last whole chunk: the wave whose turn it would be
    for (int64_t r = nchunks * JAP_ROWS + lane; r < n; r += 64) {
      const uint64_t d = keys[r] - kmin;
      const uint32_t h = dense_packed_raw(tab, bits, pmask, d < range ? (uint32_t)d : pad);
      bad |= h == pmask;
      left_idx[r] = h;
      right_idx[r] = (uint32_t)r;
    }
  bad |= hmax == pmask;
  if (__ballot(bad) && lane == 0) atomicOr(miss, 1u);
}

// `skip_unless` (optional): the optimistic kernel above ran first — while its flag is clear every pair is in place
// and this launch only publishes the total
template <bool HASV>
__global__ __launch_bounds__(JD_BLOCK, JD_OCC) void join_probe_dense_kernel(
    const uint64_t *__restrict__ keys, const uint64_t *__restrict__ validity, int64_t n, int64_t num_tiles,
    DenseTable dt, uint64_t *__restrict__ left_idx, uint32_t *__restrict__ right_idx, uint64_t *desc,
    unsigned *ticket, uint64_t *total, int use_ticket, const unsigned int *__restrict__ skip_unless = nullptr) {
  if (skip_unless && *skip_unless == 0) { // (uniform over the grid: read before any barrier or ticket)
    if (blockIdx.x == 0 && threadIdx.x == 0) *total = (uint64_t)n;
    return;
  }
  unsigned *timeout = use_ticket ? nullptr : ticket + 1;
  __shared__ int64_t s_tile;
  __shared__ uint32_t s_wave[JD_WAVES];
  __shared__ uint64_t s_excl;
  int64_t tile = blockIdx.x;
  if (use_ticket) {
    if (threadIdx.x == 0) s_tile = (int64_t)atomicAdd(ticket, 1u);
    __syncthreads();
    tile = s_tile;
  }
  const int lane = lane_id(), w = wave_id();
  if (w == JD_WAVES) { // ---- scan wave
    __syncthreads(); // (1) the workers' counts are in s_wave
    uint32_t c = lane < JD_WAVES ? s_wave[lane] : 0;
    uint64_t agg = wave_sum_u32(c);
#if JD_DBG & 2 // (measurement: no look-back; wrong offsets)
    uint64_t excl = (uint64_t)tile * (JD_TILE / 2);
#else
    uint64_t excl = lookback_wave(desc, tile, agg, timeout);
#endif
    if (lane == 0) {
      s_excl = excl;
      if (tile == num_tiles - 1) *total = excl + agg;
    }
    __syncthreads(); // (2)
    return;
  }
  // ---- worker waves
  const int64_t wrow = tile * JD_TILE + (int64_t)w * (JD_ITEMS * 64) + lane;
  uint64_t k[JD_ITEMS];
#pragma unroll
  for (int j = 0; j < JD_ITEMS; j++) // streamed once: keep the table cached
    k[j] = __builtin_nontemporal_load(keys + min(wrow + j * 64, n - 1));
  uint32_t head[JD_ITEMS];
#pragma unroll
  for (int j = 0; j < JD_ITEMS; j++) { // independent 4-byte table loads, all in flight together
    const int64_t r = wrow + j * 64;
    const uint64_t d = k[j] - dt.kmin;
    bool isnull = false;
    if (HASV) {
      const int64_t rc = min(r, n - 1);
      isnull = !((validity[rc >> 6] >> (rc & 63)) & 1);
    }
    uint32_t h = DENSE_EMPTY;
#if JD_DBG & 1 // (measurement: no table lookups)
    if (r < n && !isnull && d < dt.range) h = (uint32_t)d;
#else
    if (r < n && !isnull && d < dt.range) h = dense_get(dt, d);
#endif
    if (HASV && r < n && isnull) h = dt.null_head;
    head[j] = h;
  }
  uint64_t mine = 0; // lane j keeps the hit mask of chunk j
  uint32_t wave_cnt = 0;
#pragma unroll
  for (int j = 0; j < JD_ITEMS; j++) {
    uint64_t b = __ballot(head[j] != DENSE_EMPTY);
    mine = (lane == j) ? b : mine;
    wave_cnt += (uint32_t)__popcll(b);
  }
  if (lane == 0) s_wave[w] = wave_cnt;
  __syncthreads(); // (1)
  __syncthreads(); // (2) the scan wave has published the tile's offset
  uint64_t pos = s_excl;
  for (int q = 0; q < w; q++) pos += s_wave[q];
  const uint32_t mlo = (uint32_t)mine, mhi = (uint32_t)(mine >> 32);
#pragma unroll
  for (int j = 0; j < JD_ITEMS; j++) {
    uint64_t m = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)mhi, j) << 32) |
                 (uint32_t)__builtin_amdgcn_readlane((int)mlo, j);
    if ((m >> lane) & 1) {
      uint64_t o = pos + mbcnt(m);
      __builtin_nontemporal_store((uint64_t)head[j], &left_idx[o]);
      __builtin_nontemporal_store((uint32_t)(wrow + j * 64), &right_idx[o]);
    }
    pos += (uint32_t)__popcll(m);
  }
}

// (The dense probe's block shape — 8 worker waves x 16 slot loads per lane + scan wave, 8192-row tiles — was
// tried for the general hash table too and measured SLOWER than join_probe_unique_kernel's 4 waves x 8 loads,
// 3.27 vs 2.56 ms per 1e8 probe rows on a 32 MiB table: the probe is bound by the random-access rate of a
// table beyond one XCD's L2 (65 G 16-byte loads/s = 1.5 ms, profiles/r01_ubench_mi355x.txt) plus its key stream
// and pair stores, and many small blocks keep more of those lookups in flight than few large ones.)
// UNIQUE build keys, Right/Full: every probe row emits exactly one pair (hash_join.rs:235-247)
template <bool DENSE>
__global__ __launch_bounds__(BLOCK) void join_probe_unique_outer_kernel(
    const uint64_t *__restrict__ keys, const uint64_t *__restrict__ validity, int64_t n,
    const Slot *__restrict__ table, uint64_t mask, DenseTable dt, uint64_t *__restrict__ left_idx,
    uint32_t *__restrict__ right_idx, uint64_t *__restrict__ left_validity) {
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  bool hit = false;
  if (r < n) {
    bool is_null = validity && !((validity[r >> 6] >> (r & 63)) & 1);
    uint32_t h;
    if (DENSE) {
      uint64_t d = keys[r] - dt.kmin;
      h = is_null ? dt.null_head : (d < dt.range ? dense_get(dt, d) : DENSE_EMPTY);
      hit = h != DENSE_EMPTY;
    } else {
      Slot s = probe_slot(table, mask, keys[r], is_null);
      hit = s.count != 0;
      h = s.head;
    }
    left_idx[r] = hit ? h : 0;
    right_idx[r] = (uint32_t)r;
  }
  uint64_t mm = __ballot(hit);
  if (lane_id() == 0 && r < n) left_validity[r >> 6] = mm;
}

static uint64_t dense_slots_per_key_owned() {
  const char *e = hook("SQLRS_DENSE_JOIN_SLOTS"); // test / tuning hook, read per call
  return e ? (uint64_t)std::max(1, std::atoi(e)) : 16;
}
// min / max of the valid build keys as signed integers (dense-range detection)
__global__ __launch_bounds__(256) void key_minmax_kernel(const uint64_t *__restrict__ keys, const uint64_t *__restrict__ validity,
                                  int64_t n, unsigned long long *mn, unsigned long long *mx) {
  unsigned long long lo = ~0ull, hi = 0;
  constexpr int KU = 8; // independent loads in flight per lane (rows past the end re-read the last row)
  for (int64_t base = blockIdx.x * (int64_t)(blockDim.x * KU) + threadIdx.x; base < n;
       base += (int64_t)gridDim.x * (blockDim.x * KU)) {
    uint64_t k[KU];
#pragma unroll
    for (int u = 0; u < KU; u++) k[u] = __builtin_nontemporal_load(keys + min(base + (int64_t)u * blockDim.x, n - 1));
#pragma unroll
    for (int u = 0; u < KU; u++) {
      const int64_t r = min(base + (int64_t)u * blockDim.x, n - 1);
      if (validity && !((validity[r >> 6] >> (r & 63)) & 1)) continue;
      unsigned long long o = i64_to_ordered((int64_t)k[u]);
      lo = o < lo ? o : lo;
      hi = o > hi ? o : hi;
    }
  }
  for (int m = 32; m >= 1; m >>= 1) {
    unsigned long long a = shfl_xor_u64(lo, m), b = shfl_xor_u64(hi, m);
    lo = a < lo ? a : lo;
    hi = b > hi ? b : hi;
  }
  __shared__ unsigned long long s_lo[4], s_hi[4]; // 256 threads; one pair of atomics per block
  if (lane_id() == 0) {
    s_lo[wave_id()] = lo;
    s_hi[wave_id()] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) {
      lo = s_lo[w] < lo ? s_lo[w] : lo;
      hi = s_hi[w] > hi ? s_hi[w] : hi;
    }
    atomicMin(mn, lo);
    atomicMax(mx, hi);
  }
}
__global__ void dense_fill_kernel(const uint64_t *__restrict__ keys, const uint64_t *__restrict__ validity,
                                  int64_t n, uint64_t kmin, uint32_t *__restrict__ heads, uint32_t *null_head,
                                  unsigned long long *counts /* [1] += NULL keys */) {
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= n) return;
  if (validity && !((validity[r >> 6] >> (r & 63)) & 1)) {
    *null_head = (uint32_t)r; // unique build keys: at most one NULL row
    atomicAdd(counts + 1, 1ull);
    return;
  }
  heads[keys[r] - kmin] = (uint32_t)r;
}

// After dense_fill_kernel (last writer wins): the valid keys are unique exactly when they occupy as
// many slots as there are valid rows — a streaming count of the table (4 B per possible key)
// instead of a second random access per build row (0.19 -> 0.02 ms for 1e7 keys).  Two NULL keys
// are duplicates too (NULL = NULL matches): the host checks counts[1] <= 1.
__global__ __launch_bounds__(256) void dense_count_kernel(const uint32_t *__restrict__ heads, int64_t range,
                                                          unsigned long long *counts /* [0] += occupied slots */) {
  uint32_t c = 0;
  constexpr int KU = 8;
  for (int64_t base = blockIdx.x * (int64_t)(256 * KU) + threadIdx.x; base < range; base += (int64_t)gridDim.x * (256 * KU)) {
    uint32_t h[KU];
#pragma unroll
    for (int u = 0; u < KU; u++) h[u] = heads[min(base + u * 256, range - 1)];
#pragma unroll
    for (int u = 0; u < KU; u++) c += (base + u * 256 < range) && h[u] != DENSE_EMPTY;
  }
  c = wave_sum_u32(c);
  __shared__ uint32_t s_c[4];
  if (lane_id() == 0) s_c[wave_id()] = c;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(counts, (unsigned long long)(s_c[0] + s_c[1] + s_c[2] + s_c[3]));
}

// ---- the direct-address build without a host round trip in the middle (round 5) -------------------------------------
// The build above fetches the key range, sizes the table from it, fills, counts and fetches the verdict: two stream
// synchronisations and five small device operations for 8 MB of input (0.095 ms for 1e6 keys — 12 % of C3's
// build + probe).  Here the table is allocated for the LARGEST range that would still take the route (slots per key x
// rows + 1024, known without looking at a key), the kernels read the range where key_minmax left it on the device and
// return at once when it is too large, and the ONE fetch at the end carries everything the host decides on:
//   st[0 .. 16) = ~min (ordered image; atomicMax, so that a zeroed block is the neutral start), st[16 .. 32) = max (ordered
//   image) — SIXTEEN words each, block b adds to word b % 16: two atomics per block on ONE pair of words serialise at
//   ~12 ns each (489 blocks for 1e6 keys: 12 of the kernel's 14 us); st[32] = occupied slots, st[33] = NULL keys,
//   st[34] = the NULL row's head.
// dense_pack_count_kernel also writes the bit-packed copy the probe kernels read (DenseTable).
struct DenseDev { // what every kernel of the sequence derives from st[0..1]
  bool ok;
  uint64_t kmin, range;
};
constexpr int DENSE_MM = 16, DENSE_ST_WORDS = 2 * DENSE_MM + 3;
__device__ __forceinline__ DenseDev dense_dev(const unsigned long long *__restrict__ st, uint64_t max_range) {
  uint64_t nlo = 0, hi = 0; // (uniform addresses: scalar loads)
#pragma unroll
  for (int i = 0; i < DENSE_MM; i++) {
    nlo = st[i] > nlo ? st[i] : nlo;
    hi = st[DENSE_MM + i] > hi ? st[DENSE_MM + i] : hi;
  }
  const uint64_t lo = ~nlo;
  DenseDev d;
  d.range = hi - lo + 1;
  d.ok = lo <= hi && d.range <= max_range && d.range < (1ull << 31);
  d.kmin = lo ^ (1ull << 63);
  return d;
}
__device__ __forceinline__ bool dense_table_from_device(DenseTable &dt) {
  const DenseDev d = dense_dev(dt.st, dt.st_max_range);
  const unsigned long long occupied = dt.st[2 * DENSE_MM], nulls = dt.st[2 * DENSE_MM + 1];
  dt.kmin = d.kmin;
  dt.range = d.range;
  return d.ok && nulls <= 1 && occupied + nulls == dt.st_rows; // (what the host decides on the same words, dense_resolve)
}
// `init4` (optional, round 6): the direct-address table of the LARGEST admissible range is set to "empty" by this launch too
// — the build of a small dimension is a chain of launch-bound kernels (7 + 3 + 20 + 11 us of work behind ~5 us of launch
// each), and a table of <= 32 MiB is written faster than a separate launch is issued
__global__ __launch_bounds__(256) void key_minmax_inv_kernel(const uint64_t *__restrict__ keys, const uint64_t *__restrict__ validity,
                                                             int64_t n, unsigned long long *st, uint4 *__restrict__ init4, int64_t init_n4) {
  if (init4) {
    const uint4 e = make_uint4(DENSE_EMPTY, DENSE_EMPTY, DENSE_EMPTY, DENSE_EMPTY);
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < init_n4; i += (int64_t)gridDim.x * 256) init4[i] = e;
  }
  unsigned long long lo = ~0ull, hi = 0;
  constexpr int KU = 8;
  for (int64_t base = blockIdx.x * (int64_t)(256 * KU) + threadIdx.x; base < n; base += (int64_t)gridDim.x * (256 * KU)) {
    uint64_t k[KU];
#pragma unroll
    for (int u = 0; u < KU; u++) k[u] = __builtin_nontemporal_load(keys + min(base + (int64_t)u * 256, n - 1));
#pragma unroll
    for (int u = 0; u < KU; u++) {
      const int64_t r = min(base + (int64_t)u * 256, n - 1);
      if (validity && !((validity[r >> 6] >> (r & 63)) & 1)) continue;
      const unsigned long long o = i64_to_ordered((int64_t)k[u]);
      lo = o < lo ? o : lo;
      hi = o > hi ? o : hi;
    }
  }
  for (int m = 32; m >= 1; m >>= 1) {
    const unsigned long long a = shfl_xor_u64(lo, m), b = shfl_xor_u64(hi, m);
    lo = a < lo ? a : lo;
    hi = b > hi ? b : hi;
  }
  __shared__ unsigned long long s_lo[4], s_hi[4];
  if (lane_id() == 0) {
    s_lo[wave_id()] = lo;
    s_hi[wave_id()] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) {
      lo = s_lo[w] < lo ? s_lo[w] : lo;
      hi = s_hi[w] > hi ? s_hi[w] : hi;
    }
    if (lo <= hi) { // (a block that saw only NULL keys adds nothing)
      atomicMax(st + (blockIdx.x % DENSE_MM), ~lo);
      atomicMax(st + DENSE_MM + (blockIdx.x % DENSE_MM), hi);
    }
  }
}
__global__ __launch_bounds__(256) void dense_init_dev_kernel(const unsigned long long *__restrict__ st, uint64_t max_range,
                                                             uint4 *__restrict__ heads4) {
  const DenseDev d = dense_dev(st, max_range);
  if (!d.ok) return;
  const int64_t n4 = (int64_t)((d.range + 2 + 3) / 4); // (the allocation is rounded up to 16 bytes and more)
  const uint4 e = make_uint4(DENSE_EMPTY, DENSE_EMPTY, DENSE_EMPTY, DENSE_EMPTY);
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) heads4[i] = e;
}
// DENSE_FILL_U build rows per thread (independent key loads / table stores in flight): 4 for a small dimension, whose build is a
// chain of launch-bound kernels (1e6 keys: C3), 1 from 2^21 rows on — the 1e7-row dimension of C5 fills in 0.145 ms with one row
// per thread and 0.217 with four (profiles/r05zzzzz_kernel_stats.csv against r06k: the round-6 change had cost the C5 step 70 us)
template <int DENSE_FILL_U>
__global__ __launch_bounds__(256) void dense_fill_dev_kernel(const uint64_t *__restrict__ keys, const uint64_t *__restrict__ validity,
                                                             int64_t n, const unsigned long long *__restrict__ st, uint64_t max_range,
                                                             uint32_t *__restrict__ heads, unsigned long long *counts /* st + 2 DENSE_MM */) {
  const DenseDev d = dense_dev(st, max_range);
  if (!d.ok) return;
  const int64_t r0 = blockIdx.x * (256ll * DENSE_FILL_U) + threadIdx.x;
  uint64_t k[DENSE_FILL_U];
#pragma unroll
  for (int u = 0; u < DENSE_FILL_U; u++) k[u] = keys[min(r0 + u * 256, n - 1)];
#pragma unroll
  for (int u = 0; u < DENSE_FILL_U; u++) {
    const int64_t r = r0 + u * 256;
    if (r >= n) continue;
    if (validity && !((validity[r >> 6] >> (r & 63)) & 1)) {
      heads[d.range] = (uint32_t)r; // the spare slot behind the table (unique build keys: at most one NULL row)
      atomicAdd(counts + 1, 1ull);
      continue;
    }
    heads[k[u] - d.kmin] = (uint32_t)r;
  }
}
// lane t of the grid owns entries [32 t, 32 t + 32): `bits` whole dwords of the packed table; counts the occupied slots
// of [0, range) on the way (what dense_count_kernel does) and leaves the NULL row's head where the host fetches it
// (Round 6 measured a form that stages a block's 8192 entries and its packed dwords through LDS — coalesced both ways —
//  at 15.9 us against this one's 11.0 for a 1e6-entry range: the range fills 122 blocks, and two barriers per trip cost more
//  there than the strided 16-byte loads.)
__global__ __launch_bounds__(256) void dense_pack_count_kernel(const uint32_t *__restrict__ heads, const unsigned long long *__restrict__ st,
                                                               uint64_t max_range, uint32_t bits, uint32_t *__restrict__ packed,
                                                               unsigned long long *counts /* st + 2 DENSE_MM */) {
  const DenseDev d = dense_dev(st, max_range);
  if (!d.ok) return;
  const int64_t total = (int64_t)d.range + 2, ngroups = (total + 31) / 32;
  const uint32_t pmask = (1u << bits) - 1;
  uint32_t c = 0;
  for (int64_t g = blockIdx.x * 256ll + threadIdx.x; g < ngroups; g += (int64_t)gridDim.x * 256) {
    uint64_t buf = 0;
    uint32_t fill = 0;
    uint32_t *out = packed + g * bits;
#pragma unroll 1
    for (int q = 0; q < 8; q++) {
      const int64_t e0 = g * 32 + q * 4;
      uint32_t v[4];
      if (e0 + 3 < total) {
        const uint4 t = *(const uint4 *)(heads + e0);
        v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w;
      } else {
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = e0 + u < total ? heads[e0 + u] : DENSE_EMPTY;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        c += (e0 + u < (int64_t)d.range) && v[u] != DENSE_EMPTY;
        buf |= (uint64_t)(v[u] == DENSE_EMPTY ? pmask : v[u]) << fill;
        fill += bits;
        if (fill >= 32) {
          *out++ = (uint32_t)buf;
          buf >>= 32;
          fill -= 32;
        }
      }
    }
  }
  c = wave_sum_u32(c);
  __shared__ uint32_t s_c[4];
  if (lane_id() == 0) s_c[wave_id()] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t t = s_c[0] + s_c[1] + s_c[2] + s_c[3];
    if (t) atomicAdd(counts, (unsigned long long)t);
    if (blockIdx.x == 0) counts[2] = heads[d.range];
  }
}
// the same sequence's last step when the table is not packed (more than 25 bits per entry, or 2^32 bits and more)
__global__ __launch_bounds__(256) void dense_count_dev_kernel(const uint32_t *__restrict__ heads, const unsigned long long *__restrict__ st,
                                                              uint64_t max_range, unsigned long long *counts) {
  const DenseDev d = dense_dev(st, max_range);
  if (!d.ok) return;
  const int64_t range = (int64_t)d.range;
  uint32_t c = 0;
  constexpr int KU = 8;
  for (int64_t base = blockIdx.x * (int64_t)(256 * KU) + threadIdx.x; base < range; base += (int64_t)gridDim.x * (256 * KU)) {
    uint32_t h[KU];
#pragma unroll
    for (int u = 0; u < KU; u++) h[u] = heads[min(base + u * 256, range - 1)];
#pragma unroll
    for (int u = 0; u < KU; u++) c += (base + u * 256 < range) && h[u] != DENSE_EMPTY;
  }
  c = wave_sum_u32(c);
  __shared__ uint32_t s_c[4];
  if (lane_id() == 0) s_c[wave_id()] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t t = s_c[0] + s_c[1] + s_c[2] + s_c[3];
    if (t) atomicAdd(counts, (unsigned long long)t);
    if (blockIdx.x == 0) counts[2] = heads[d.range];
  }
}

// ---- key-only build side: existence bitmap ------------------------------------------------------------------
// An Inner join whose build side contributes nothing but its (unique, exactly compared, dense) key column — the
// dimension of a PK-FK join projected to its key, what `HashAgg(HashJoin(dim, fact))` leaves of the dim when the
// aggregates read fact columns only — needs no build row per probe row: the joined batch is the probe batch
// restricted to the rows whose key EXISTS on the build side, with the key column repeated.  One bit per possible
// key (1.25 MB for 1e7 keys: resident in every XCD's L2) replaces the 4-byte head (40 MB: ten times one L2, probed
// at 56-66 G lookups/s — a third of the three-operator C5 step); and when every probe row has a partner, the
// common PK-FK case, the output shares the probe columns outright.
__global__ void dense_bits_kernel(const uint32_t *__restrict__ heads, int64_t range, uint64_t *__restrict__ bits) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const uint64_t w = __ballot(i < range && heads[min(i, range - 1)] != DENSE_EMPTY);
  if (lane_id() == 0 && (i >> 6) < (range + 63) / 64) bits[i >> 6] = w;
}
constexpr int SM_U = 8; // 64-row chunks per wave and trip
__global__ __launch_bounds__(BLOCK) void semi_mask_kernel(const uint64_t *__restrict__ keys, int64_t n,
                                                          const uint64_t *__restrict__ bits, uint64_t kmin, uint64_t range,
                                                          uint64_t *__restrict__ mask, unsigned long long *__restrict__ hits) {
  const int lane = lane_id();
  const int64_t nchunks = (n + 63) / 64;
  uint32_t wave_hits = 0;
  for (int64_t c0 = ((int64_t)blockIdx.x * WAVES_PER_BLOCK + wave_id()) * SM_U; c0 < nchunks;
       c0 += (int64_t)gridDim.x * WAVES_PER_BLOCK * SM_U) {
    uint64_t k[SM_U];
#pragma unroll
    for (int u = 0; u < SM_U; u++) k[u] = __builtin_nontemporal_load(keys + min((c0 + u) * 64 + lane, n - 1));
    uint64_t wd[SM_U];
#pragma unroll
    for (int u = 0; u < SM_U; u++) {
      const uint64_t d = k[u] - kmin;
      wd[u] = d < range ? (bits ? bits[d >> 6] : ~0ull) : 0ull; // independent 8-byte L2 reads, all in flight together (bits == nullptr: every key of the range has a build row)
    }
    uint64_t mine = 0;
#pragma unroll
    for (int u = 0; u < SM_U; u++) {
      const uint64_t d = k[u] - kmin;
      const uint64_t b = __ballot((c0 + u) * 64 + lane < n && ((wd[u] >> (d & 63)) & 1));
      mine = lane == u ? b : mine;
      wave_hits += (uint32_t)__popcll(b);
    }
    if (lane < SM_U && c0 + lane < nchunks) mask[c0 + lane] = mine; // one 64-byte store per trip
  }
  if (lane == 0 && wave_hits) atomicAdd(hits, (unsigned long long)wave_hits); // (one atomic per wave: all rows matched?)
}

__global__ void bytes_to_bits_kernel(const uint8_t *__restrict__ bytes, int64_t n,
                                     uint64_t *__restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  bool b = (i < n) && bytes[i];
  uint64_t m = __ballot(b);
  if (lane_id() == 0 && i < n) out[i >> 6] = m;
}

// sets bit idx[i] for every valid i  (visited_left_side / visited_right_side)
template <class I>
__global__ void mark_bits_kernel(const I *__restrict__ idx, const uint64_t *__restrict__ idx_validity,
                                 int64_t n, unsigned long long *__restrict__ bits) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (idx_validity && !((idx_validity[i >> 6] >> (i & 63)) & 1)) return;
  uint64_t x = (uint64_t)idx[i];
  // bits only ever get set: a plain (possibly stale) read decides whether the atomic is needed at
  // all — after a build row's first match it is not (2e7 pairs on 1e6 build rows: 0.8 ms of
  // atomics on 16 K words, the difference between a Left and an Inner join, -> ~0.05 ms)
  const unsigned long long bit = 1ull << (x & 63);
  if (__hip_atomic_load(&bits[x >> 6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) return;
  atomicOr(&bits[x >> 6], bit);
}

} // namespace sq
