// join.hip — HashJoinExecutor on device (src/executor/join/hash_join.rs:146-323).
//
// Build (left child): integer keys that cover a small range (a dimension table's surrogate keys)
// get a direct-address table heads[key - min] = row, verified to be duplicate-free; anything
// else an open-addressing table in HBM, 16-byte slots {key, head, count}, linear probing, slots
// claimed with one 64-bit CAS on the key.  Unique build keys (the PK-FK case) store the build
// row directly; otherwise rows of one key are laid out CSR-style in insertion order (stable
// radix sort by slot), which is what makes the output pair order equal to the reference's
// Vec<usize> per hash (:172-177).
//
// Probe (right child): unique build keys -> one lookup per probe row, hits compacted with ballots
// and a decoupled look-back in a single pass; duplicates -> count matches per probe row ->
// exclusive scan -> fill.  Either way pairs come out probe-row major / build-insertion minor
// exactly like the reference loop (:225-248).
// Algorithmic HBM bytes: 8 B per build row + 8 B per probe row + 12 B per emitted pair.
// The random-access working set is the table: 4 B x key range (direct-address, 4 MiB for 1e6
// keys = one XCD's L2, ~265 G lookups/s) or 16 B x 2 x build rows (hash table, ~56-66 G
// lookups/s once it exceeds the L2; profiles/r01_ubench_mi355x.txt).
#include <cstdlib>

#include "common.hpp"
#include "device_utils.hpp"
#include "prims.hpp"
#include "small_async.hpp"
#include "join_kernels.hpp"
#include "join_state.hpp"

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

using namespace sq;

#include "join_state.hpp"
#include "radix_part.hpp"

extern "C" void sqlrs_batch_release(sqlrs_batch_t *batch);

namespace sq {

struct Pairs {
  int64_t m = 0;
  BufP left, right;   // u64[m], u32[m]
  BufP left_validity; // bitmap or null (Right/Full only)
  // pair i = (some build row, probe row i) for EVERY probe row (the all-hit attempt succeeded): the probe side of the
  // joined batch is the probe batch itself, nothing to gather
  bool right_identity = false;
};

static std::vector<DCol> eval_key_cols(Ctx *ctx, const std::vector<Expr> &exprs, const std::function<const DCol &(int)> &col,
                                       int64_t rows) {
  std::vector<DCol> kc;
  for (const Expr &e : exprs) kc.push_back(eval_expr(ctx, e, col, rows, true));
  return kc;
}
static NKeys eval_keys(Ctx *ctx, const std::vector<Expr> &exprs,
                       const std::function<const DCol &(int)> &col, int64_t rows) {
  return normalize_keys(ctx, eval_key_cols(ctx, exprs, col, rows), rows);
}

// ---- several integer key columns as ONE exact key (sqlrs_hash_join::Composite) ----------------------------------------
// The reference hashes the key columns into 64 bits and matches by hash alone (hash_join.rs:161-232) — and its fold of the
// column hashes collides readily: on a 1000 x 1000 grid of (x, y) build keys, 3e6 probe rows find 3.71e6 partners where
// 2.96e6 exist.  The default of this library reproduces exactly that (normalize_keys, hash mode: same pairs, same order as the
// reference, false matches included).  This is the OPT-IN alternative for callers who want the SQL answer: an exact
// composite key, which also makes dense key grids eligible for the direct-address table and sparse ones for the LDS route.
// NULL in a key column never matches here; the composite is taken only when no build key is NULL, and a probe row with a NULL
// key, a value outside the build side's ranges, or a key column of another integer type gets a key that matches nothing.
struct CompJoinKeys {
  const void *vals[4];
  const uint64_t *valid[4];
  int64_t min[4];
  uint64_t range[4], stride[4];
  int is32[4];
  int nk; // < 0: every row gets the no-match key (key column types differ between the sides)
};
constexpr uint64_t COMP_NO_MATCH = ~0ull; // (composites are < 2^62)
__device__ __forceinline__ int64_t comp_value(const CompJoinKeys &ck, int c, int64_t i) {
  return ck.is32[c] ? (int64_t)((const int32_t *)ck.vals[c])[i] : ((const int64_t *)ck.vals[c])[i];
}
__global__ __launch_bounds__(256) void comp_join_minmax_kernel(const void *__restrict__ vals, int is32, int64_t n, long long *mm) {
  long long lo = INT64_MAX, hi = INT64_MIN;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const long long v = is32 ? (long long)((const int32_t *)vals)[i] : ((const long long *)vals)[i];
    lo = min(lo, v);
    hi = max(hi, v);
  }
  for (int m = 32; m >= 1; m >>= 1) {
    lo = min(lo, (long long)shfl_xor_u64((uint64_t)lo, m));
    hi = max(hi, (long long)shfl_xor_u64((uint64_t)hi, m));
  }
  if (lane_id() == 0) {
    atomicMin(mm, lo);
    atomicMax(mm + 1, hi);
  }
}
__global__ __launch_bounds__(256) void comp_join_keys_kernel(CompJoinKeys ck, int64_t n, uint64_t *__restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t key = 0;
  bool ok = ck.nk > 0;
  for (int c = 0; c < ck.nk && ok; c++) {
    if (ck.valid[c] && !((ck.valid[c][i >> 6] >> (i & 63)) & 1ull)) ok = false;
    else {
      const uint64_t d = (uint64_t)comp_value(ck, c, i) - (uint64_t)ck.min[c]; // (wraps to a huge value below the minimum)
      if (d >= ck.range[c]) ok = false;
      else key += d * ck.stride[c];
    }
  }
  out[i] = ok ? key : COMP_NO_MATCH;
}
// build side: ranges of the key columns over all build batches -> j->comp, and the composite key of every build row
static BufP composite_build_keys(sqlrs_hash_join *j) {
  Ctx *ctx = j->ctx;
  const size_t nk = j->lkeys.size();
  // OPT-IN (SQLRS_JOIN_COMPOSITE=1, read per build): exact equality is NOT what the reference computes — see the note above
  const char *e = std::getenv("SQLRS_JOIN_COMPOSITE");
  if (!e || e[0] != '1') return nullptr;
  if (nk < 2 || nk > 4 || j->lazy_table || j->nB < 1 || j->left_keycol_parts.size() != j->left_batches.size()) return nullptr;
  for (const std::vector<DCol> &part : j->left_keycol_parts) {
    if (part.size() != nk) return nullptr;
    for (size_t c = 0; c < nk; c++) {
      const DCol &k = part[c];
      if ((k.dtype != SQLRS_INT64 && k.dtype != SQLRS_INT32) || k.dtype != j->left_keycol_parts[0][c].dtype || k.stride == 0 ||
          (k.validity && count_nulls(ctx, k) != 0))
        return nullptr;
    }
  }
  BufP mm = ctx->alloc(16 * nk);
  std::vector<long long> init;
  for (size_t c = 0; c < nk; c++) {
    init.push_back(INT64_MAX);
    init.push_back(INT64_MIN);
  }
  SQ_HIP(hipMemcpyAsync(mm->p, init.data(), 16 * nk, hipMemcpyHostToDevice, ctx->stream));
  SQ_HIP(hipStreamSynchronize(ctx->stream)); // (`init` is pageable host memory)
  for (const std::vector<DCol> &part : j->left_keycol_parts)
    for (size_t c = 0; c < nk; c++) {
      const int64_t rows = part[c].length;
      if (rows == 0) continue;
      const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(rows, 256), 4 * (int64_t)ctx->num_cus);
      comp_join_minmax_kernel<<<dim3(blocks), dim3(256), 0, ctx->stream>>>(part[c].values, part[c].dtype == SQLRS_INT32, rows,
                                                                         mm->as<long long>() + 2 * c);
    }
  SQ_HIP(hipGetLastError());
  const long long *h = (const long long *)ctx->fetch(mm->p, 16 * nk);
  sqlrs_hash_join::Composite cp;
  cp.nk = (int)nk;
  unsigned __int128 total = 1;
  for (size_t c = 0; c < nk; c++) {
    if (h[2 * c] > h[2 * c + 1]) return nullptr;
    cp.dtype[c] = j->left_keycol_parts[0][c].dtype;
    cp.min[c] = h[2 * c];
    cp.range[c] = (uint64_t)h[2 * c + 1] - (uint64_t)h[2 * c] + 1; // (0 = all 2^64 values: caught by the product below)
    if (cp.range[c] == 0) return nullptr;
    total *= cp.range[c];
    if (total >= ((unsigned __int128)1 << 62)) return nullptr; // the composite does not fit: hashes
  }
  uint64_t stride = 1;
  for (size_t c = nk; c-- > 0;) {
    cp.stride[c] = stride;
    stride *= cp.range[c];
  }
  cp.on = true;
  j->comp = cp;
  BufP keys = ctx->alloc(8 * (size_t)std::max<int64_t>(j->nB, 1));
  int64_t off = 0;
  ProfScope ps(ctx, "normalize_keys");
  for (const std::vector<DCol> &part : j->left_keycol_parts) {
    const int64_t rows = part[0].length;
    if (rows == 0) continue;
    CompJoinKeys ck{};
    ck.nk = (int)nk;
    for (size_t c = 0; c < nk; c++) {
      ck.vals[c] = part[c].values;
      ck.valid[c] = nullptr;
      ck.min[c] = cp.min[c];
      ck.range[c] = cp.range[c];
      ck.stride[c] = cp.stride[c];
      ck.is32[c] = cp.dtype[c] == SQLRS_INT32;
    }
    comp_join_keys_kernel<<<dim3((unsigned)ceil_div(rows, 256)), dim3(256), 0, ctx->stream>>>(ck, rows, keys->as<uint64_t>() + off);
    off += rows;
  }
  SQ_HIP(hipGetLastError());
  return keys;
}
// probe side: the same composite, the no-match key for rows that cannot have a partner
static NKeys composite_probe_keys(sqlrs_hash_join *j, const std::function<const DCol &(int)> &col, int64_t rows) {
  Ctx *ctx = j->ctx;
  std::vector<DCol> kc = eval_key_cols(ctx, j->rkeys, col, rows);
  NKeys k;
  k.rows = rows;
  k.exact = true;
  k.dtype = SQLRS_INT64;
  k.keys = ctx->alloc(8 * (size_t)std::max<int64_t>(rows, 1));
  if (rows == 0) return k;
  const sqlrs_hash_join::Composite &cp = j->comp;
  CompJoinKeys ck{};
  ck.nk = cp.nk;
  for (int c = 0; c < cp.nk; c++) {
    const DCol &d = kc[(size_t)c];
    if (d.dtype != cp.dtype[c]) ck.nk = -1; // (the reference's hashes differ by type: nothing matches)
    ck.vals[c] = d.values;
    ck.valid[c] = (d.validity && d.null_count != 0) ? d.validity : nullptr;
    ck.min[c] = cp.min[c];
    ck.range[c] = cp.range[c];
    ck.stride[c] = cp.stride[c];
    ck.is32[c] = cp.dtype[c] == SQLRS_INT32;
  }
  ProfScope ps(ctx, "normalize_keys");
  comp_join_keys_kernel<<<dim3((unsigned)ceil_div(rows, 256)), dim3(256), 0, ctx->stream>>>(ck, rows, k.keys->as<uint64_t>());
  SQ_HIP(hipGetLastError());
  return k;
}

static void build_hash_table(sqlrs_hash_join *j);
static bool build_dense_dup(sqlrs_hash_join *j);
// The direct-address build's verdict (dense_pack_count_kernel: st[0 .. 35)) -> the join's state; what is not a unique dense
// key set goes on to the general table.  `h`: the words when the caller has fetched them already (with its own answer, in
// one round trip), else they are fetched here.
static void dense_resolve(sqlrs_hash_join *j, const uint64_t *h = nullptr) {
  if (!j->dense_pending) return;
  Ctx *ctx = j->ctx;
  j->dense_pending = false;
  const int64_t n = j->nB;
  uint64_t hw[DENSE_ST_WORDS];
  std::memcpy(hw, h ? h : (const uint64_t *)ctx->fetch(j->pend_st->p, 8 * DENSE_ST_WORDS), sizeof(hw)); // (later fetches reuse the staging buffer)
  BufP dense = std::move(j->pend_dense), packed = std::move(j->pend_packed);
  j->pend_st = nullptr;
  uint64_t nlo = 0, hi = 0;
  for (int i = 0; i < DENSE_MM; i++) nlo = std::max(nlo, hw[i]), hi = std::max(hi, hw[DENSE_MM + i]);
  const uint64_t lo = ~nlo, occupied = hw[2 * DENSE_MM], nulls = hw[2 * DENSE_MM + 1];
  const uint32_t null_head = (uint32_t)hw[2 * DENSE_MM + 2];
  const uint64_t range = hi - lo + 1;
  if (lo <= hi && range <= j->pend_max_range && range < (1ull << 31)) { // (what dense_dev decided)
    const uint64_t dmin = lo ^ (1ull << 63);
    if (nulls <= 1 && occupied + nulls == (uint64_t)n) {
      j->unique = true;
      j->unique_known = j->table_built = true;
      j->dense = dense;
      j->dense_min = dmin;
      j->dense_range = range;
      j->dense_null_head = null_head;
      j->dense_packed = packed;
      j->dense_pbits = j->pend_bits;
      return;
    }
    // fewer occupied slots than valid keys (or several NULL keys, which match each other): the build keys are NOT
    // unique — a fact the fused join+aggregate need not discover again by inserting them into its bucket tables
    j->unique = false;
    j->unique_known = true;
    if (nulls == 0 && !j->pend_validity) { // (no NULL key: the fused join+aggregate can take multiplicities per key, hash_join_dup_mult)
      j->dup_min = dmin;
      j->dup_range = range;
    }
  }
  if (j->lazy_table) return; // built by hash_join_ensure_table when something probes it
  if (build_dense_dup(j)) return; // duplicate keys over a dense range: runs by key, no general table
  if (lds_build_first(j)) return; // general keys on LDS tables: uniqueness from there, the global table on first need only
  build_hash_table(j);
}
static void build_table(sqlrs_hash_join *j) {
  Ctx *ctx = j->ctx;
  // concat key parts
  int64_t n = j->nB;
  // one build batch (the usual case): its normalised keys ARE the key array, no concat copy
  BufP ckeys = composite_build_keys(j); // several integer key columns: one exact key (or null: as normalised per batch)
  j->left_keycol_parts.clear();
  const bool single = !ckeys && j->left_key_parts.size() == 1 && j->left_key_parts[0].keys && j->left_key_parts[0].keys->owned;
  BufP keys = ckeys ? ckeys : (single ? j->left_key_parts[0].keys : ctx->alloc(8 * (size_t)std::max<int64_t>(n, 1)));
  BufP validity;
  bool any_null = false;
  for (const NKeys &p : j->left_key_parts) any_null |= (p.validity != nullptr);
  j->exact = ckeys ? true : j->left_key_parts[0].exact;
  j->key_dtype = ckeys ? SQLRS_INT64 : j->left_key_parts[0].dtype;
  if (!ckeys) {
    size_t off = 0;
    std::vector<DCol> vparts;
    for (const NKeys &p : j->left_key_parts) {
      if (p.exact != j->exact || p.dtype != j->key_dtype)
        fail(SQLRS_ERR_ARROW, "join key type changed between build batches");
      if (p.rows && !single)
        SQ_HIP(hipMemcpyAsync(keys->as<uint8_t>() + off, p.keys->p, 8 * (size_t)p.rows,
                              hipMemcpyDeviceToDevice, ctx->stream));
      off += 8 * (size_t)p.rows;
    }
    if (any_null) { // reuse the bitmap concat of concat_columns through BOOLEAN pseudo columns
      std::vector<DCol> tmp(j->left_key_parts.size());
      std::vector<const DCol *> ptrs;
      for (size_t i = 0; i < tmp.size(); i++) {
        const NKeys &p = j->left_key_parts[i];
        tmp[i].dtype = SQLRS_INT64;
        tmp[i].length = p.rows;
        tmp[i].values = p.keys->p;
        tmp[i].validity = p.validity;
        tmp[i].null_count = p.validity ? -1 : 0;
        ptrs.push_back(&tmp[i]);
      }
      DCol c = concat_columns(ctx, ptrs);
      if (tmp.size() == 1) {
        validity = j->left_key_parts[0].own_validity;
        if (!validity) { // borrowed bitmap: copy it
          validity = ctx->alloc(bitmap_bytes(n));
          SQ_HIP(hipMemcpyAsync(validity->p, tmp[0].validity, bitmap_bytes(n),
                                hipMemcpyDeviceToDevice, ctx->stream));
        }
      } else
        validity = c.own_validity;
    }
  }
  j->bkeys = keys; // kept for the fused join+aggregate route (hashagg_op.hip)
  j->bkeys_validity = validity;
  // 1. dense surrogate keys (range <= 4 x rows — 16 x for a join+aggregate's join — and < 2^31) -> direct-address table.  It is tried
  //    first: when the build keys turn out unique nothing else is needed, and the 16-byte-slot
  //    hash table (1.1 ms for 1e7 keys) is never built.
  const char *db1_e = hook("SQLRS_DENSE_BUILD_ONE_FETCH"); // A/B hook, read per call (0 = the two-fetch sequence below)
  // (the one-fetch form allocates the table of the LARGEST admissible range before it has looked at a key: only while that
  //  stays under 1 GiB — advisor r05; beyond, the two-fetch sequence below sizes the table from the range it has seen)
  const uint64_t spk1 = j->lazy_table ? dense_slots_per_key_owned() : 4;
  if (j->exact && n > 0 && n <= (1ll << 24) && 4 * (spk1 * (uint64_t)n + 1024) <= (1ull << 30) && j->key_dtype != SQLRS_FLOAT64 &&
      !(db1_e && std::atoi(db1_e) == 0)) {
    // one fetch (see dense_pack_count_kernel): the table is sized for the largest range that takes the route
    ProfScope ps(ctx, "join_build_dense");
    const char *pj_e = hook("SQLRS_DENSE_JOIN_SLOTS_PLAIN"); // tuning hook, read per call
    const uint64_t slots_per_key = j->lazy_table ? dense_slots_per_key_owned() : (pj_e ? (uint64_t)std::max(1, std::atoi(pj_e)) : 4);
    const uint64_t max_range = slots_per_key * (uint64_t)n + 1024;
    uint32_t bits = 1;
    while (((1ull << bits) - 1) < (uint64_t)n) bits++; // all ones = empty must not be a build row
    const char *pk_e = hook("SQLRS_DENSE_PACKED"); // A/B hook, read per call (0 = the probe reads the 4-byte table)
    if (bits < 8) bits = 8;
    if (bits > 25 || (max_range + 2) * bits >= (1ull << 32) || j->lazy_table || (pk_e && std::atoi(pk_e) == 0)) bits = 0;
    BufP st = ctx->alloc_zero(8 * (DENSE_ST_WORDS + 1)); // (+ the miss flag of a first probe that runs before the verdict is fetched)
    BufP dense = ctx->alloc(4 * (size_t)max_range + 64);
    BufP packed = bits ? ctx->alloc(4 * (size_t)((max_range + 2 + 31) / 32) * bits + 16) : nullptr;
    const uint64_t *vp = validity ? validity->as<uint64_t>() : nullptr;
    unsigned long long *stp = st->as<unsigned long long>();
    const unsigned mblocks = (unsigned)std::min<int64_t>(ceil_div(n, 256 * 8), 4 * (int64_t)ctx->num_cus);
    const bool init_fused = max_range + 2 <= (8ull << 20); // (entries: <= 32 MiB of table)
    key_minmax_inv_kernel<<<dim3(mblocks), dim3(256), 0, ctx->stream>>>(keys->as<uint64_t>(), vp, n, stp, init_fused ? dense->as<uint4>() : nullptr,
                                                                        (int64_t)((max_range + 2 + 3) / 4));
    if (!init_fused) {
      const unsigned iblocks = (unsigned)std::min<int64_t>(ceil_div((int64_t)max_range + 2, 256 * 4 * 4), 4 * (int64_t)ctx->num_cus);
      dense_init_dev_kernel<<<dim3(iblocks), dim3(256), 0, ctx->stream>>>(stp, max_range, dense->as<uint4>());
    }
    if (n < (1ll << 21))
      dense_fill_dev_kernel<4><<<dim3((unsigned)ceil_div(n, 256 * 4)), dim3(256), 0, ctx->stream>>>(keys->as<uint64_t>(), vp, n, stp, max_range,
                                                                                                  dense->as<uint32_t>(), stp + 2 * DENSE_MM);
    else
      dense_fill_dev_kernel<1><<<dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream>>>(keys->as<uint64_t>(), vp, n, stp, max_range,
                                                                                              dense->as<uint32_t>(), stp + 2 * DENSE_MM);
    if (bits) {
      const unsigned pblocks = (unsigned)std::min<int64_t>(ceil_div(ceil_div((int64_t)max_range + 2, 32), 256), 8 * (int64_t)ctx->num_cus);
      dense_pack_count_kernel<<<dim3(pblocks), dim3(256), 0, ctx->stream>>>(dense->as<uint32_t>(), stp, max_range, bits,
                                                                          packed->as<uint32_t>(), stp + 2 * DENSE_MM);
    } else {
      const unsigned cblocks = (unsigned)std::min<int64_t>(ceil_div((int64_t)max_range, 256 * 8), 4 * (int64_t)ctx->num_cus);
      dense_count_dev_kernel<<<dim3(cblocks), dim3(256), 0, ctx->stream>>>(dense->as<uint32_t>(), stp, max_range, stp + 2 * DENSE_MM);
    }
    SQ_HIP(hipGetLastError());
    j->dense_pending = true;
    j->pend_st = st;
    j->pend_dense = dense;
    j->pend_packed = packed;
    j->pend_bits = bits;
    j->pend_max_range = max_range;
    j->pend_validity = validity != nullptr;
    // The verdict stays on the device when the first probe can take it from there (dense_resolve): a plain join whose probe
    // kernels read the packed table.  SQLRS_DENSE_BUILD_DEFER=0 (read per call): decide here.
    const char *df_e = hook("SQLRS_DENSE_BUILD_DEFER");
    if (bits && !j->lazy_table && !(df_e && std::atoi(df_e) == 0)) return;
    dense_resolve(j);
    return;
  } else if (j->exact && n > 0 && j->key_dtype != SQLRS_FLOAT64) {
    BufP mm = ctx->alloc(16); // {min = ~0, max = 0} without a host round trip
    SQ_HIP(hipMemsetAsync(mm->p, 0xff, 8, ctx->stream));
    SQ_HIP(hipMemsetAsync(mm->as<uint8_t>() + 8, 0, 8, ctx->stream));
    const uint64_t *vp = validity ? validity->as<uint64_t>() : nullptr;
    unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(n, 256 * 8), 4 * (int64_t)ctx->num_cus);
    key_minmax_kernel<<<dim3(blocks), dim3(256), 0, ctx->stream>>>(keys->as<uint64_t>(), vp, n,
                                                                  mm->as<unsigned long long>(),
                                                                  mm->as<unsigned long long>() + 1);
    SQ_HIP(hipGetLastError());
    const uint64_t *h = (const uint64_t *)ctx->fetch(mm->p, 16);
    uint64_t lo = h[0], hi = h[1];
    if (lo <= hi) {
      uint64_t range = hi - lo + 1; // ordered images differ like the signed values
      // (a join owned by a HashJoin+HashAgg takes the table up to 16 slots per key: its fused route then partitions
      //  by key range and needs only the existence bitmap of the range — a filtered dimension, or the hash-partitioned
      //  shard of one that a rank of the multi-GPU plan receives, 1/8 of the keys of the range for 8 ranks)
      const char *pj_e = hook("SQLRS_DENSE_JOIN_SLOTS_PLAIN"); // tuning hook, read per call
      const uint64_t slots_per_key = j->lazy_table ? dense_slots_per_key_owned() : (pj_e ? (uint64_t)std::max(1, std::atoi(pj_e)) : 4);
      if (range <= slots_per_key * (uint64_t)n + 1024 && range < (1ull << 31)) {
        ProfScope ps(ctx, "join_build_dense");
        BufP dense = ctx->alloc(4 * (size_t)range + 8);
        SQ_HIP(hipMemsetAsync(dense->p, 0xff, 4 * (size_t)range + 8, ctx->stream));
        uint64_t dmin = lo ^ (1ull << 63); // back from the ordered image to the two's complement bits
        uint32_t *null_head = dense->as<uint32_t>() + range; // spare slot after the table
        BufP cnt = ctx->alloc_zero(24); // {occupied slots, NULL keys, copy of the NULL row's head (u32)}
        dense_fill_kernel<<<dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream>>>(
            keys->as<uint64_t>(), vp, n, dmin, dense->as<uint32_t>(), null_head, cnt->as<unsigned long long>());
        unsigned cblocks = (unsigned)std::min<int64_t>(ceil_div((int64_t)range, 256 * 8), 4 * (int64_t)ctx->num_cus);
        dense_count_kernel<<<dim3(cblocks), dim3(256), 0, ctx->stream>>>(dense->as<uint32_t>(), (int64_t)range,
                                                                        cnt->as<unsigned long long>());
        SQ_HIP(hipMemcpyAsync(cnt->as<uint64_t>() + 2, null_head, 4, hipMemcpyDeviceToDevice, ctx->stream));
        SQ_HIP(hipGetLastError());
        const uint64_t *hc = (const uint64_t *)ctx->fetch(cnt->p, 24); // one round trip for all three
        const uint32_t hd[2] = {(hc[1] <= 1 && hc[0] + hc[1] == (uint64_t)n) ? 0u : 1u, (uint32_t)hc[2]};
        if (hd[0] == 0) {
          j->unique = true;
          j->unique_known = j->table_built = true;
          j->dense = dense;
          j->dense_min = dmin;
          j->dense_range = range;
          j->dense_null_head = hd[1];
          return;
        }
        // fewer occupied slots than valid keys (or several NULL keys, which match each other): the build keys are NOT
        // unique — a fact the fused join+aggregate need not discover again by inserting them into its bucket tables
        j->unique = false;
        j->unique_known = true;
        if (hc[1] == 0 && !validity) { // (no NULL key: the fused join+aggregate can take multiplicities per key, hash_join_dup_mult)
          j->dup_min = dmin;
          j->dup_range = range;
        }
      }
    }
  }
  if (j->lazy_table) return; // built by hash_join_ensure_table when something probes it
  if (build_dense_dup(j)) return; // duplicate keys over a dense range: runs by key, no general table
  if (lds_build_first(j)) return; // general keys on LDS tables: uniqueness from there, the global table on first need only
  build_hash_table(j);
}

// 1b. (round 6) DUPLICATE build keys over a dense range — a foreign key joined to a foreign key, a dimension attribute: the direct-
// address attempt above has found the range and that some key repeats.  The general table (32 MiB of slots for 1e6 rows, a CAS
// insert per row, a stable radix sort by SLOT for the runs, then — for the LDS route — its distinct keys partitioned again: 0.56 ms
// of small launches and six host round trips for 1e6 rows) is not needed: the rows counted per key of the range
// (hash_join_dup_mult), an exclusive scan, and the rows stably sorted by (key - min) ARE the runs; the probe's count pass reads
// {run start, rows} with one 8-byte load per probe row from a table the size of the range (L2-resident for C3's shapes) instead
// of three passes over LDS tables.  1e8 x 1e6 rows, every key ~4 times (4e8 pairs), build + probe: 3.76 -> 2.57 ms.
// the run starts with three entries behind the range: start[range] = rows (the end of the last run), start[range + 1] =
// start[range + 2] = rows — the empty run every key without partner reads (unconditional loads in dd_count_kernel)
__global__ void dd_table_kernel(const uint32_t *__restrict__ start, int64_t range, uint32_t rows, uint32_t *__restrict__ t) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < range) t[i] = start[i];
  else if (i < range + 3) t[i] = rows;
}
__global__ void dd_sort_keys_kernel(const uint64_t *__restrict__ keys, int64_t n, uint64_t kmin, uint64_t *__restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = keys[i] - kmin;
}
// the count pass over that table: match[r] = {run start, rows}, pair counts per row or per 64-row group (join_count_kernel's outputs)
#ifndef DD_DBG
#define DD_DBG 0
#endif
#ifndef DD_U_N
#define DD_U_N 8
#endif
constexpr int DD_U = DD_U_N;
struct __attribute__((aligned(4))) uint2_unaligned { uint32_t x, y; }; // (two adjacent 4-byte entries, one 8-byte load)
// 64-row groups per wave: their keys, then their table entries, in flight together
__global__ __launch_bounds__(BLOCK) void dd_count_kernel(const uint64_t *__restrict__ keys, const uint64_t *__restrict__ validity, int64_t n,
                                                         uint64_t kmin, uint64_t range, const uint32_t *__restrict__ table, int outer_right,
                                                         uint32_t *__restrict__ counts, uint2 *__restrict__ match, int grouped) {
  const int lane = lane_id();
  const int64_t wbase = (blockIdx.x * (int64_t)WAVES_PER_BLOCK + wave_id()) * (64 * DD_U);
  if (wbase >= n) return;
  uint64_t k[DD_U];
#pragma unroll
  for (int u = 0; u < DD_U; u++) k[u] = __builtin_nontemporal_load(keys + min(wbase + u * 64 + lane, n - 1));
  unsigned long long m[DD_U];
#pragma unroll
  for (int u = 0; u < DD_U; u++) {
    const int64_t r = min(wbase + u * 64 + lane, n - 1);
    const bool is_null = validity && !((validity[r >> 6] >> (r & 63)) & 1); // (no NULL build key on this route: a NULL probe key has no partner)
    const uint64_t d = k[u] - kmin;
#if DD_DBG & 1 // (measurement: no table lookup)
    m[u] = (!is_null && d < range) ? (d | (4ull << 32)) : 0ull;
#else
    // UNCONDITIONAL: a key without partner reads the empty entry behind the range — a load under a condition is a branch, and the
    // eight lookups of a lane then wait for one another (0.73 ms per 1e8 rows; without any lookup 0.27)
    // (4-byte entries, a run's length = the next start - its own: half the table of {start, rows} pairs, 1 MiB for 2.5e5 keys)
    const uint2_unaligned se = *(const uint2_unaligned *)(table + ((!is_null && d < range) ? d : range + 1)); // {start, next start}
    m[u] = (unsigned long long)se.x | ((unsigned long long)(se.y - se.x) << 32); // {run start | rows << 32}
#endif
  }
#pragma unroll
  for (int u = 0; u < DD_U; u++) {
    const int64_t r = wbase + u * 64 + lane;
    uint32_t c = 0;
    if (r < n) {
      c = (uint32_t)(m[u] >> 32);
#if !(DD_DBG & 2) // (measurement: no match store)
      __builtin_nontemporal_store(m[u], (unsigned long long *)(match + r));
#endif
      if (outer_right && c == 0) c = 1;
      if (!grouped) counts[r] = c;
    }
    if (grouped) {
      const uint32_t wsum = wave_sum_u32(c);
      if (lane == 0 && wbase + u * 64 < n) counts[(wbase + u * 64) >> 6] = wsum;
    }
  }
}
// The same pass in the shape of join_probe_dense_allhit_packed_kernel (persistent waves, 512 CONSECUTIVE rows per wave and trip,
// 16-byte key loads and match stores): probe keys without NULLs in a 16-byte aligned column.  The lookups and the two streams
// share the CU's vector memory path; this shape is what got the all-hit probe from 0.69 to 0.52 ms per 1e8 rows (tools/ubench2.hip).
__global__ __launch_bounds__(256) void dd_count_stream_kernel(const uint64_t *__restrict__ keys, int64_t n, uint64_t kmin, uint64_t range,
                                                              const uint32_t *__restrict__ table, int outer_right,
                                                              uint32_t *__restrict__ counts, uint2 *__restrict__ match, int grouped) {
  const int lane = lane_id();
  const int64_t nchunks = n / JAP_ROWS, gw = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)),
                nw = (int64_t)gridDim.x * 4;
  const uint32_t miss = (uint32_t)range + 1;
  for (int64_t c = gw; c < nchunks; c += nw) {
    const int64_t r0 = c * JAP_ROWS + 2 * lane;
    u64x2_vec k[4];
    uint2_unaligned se[8];
#pragma unroll
    for (int g = 0; g < 4; g++) k[g] = __builtin_nontemporal_load((const u64x2_vec *)(keys + r0 + g * 128));
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const uint64_t d0 = k[g].x - kmin, d1 = k[g].y - kmin;
      se[2 * g] = *(const uint2_unaligned *)(table + (d0 < range ? (uint32_t)d0 : miss));
      se[2 * g + 1] = *(const uint2_unaligned *)(table + (d1 < range ? (uint32_t)d1 : miss));
    }
#pragma unroll
    for (int g = 0; g < 4; g++) { // rows r0 + 128 g, + 1: lanes 2 l, 2 l + 1 of the 128-row piece
      const uint32_t c0 = se[2 * g].y - se[2 * g].x, c1 = se[2 * g + 1].y - se[2 * g + 1].x;
      u64x2_vec mv;
      mv.x = (unsigned long long)se[2 * g].x | ((unsigned long long)c0 << 32);
      mv.y = (unsigned long long)se[2 * g + 1].x | ((unsigned long long)c1 << 32);
      __builtin_nontemporal_store(mv, (u64x2_vec *)(match + r0 + g * 128));
      const uint32_t e0 = (outer_right && c0 == 0) ? 1u : c0, e1 = (outer_right && c1 == 0) ? 1u : c1;
      if (grouped) { // two 64-row groups per piece: lanes 0-31 hold the first, 32-63 the second
        uint32_t v = e0 + e1;
        for (int m = 16; m >= 1; m >>= 1) v += (uint32_t)__shfl_xor((int)v, m, 64);
        if ((lane & 31) == 0) counts[((c * JAP_ROWS + g * 128) >> 6) + (lane >> 5)] = v;
      } else {
        *(uint2 *)(counts + r0 + g * 128) = make_uint2(e0, e1);
      }
    }
  }
  if (gw == nchunks % nw) // the rows behind the last whole chunk (< 512, whole 64-row groups first): the wave whose turn it would be
    for (int64_t rb = nchunks * JAP_ROWS; rb < n; rb += 64) {
      const int64_t r = rb + lane;
      uint32_t cv = 0;
      if (r < n) {
        const uint64_t d = keys[r] - kmin;
        const uint2_unaligned e = *(const uint2_unaligned *)(table + (d < range ? (uint32_t)d : miss));
        const uint32_t c0 = e.y - e.x;
        match[r] = make_uint2(e.x, c0);
        cv = (outer_right && c0 == 0) ? 1u : c0;
        if (!grouped) counts[r] = cv;
      }
      if (grouped) {
        const uint32_t wsum = wave_sum_u32(cv);
        if (lane == 0) counts[rb >> 6] = wsum;
      }
    }
}
static bool build_dense_dup(sqlrs_hash_join *j) {
  Ctx *ctx = j->ctx;
  const char *dd_e = hook("SQLRS_JOIN_DENSE_DUP"); // test / A-B hook, read per call: 0 = the general table
  if ((dd_e && std::atoi(dd_e) == 0) || !j->dup_range || !j->exact || !j->bkeys || j->bkeys_validity || j->nB <= 0 ||
      j->nB > 0x7fffffffll || j->dup_range >= (1ull << 31))
    return false;
  const uint32_t *mult = hash_join_dup_mult(j);
  if (!mult) return false;
  ProfScope ps(ctx, "join_build_dense_dup");
  const int64_t range = (int64_t)j->dup_range, n = j->nB;
  BufP start = ctx->alloc(4 * (size_t)range + 8), total = ctx->alloc(8);
  exclusive_scan_u32(ctx, mult, range, nullptr, start->as<uint32_t>(), total->as<uint64_t>());
  j->dd_table = ctx->alloc(4 * (size_t)range + 16);
  dd_table_kernel<<<dim3((unsigned)ceil_div(range + 3, 256)), dim3(256), 0, ctx->stream>>>(start->as<uint32_t>(), range, (uint32_t)n, j->dd_table->as<uint32_t>());
  BufP k64 = ctx->alloc(8 * (size_t)n);
  j->rows_by_slot = ctx->alloc(4 * (size_t)n);
  dd_sort_keys_kernel<<<dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream>>>(j->bkeys->as<uint64_t>(), n, j->dup_min, k64->as<uint64_t>());
  iota_u32(ctx, j->rows_by_slot->as<uint32_t>(), n);
  SQ_HIP(hipGetLastError());
  int bits = 1;
  while ((1ull << bits) < (uint64_t)range) bits++;
  radix_sort_pairs(ctx, k64->as<uint64_t>(), j->rows_by_slot->as<uint32_t>(), n, 0, bits); // (stable: a run keeps build insertion order, hash_join.rs:172-177)
  j->unique = false;
  j->table_built = j->unique_known = true; // (no general table: every probe of this join counts on dd_table)
  return true;
}

// 2. open-addressing table over the key hash (any key type, duplicates allowed)
static void build_hash_table(sqlrs_hash_join *j) {
  Ctx *ctx = j->ctx;
  const int64_t n = j->nB;
  BufP keys = j->bkeys, validity = j->bkeys_validity;
  j->table_built = j->unique_known = true;
  uint64_t cap = 64;
  while (2 * cap < 3 * (uint64_t)n) cap <<= 1; // load factor <= 2/3
  j->mask = cap - 1;
  int64_t nslots = (int64_t)cap + 2;
  j->table = ctx->alloc(sizeof(Slot) * (size_t)nslots);
  BufP row_slot = ctx->alloc(4 * (size_t)std::max<int64_t>(n, 1));
  BufP dup = ctx->alloc_zero(8);
  {
    ProfScope ps(ctx, "join_build");
    table_init_kernel<<<dim3((unsigned)ceil_div(nslots, 256)), dim3(256), 0, ctx->stream>>>(
        (j->table ? j->table->as<Slot>() : nullptr), nslots);
    if (n)
      join_insert_kernel<<<dim3((unsigned)ceil_div(n, BLOCK)), dim3(BLOCK), 0, ctx->stream>>>(
          keys->as<uint64_t>(), validity ? validity->as<uint64_t>() : nullptr, n,
          (j->table ? j->table->as<Slot>() : nullptr), j->mask, row_slot->as<uint32_t>(), dup->as<int>());
    SQ_HIP(hipGetLastError());
  }
  j->unique = ctx->fetch_value(dup->as<int>()) == 0;
  if (!j->unique) {
    // CSR: head = exclusive scan of counts in slot order; rows stably sorted by slot
    ProfScope ps(ctx, "join_build_csr");
    BufP counts = ctx->alloc(4 * (size_t)nslots), heads = ctx->alloc(4 * (size_t)nslots);
    BufP total = ctx->alloc(8);
    slot_counts_kernel<<<dim3((unsigned)ceil_div(nslots, 256)), dim3(256), 0, ctx->stream>>>(
        (j->table ? j->table->as<Slot>() : nullptr), nslots, counts->as<uint32_t>());
    exclusive_scan_u32(ctx, counts->as<uint32_t>(), nslots, nullptr, heads->as<uint32_t>(),
                       total->as<uint64_t>());
    slot_heads_kernel<<<dim3((unsigned)ceil_div(nslots, 256)), dim3(256), 0, ctx->stream>>>(
        (j->table ? j->table->as<Slot>() : nullptr), nslots, heads->as<uint32_t>());
    BufP k64 = ctx->alloc(8 * (size_t)n);
    j->rows_by_slot = ctx->alloc(4 * (size_t)n);
    u32_to_u64_kernel<<<dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream>>>(
        row_slot->as<uint32_t>(), n, k64->as<uint64_t>());
    iota_u32(ctx, j->rows_by_slot->as<uint32_t>(), n);
    SQ_HIP(hipGetLastError());
    int bits = 1;
    while ((1ull << bits) < (uint64_t)nslots) bits++;
    radix_sort_pairs(ctx, k64->as<uint64_t>(), j->rows_by_slot->as<uint32_t>(), n, 0, bits);
  }
}

void hash_join_ensure_table(sqlrs_hash_join *j) {
  dense_resolve(j);
  if (!j->table_built && j->finished && !j->empty_build && !build_dense_dup(j)) build_hash_table(j);
}

static DenseTable dense_table_of(const sqlrs_hash_join *j) {
  DenseTable dt;
  dt.heads = j->dense ? j->dense->as<uint32_t>() : nullptr;
  dt.kmin = j->dense_min;
  dt.range = j->dense_range;
  dt.null_head = j->dense_null_head;
  if (j->dense && j->dense_packed) {
    dt.packed = j->dense_packed->as<uint8_t>();
    dt.bits = j->dense_pbits;
    dt.pmask = (1u << j->dense_pbits) - 1;
  }
  return dt;
}
static Pairs probe_pairs(sqlrs_hash_join *j, const NKeys &pk) {
  Ctx *ctx = j->ctx;
  if (pk.exact != j->exact || (pk.exact && pk.dtype != j->key_dtype))
    fail(SQLRS_ERR_INTERNAL, "join keys of different types on the two sides are not supported");
  Pairs p;
  int64_t n = pk.rows;
  int outer_right = j->join_type == SQLRS_JOIN_RIGHT || j->join_type == SQLRS_JOIN_FULL;
  // The first probe of a direct-address build whose verdict is still on the device: the optimistic all-hit kernel takes
  // key range and uniqueness from the device-side words and ONE fetch brings back the build's verdict and the probe's —
  // a host round trip less per join (C3: 8 MB of build keys cost 0.07 ms, a third of it that round trip).
  if (j->dense_pending) {
    const char *ah_e = hook("SQLRS_PROBE_ALLHIT");
    if (!outer_right && !pk.validity && n >= (1 << 16) && n <= 0xffffffffll && j->pend_bits && !(ah_e && std::atoi(ah_e) == 0)) {
      ProfScope ps(ctx, "join_probe_dense");
      p.left = ctx->alloc(8 * (size_t)n);
      p.right = ctx->alloc(4 * (size_t)n);
      unsigned long long *stp = j->pend_st->as<unsigned long long>();
      unsigned int *miss = (unsigned int *)(stp + DENSE_ST_WORDS);
      DenseTable dt;
      dt.heads = j->pend_dense->as<uint32_t>();
      dt.packed = j->pend_packed->as<uint8_t>();
      dt.bits = j->pend_bits;
      dt.pmask = (1u << j->pend_bits) - 1;
      dt.st = stp;
      dt.st_max_range = j->pend_max_range;
      dt.st_rows = (uint64_t)j->nB;
      const int64_t every = std::max<int64_t>(1, n >> 14); // ~16 K sampled rows
      join_probe_dense_sample_kernel<<<dim3((unsigned)ceil_div(ceil_div(n, every), 256)), dim3(256), 0, ctx->stream>>>(
          pk.keys->as<uint64_t>(), n, every, dt, miss);
      const unsigned pblocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n / JAP_ROWS, 4), JAP_GRID * (int64_t)ctx->num_cus));
      if (((uintptr_t)pk.keys->p & 15) == 0)
        join_probe_dense_allhit_packed_kernel<true><<<dim3(pblocks), dim3(256), 0, ctx->stream>>>(pk.keys->as<uint64_t>(), n, dt, p.left->as<uint64_t>(),
                                                                                               p.right->as<uint32_t>(), miss);
      else
        join_probe_dense_allhit_packed_kernel<false><<<dim3(pblocks), dim3(256), 0, ctx->stream>>>(pk.keys->as<uint64_t>(), n, dt, p.left->as<uint64_t>(),
                                                                                                p.right->as<uint32_t>(), miss);
      SQ_HIP(hipGetLastError());
      uint64_t hw[DENSE_ST_WORDS + 1];
      std::memcpy(hw, ctx->fetch(j->pend_st->p, 8 * (DENSE_ST_WORDS + 1)), sizeof(hw));
      dense_resolve(j, hw);
      if (j->dense && (unsigned int)hw[DENSE_ST_WORDS] == 0) { // a unique dense key set and every probe row found its partner
        p.m = n;
        p.right_identity = true;
        return p;
      }
      if (j->dense) j->probe_miss_seen = true; // (later batches of this join go straight to the compacting kernel)
      p = Pairs();
    }
  }
  // a build side whose uniqueness came from its LDS bucket tables (lds_build_first) has no global table yet: it is built
  // only when this batch cannot take the LDS route
  LdsJoinMatch lm;
  dense_resolve(j);
  if (j->lds_first && !j->table_built && j->unique && !outer_right && !j->dense && n > 0) lm = lds_join_match(j, pk);
  if (!lm.ok) hash_join_ensure_table(j);
  if (n == 0) {
    p.left = ctx->alloc(8);
    p.right = ctx->alloc(8);
    return p;
  }
  if (n > 0xffffffffll) fail(SQLRS_ERR_INTERNAL, "probe batch larger than 2^32 rows");
  dim3 g((unsigned)ceil_div(n, BLOCK)), b(BLOCK);
  // general keys, build side beyond an L2-resident table: LDS tables over a blocked partition (see lds_join_probe_kernel)
  if (!lm.ok && j->unique && !outer_right && !j->dense) lm = lds_join_match(j, pk);
  if (j->unique && !outer_right) { // one lookup per row, compaction with look-back
    int64_t tiles = lm.ok ? lds_join_tiles(n) : ceil_div(n, j->dense ? JD_TILE : JP_TILE);
    p.left = ctx->alloc(8 * (size_t)n);
    p.right = ctx->alloc(4 * (size_t)n);
    // optimistic all-hit attempt of the direct-address probe (join_probe_dense_allhit_kernel); its miss flag is a word of its
    // own (a zeroed slab: no memset) and is NOT cleared by a look-back rerun.  SQLRS_PROBE_ALLHIT=0 (read per call): never.
    // The look-back descriptors are allocated (and cleared: a memset) only when the compacting kernel runs.
    unsigned int *miss = nullptr;
    BufP miss_buf;
    {
      const char *ah_e = hook("SQLRS_PROBE_ALLHIT");
      if (j->dense && !lm.ok && !pk.validity && n >= (1 << 16) && !j->probe_miss_seen && !(ah_e && std::atoi(ah_e) == 0)) {
        miss_buf = ctx->alloc_zero(8);
        miss = miss_buf->as<unsigned int>();
        ProfScope ps(ctx, "join_probe_dense");
        DenseTable dt = dense_table_of(j);
        const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(n, 256 * JA_ILP), 16 * (int64_t)ctx->num_cus);
        const int64_t every = std::max<int64_t>(1, n >> 14); // ~16 K sampled rows
        join_probe_dense_sample_kernel<<<dim3((unsigned)ceil_div(ceil_div(n, every), 256)), dim3(256), 0, ctx->stream>>>(
            pk.keys->as<uint64_t>(), n, every, dt, miss);
        const char *sc_e = hook("SQLRS_PROBE_ALLHIT_SC1"); // A/B hook, read per call
        if (dt.packed) { // wave-contiguous chunks over the bit-packed table
          const unsigned pblocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n / JAP_ROWS, 4), JAP_GRID * (int64_t)ctx->num_cus));
          if (((uintptr_t)pk.keys->p & 15) == 0)
            join_probe_dense_allhit_packed_kernel<true><<<dim3(pblocks), dim3(256), 0, ctx->stream>>>(pk.keys->as<uint64_t>(), n, dt, p.left->as<uint64_t>(),
                                                                                                   p.right->as<uint32_t>(), miss);
          else
            join_probe_dense_allhit_packed_kernel<false><<<dim3(pblocks), dim3(256), 0, ctx->stream>>>(pk.keys->as<uint64_t>(), n, dt, p.left->as<uint64_t>(),
                                                                                                    p.right->as<uint32_t>(), miss);
        } else if (sc_e && std::atoi(sc_e) == 1)
          join_probe_dense_allhit_kernel<true><<<dim3(blocks), dim3(256), 0, ctx->stream>>>(pk.keys->as<uint64_t>(), n, dt, p.left->as<uint64_t>(),
                                                                                          p.right->as<uint32_t>(), miss);
        else
          join_probe_dense_allhit_kernel<false><<<dim3(blocks), dim3(256), 0, ctx->stream>>>(pk.keys->as<uint64_t>(), n, dt, p.left->as<uint64_t>(),
                                                                                           p.right->as<uint32_t>(), miss);
        SQ_HIP(hipGetLastError());
        const char *hc_e = hook("SQLRS_PROBE_ALLHIT_HOSTCHECK"); // A/B hook, read per call (default on)
        if (!(hc_e && std::atoi(hc_e) == 0)) {
          if (ctx->fetch_value(miss) == 0) { // every pair is in place
            p.m = n;
            p.right_identity = true;
            return p;
          }
          j->probe_miss_seen = true; // (later batches of this join go straight to the compacting kernel)
        }
      }
    }
    BufP desc = ctx->alloc_zero(8 * (size_t)tiles + 24);
    unsigned *ticket = (unsigned *)(desc->as<uint64_t>() + tiles);
    uint64_t *tot = desc->as<uint64_t>() + tiles + 1;
    for (int use_ticket = lookback_start_mode(ctx), attempt = 0; use_ticket < 2; use_ticket++, attempt++) {
      if (attempt) SQ_HIP(hipMemsetAsync(desc->p, 0, 8 * (size_t)tiles + 16, ctx->stream)); // rerun after a timeout
      {
        ProfScope ps(ctx, lm.ok ? "join_match_compact" : (j->dense ? "join_probe_dense" : "join_probe_unique"));
        dim3 gt((unsigned)tiles);
        DenseTable dt = dense_table_of(j);
        if (lm.ok) {
          lds_join_restore(ctx, lm, n, p.left->as<uint64_t>(), p.right->as<uint32_t>(), desc->as<uint64_t>(), ticket, tot, use_ticket);
        } else if (j->dense && pk.validity)
          join_probe_dense_kernel<true><<<gt, dim3(JD_BLOCK), 0, ctx->stream>>>(
              pk.keys->as<uint64_t>(), pk.validity, n, tiles, dt, p.left->as<uint64_t>(), p.right->as<uint32_t>(),
              desc->as<uint64_t>(), ticket, tot, use_ticket);
        else if (j->dense)
          join_probe_dense_kernel<false><<<gt, dim3(JD_BLOCK), 0, ctx->stream>>>(
              pk.keys->as<uint64_t>(), pk.validity, n, tiles, dt, p.left->as<uint64_t>(), p.right->as<uint32_t>(),
              desc->as<uint64_t>(), ticket, tot, use_ticket, miss);
        else
          join_probe_unique_kernel<false><<<gt, b, 0, ctx->stream>>>(
              pk.keys->as<uint64_t>(), pk.validity, n, tiles, (j->table ? j->table->as<Slot>() : nullptr), j->mask, dt,
              p.left->as<uint64_t>(), p.right->as<uint32_t>(), desc->as<uint64_t>(), ticket, tot, use_ticket);
        SQ_HIP(hipGetLastError());
      }
      const uint64_t *h = (const uint64_t *)ctx->fetch(ticket, 16); // {ticket|timeout, total}
      p.m = (int64_t)h[1];
      if (use_ticket || (h[0] >> 32) == 0) break;
      lookback_timed_out(ctx);
    }
    // unique build keys, pairs in probe-row order: as many pairs as probe rows = every row matched once = pair i is (.., i)
    p.right_identity = p.m == n;
    return p;
  }
  // duplicate build keys, and Right / Full joins over general keys: matched on the LDS tables too, un-permuted into the
  // {run, pairs} the fill pass expands (lds_join_unpermute_kernel)
  LdsJoinMatch lmg;
  if (!j->dense && !j->dd_table && (!j->unique || outer_right)) lmg = lds_join_match(j, pk, !j->unique);
  if (j->unique && outer_right && !lmg.ok) { // exactly one pair per probe row
    p.m = n;
    p.right_identity = true; // (pair i = (build row | NULL, probe row i))
    p.left = ctx->alloc(8 * (size_t)n);
    p.right = ctx->alloc(4 * (size_t)n);
    p.left_validity = ctx->alloc(bitmap_bytes(n));
    ProfScope ps(ctx, "join_probe_unique");
    int64_t n64 = (int64_t)round_up((size_t)n, 64);
    DenseTable dt = dense_table_of(j);
    if (j->dense)
      join_probe_unique_outer_kernel<true><<<dim3((unsigned)ceil_div(n64, BLOCK)), b, 0, ctx->stream>>>(
          pk.keys->as<uint64_t>(), pk.validity, n, (j->table ? j->table->as<Slot>() : nullptr), j->mask, dt, p.left->as<uint64_t>(),
          p.right->as<uint32_t>(), p.left_validity->as<uint64_t>());
    else
      join_probe_unique_outer_kernel<false><<<dim3((unsigned)ceil_div(n64, BLOCK)), b, 0, ctx->stream>>>(
          pk.keys->as<uint64_t>(), pk.validity, n, (j->table ? j->table->as<Slot>() : nullptr), j->mask, dt, p.left->as<uint64_t>(),
          p.right->as<uint32_t>(), p.left_validity->as<uint64_t>());
    SQ_HIP(hipGetLastError());
    return p;
  }
  // The fill pass scans the pair counts of its 64 probe rows itself, so only one sum per 64-row group is scanned globally
  // (round 6: per-row counts + offsets were 0.4 + 0.8 GB written and 1.2 GB read for 1e8 probe rows, 0.45 ms of scan).  A build
  // side of >= 2^26 rows — 64 runs of that length overflow a 32-bit sum — keeps per-row counts.
  const char *gr_e = hook("SQLRS_JOIN_GROUPED"); // test hook, read per call: 0 = per-row counts whatever the build side's size
  const int grouped = (j->nB < (1ll << 26) && !(gr_e && std::atoi(gr_e) == 0)) ? 1 : 0;
  const int64_t nscan = grouped ? ceil_div(n, 64) : n;
  BufP counts = ctx->alloc(4 * (size_t)nscan), offsets = ctx->alloc(8 * (size_t)nscan), total = ctx->alloc(8);
  BufP match = ctx->alloc(8 * (size_t)n);
  if (lmg.ok) {
    ProfScope ps(ctx, "join_match_unpermute");
    lds_join_unpermute(j, lmg, n, outer_right, match->as<uint2>(), counts->as<uint32_t>(), grouped);
  } else if (j->dd_table) { // duplicate keys over a dense range: {run, rows} by direct address
    ProfScope ps(ctx, "join_probe_count_dense_dup");
    const char *ds_e = hook("SQLRS_DD_STREAM"); // A/B hook, read per call: 0 = the one-row-per-lane form for every batch
    if (!pk.validity && n >= JAP_ROWS && ((uintptr_t)pk.keys->p & 15) == 0 && !(ds_e && std::atoi(ds_e) == 0)) {
      const unsigned pblocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n / JAP_ROWS, 4), JAP_GRID * (int64_t)ctx->num_cus));
      dd_count_stream_kernel<<<dim3(pblocks), dim3(256), 0, ctx->stream>>>(pk.keys->as<uint64_t>(), n, j->dup_min, j->dup_range, j->dd_table->as<uint32_t>(),
                                                                         outer_right, counts->as<uint32_t>(), match->as<uint2>(), grouped);
    } else
      dd_count_kernel<<<dim3((unsigned)ceil_div(n, (int64_t)BLOCK * DD_U)), b, 0, ctx->stream>>>(pk.keys->as<uint64_t>(), pk.validity, n, j->dup_min, j->dup_range, j->dd_table->as<uint32_t>(), outer_right,
                                                counts->as<uint32_t>(), match->as<uint2>(), grouped);
    SQ_HIP(hipGetLastError());
  } else {
    ProfScope ps(ctx, "join_probe_count");
    join_count_kernel<<<g, b, 0, ctx->stream>>>(pk.keys->as<uint64_t>(), pk.validity, n,
                                                (j->table ? j->table->as<Slot>() : nullptr), j->mask, outer_right,
                                                counts->as<uint32_t>(), match->as<uint2>(), grouped);
    SQ_HIP(hipGetLastError());
  }
  exclusive_scan_u32(ctx, counts->as<uint32_t>(), nscan, offsets->as<uint64_t>(), nullptr,
                     total->as<uint64_t>());
  p.m = (int64_t)ctx->fetch_value(total->as<uint64_t>());
  int64_t m1 = std::max<int64_t>(p.m, 1);
  p.left = ctx->alloc(8 * (size_t)m1);
  p.right = ctx->alloc(4 * (size_t)m1);
  BufP lvb;
  if (outer_right) lvb = ctx->alloc((size_t)m1);
  if (p.m) {
    ProfScope ps(ctx, "join_probe_fill");
    join_fill_expand_kernel<<<dim3((unsigned)ceil_div(n, BLOCK)), b, 0, ctx->stream>>>(
        match->as<uint2>(), n, j->unique ? 1 : 0, outer_right, j->rows_by_slot ? j->rows_by_slot->as<uint32_t>() : nullptr,
        offsets->as<uint64_t>(), p.left->as<uint64_t>(), p.right->as<uint32_t>(), lvb ? lvb->as<uint8_t>() : nullptr, grouped);
    SQ_HIP(hipGetLastError());
  }
  if (outer_right) {
    p.left_validity = ctx->alloc(bitmap_bytes(m1));
    int64_t m64 = (int64_t)round_up((size_t)m1, 64);
    bytes_to_bits_kernel<<<dim3((unsigned)ceil_div(m64, 256)), dim3(256), 0, ctx->stream>>>(
        lvb->as<uint8_t>(), p.m, p.left_validity->as<uint64_t>());
    SQ_HIP(hipGetLastError());
  }
  return p;
}

static DBatch gather_pairs(sqlrs_hash_join *j, const DBatch &right, const Pairs &p) {
  Ctx *ctx = j->ctx;
  DBatch out;
  out.rows = p.m;
  const uint64_t *lv = p.left_validity ? p.left_validity->as<uint64_t>() : nullptr;
  // Join-key equivalence: with one exactly-compared key `left[c] = right[rc]` every emitted pair
  // carries equal key values on both sides (NULL = NULL included, hash_utils.rs:91-104), so for
  // Inner/Left the gathered build key column is the gathered probe key column: the random
  // gather over the build side is skipped.
  int lkey_col = -1, rkey_col = -1;
  bool outer_right = j->join_type == SQLRS_JOIN_RIGHT || j->join_type == SQLRS_JOIN_FULL;
  if (!outer_right && j->exact && j->lkeys.size() == 1 && j->lkeys[0].nodes.size() == 1 &&
      j->rkeys[0].nodes.size() == 1 && j->lkeys[0].nodes[0].op == SQLRS_EXPR_INPUT_REF &&
      j->rkeys[0].nodes[0].op == SQLRS_EXPR_INPUT_REF) {
    lkey_col = j->lkeys[0].nodes[0].index;
    rkey_col = j->rkeys[0].nodes[0].index;
    if (lkey_col < 0 || (size_t)lkey_col >= j->left.cols.size() || rkey_col < 0 ||
        (size_t)rkey_col >= right.cols.size() ||
        j->left.cols[(size_t)lkey_col].dtype != right.cols[(size_t)rkey_col].dtype)
      lkey_col = rkey_col = -1;
  }
  std::vector<DCol> rcols;
  for (const DCol &c : right.cols)
    rcols.push_back(p.right_identity ? c : gather_column(ctx, c, p.right->p, false, nullptr, p.m));
  for (size_t c = 0; c < j->left.cols.size(); c++) {
    if ((int)c == lkey_col)
      out.cols.push_back(rcols[(size_t)rkey_col]);
    else
      out.cols.push_back(gather_column(ctx, j->left.cols[c], p.left->p, true, lv, p.m));
  }
  for (DCol &c : rcols) out.cols.push_back(std::move(c));
  return out;
}

// apply_join_filter  (hash_join.rs:47-127)
static void apply_filter(sqlrs_hash_join *j, const DBatch &right, Pairs &p) {
  Ctx *ctx = j->ctx;
  DBatch inter = gather_pairs(j, right, p); // intermediate batch (:256-262)
  auto colfn = [&](int i) -> const DCol & {
    if (i < 0 || (size_t)i >= inter.cols.size()) fail(SQLRS_ERR_INTERNAL, "input ref out of range");
    return inter.cols[(size_t)i];
  };
  DCol mask = eval_expr(ctx, j->filter, colfn, inter.rows, false);
  mask.length = inter.rows;
  Selection sel = selection_from_mask(ctx, mask);
  DCol lcol, rcol;
  lcol.dtype = SQLRS_UINT64;
  lcol.length = p.m;
  lcol.values = p.left->p;
  lcol.own_values = p.left;
  if (p.left_validity) {
    lcol.validity = p.left_validity->as<uint64_t>();
    lcol.own_validity = p.left_validity;
    lcol.null_count = -1;
  }
  rcol.dtype = SQLRS_UINT32;
  rcol.length = p.m;
  rcol.values = p.right->p;
  rcol.own_values = p.right;
  DCol lf = compact_column(ctx, lcol, sel), rf = compact_column(ctx, rcol, sel);
  bool outer_right = j->join_type == SQLRS_JOIN_RIGHT || j->join_type == SQLRS_JOIN_FULL;
  if (!outer_right) {
    p.m = sel.count;
    p.left = lf.own_values;
    p.right = rf.own_values;
    p.left_validity = lf.own_validity; // always all-valid here
    return;
  }
  // keep every right row: rows that lost all their matches come back as (NULL, row) (:73-121)
  int64_t nr = right.rows;
  BufP visited = ctx->alloc_zero(bitmap_bytes(std::max<int64_t>(nr, 1)));
  if (sel.count)
    mark_bits_kernel<uint32_t><<<dim3((unsigned)ceil_div(sel.count, 256)), dim3(256), 0, ctx->stream>>>(
        rf.v<uint32_t>(), nullptr, sel.count, visited->as<unsigned long long>());
  SQ_HIP(hipGetLastError());
  Selection unv = selection_from_clear_bits(ctx, visited->as<uint64_t>(), nr);
  BufP uidx = selection_indices_u32(ctx, unv);
  DCol ucol;
  ucol.dtype = SQLRS_UINT32;
  ucol.length = unv.count;
  ucol.values = uidx->p;
  ucol.own_values = uidx;
  DCol lnull = make_null_column(ctx, SQLRS_UINT64, unv.count);
  if (!lf.validity) lf.null_count = 0;
  DCol l2 = concat_columns(ctx, {&lf, &lnull});
  DCol r2 = concat_columns(ctx, {&rf, &ucol});
  p.m = sel.count + unv.count;
  p.left = l2.own_values;
  p.right = r2.own_values;
  p.left_validity = l2.own_validity;
  if (!p.left_validity && unv.count == 0 && lf.own_validity) p.left_validity = lf.own_validity;
}

__global__ void dup_mult_kernel(const uint64_t *__restrict__ keys, int64_t n, uint64_t kmin, uint32_t *__restrict__ mult) {
  const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r < n) atomicAdd(&mult[keys[r] - kmin], 1u);
}
const uint32_t *hash_join_dup_mult(sqlrs_hash_join *j) {
  dense_resolve(j);
  if (!j->dup_range || !j->bkeys || j->bkeys_validity) return nullptr;
  if (!j->dup_mult) {
    Ctx *ctx = j->ctx;
    j->dup_mult = ctx->alloc_zero(4 * (size_t)j->dup_range + 8);
    dup_mult_kernel<<<dim3((unsigned)ceil_div(j->nB, 256)), dim3(256), 0, ctx->stream>>>(j->bkeys->as<uint64_t>(), j->nB, j->dup_min,
                                                                                      j->dup_mult->as<uint32_t>());
    SQ_HIP(hipGetLastError());
  }
  return j->dup_mult->as<uint32_t>();
}

const uint64_t *hash_join_dense_bits(sqlrs_hash_join *j) {
  dense_resolve(j);
  if (!j->dense || !j->dense_range) return nullptr;
  if (!j->dense_bits) {
    Ctx *ctx = j->ctx;
    const int64_t words = ceil_div((int64_t)j->dense_range, 64);
    j->dense_bits = ctx->alloc(8 * (size_t)words + 8);
    dense_bits_kernel<<<dim3((unsigned)ceil_div(words * 64, 256)), dim3(256), 0, ctx->stream>>>(
        j->dense->as<uint32_t>(), (int64_t)j->dense_range, j->dense_bits->as<uint64_t>());
    SQ_HIP(hipGetLastError());
  }
  return j->dense_bits->as<uint64_t>();
}

// Key-only build side (see dense_bits_kernel): true = `out` holds the joined batch
static bool semi_join_probe(sqlrs_hash_join *j, InBatch &ib, const NKeys &pk, DBatch *out) {
  Ctx *ctx = j->ctx;
  const char *env_e = hook("SQLRS_SEMI_JOIN"); // test hook, read per call: 0 = never
  if (env_e && std::atoi(env_e) == 0) return false;
  if (j->dense_pending && j->join_type == SQLRS_JOIN_INNER && !j->has_filter && j->left.cols.size() == 1) dense_resolve(j); // (a candidate: decide now)
  if (j->join_type != SQLRS_JOIN_INNER || j->has_filter || !j->unique || !j->dense || !j->exact || pk.validity ||
      j->left.cols.size() != 1 || j->lkeys.size() != 1 || j->lkeys[0].nodes.size() != 1 || j->rkeys[0].nodes.size() != 1 ||
      j->lkeys[0].nodes[0].op != SQLRS_EXPR_INPUT_REF || j->rkeys[0].nodes[0].op != SQLRS_EXPR_INPUT_REF ||
      j->lkeys[0].nodes[0].index != 0)
    return false;
  const int rkey_col = j->rkeys[0].nodes[0].index;
  const int64_t n = ib.rows();
  if (rkey_col < 0 || rkey_col >= ib.num_columns() || n < (1 << 16)) return false;
  if (j->left.cols[0].dtype != ib.col(rkey_col).dtype) return false;
  // Unique build keys that FILL their range (as many rows as the range has values, none NULL: a dimension's surrogate keys) —
  // a probe key inside the range has its partner, no table says more: the mask is a range test over the key stream (5e8 probe
  // rows: 2.55 -> ms of lookups in the 1.25 MB bitmap gone; the fused route takes the same shortcut, hashagg_op.hip)
  const bool full_range = j->unique && j->unique_known && !j->bkeys_validity && j->dense_range == (uint64_t)j->nB;
  if (!full_range) hash_join_dense_bits(j);
  Selection sel;
  sel.rows = n;
  const int64_t nwords = ceil_div(n, 64);
  sel.own_bits = ctx->alloc(8 * (size_t)nwords + 64);
  sel.bits = sel.own_bits->as<uint64_t>();
  {
    ProfScope ps(ctx, "join_semi_mask");
    const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(nwords, WAVES_PER_BLOCK * SM_U), 8 * (int64_t)ctx->num_cus);
    BufP hits = ctx->alloc_zero(8);
    semi_mask_kernel<<<dim3(std::max(blocks, 1u)), dim3(BLOCK), 0, ctx->stream>>>(
        pk.keys->as<uint64_t>(), n, full_range ? nullptr : j->dense_bits->as<uint64_t>(), j->dense_min, j->dense_range, sel.own_bits->as<uint64_t>(),
        hits->as<unsigned long long>());
    SQ_HIP(hipGetLastError());
    sel.count = (int64_t)ctx->fetch_value(hits->as<uint64_t>());
  }
  if (sel.count != n) selection_finish(ctx, sel); // (tile offsets are only needed to compact: 1.4 ms per 5e8 rows)
  // (all rows kept: the output SHARES the probe columns — library-owned buffers by reference, a caller's borrowed
  //  device buffers as private copies, since a batch is only borrowed for the call)
  DBatch right = ib.materialize(sel.count == n);
  out->rows = sel.count;
  std::vector<DCol> rcols;
  if (sel.count == n) { // every probe row has its partner (PK-FK): the probe columns ARE the joined rows
    rcols = right.cols;
  } else {
    for (const DCol &c : right.cols) rcols.push_back(compact_column(ctx, c, sel));
  }
  out->cols.push_back(rcols[(size_t)rkey_col]); // the build key column = the probe key column of the matched rows
  for (DCol &c : rcols) out->cols.push_back(std::move(c));
  return true;
}

// `also` (optional): the joined batch AND its pairs (probe_push_many cuts the batch at probe-row boundaries)
static DBatch probe_batch(sqlrs_hash_join *j, InBatch &ib, Pairs *pairs_only, Pairs *also = nullptr) {
  Ctx *ctx = j->ctx;
  auto colfn = [&](int i) -> const DCol & { return ib.col(i); };
  NKeys pk = j->comp.on ? composite_probe_keys(j, colfn, ib.rows()) : eval_keys(ctx, j->rkeys, colfn, ib.rows());
  if (!pairs_only && !also) {
    DBatch semi;
    if (semi_join_probe(j, ib, pk, &semi)) return semi;
  }
  Pairs p = probe_pairs(j, pk);
  if (also) *also = p;
  if (pairs_only) {
    *pairs_only = p;
    return DBatch();
  }
  if (j->has_filter) p.right_identity = false; // (the join filter selects among the pairs)
  // every probe row matched exactly once: the joined batch takes the probe columns as they are — shared when they
  // are this library's own buffers (an upstream operator's output), copied once when the caller only lent them
  DBatch right = ib.materialize(p.right_identity);
  if (j->has_filter) apply_filter(j, right, p);
  if ((j->join_type == SQLRS_JOIN_LEFT || j->join_type == SQLRS_JOIN_FULL) && p.m) {
    mark_bits_kernel<uint64_t><<<dim3((unsigned)ceil_div(p.m, 256)), dim3(256), 0, ctx->stream>>>(
        p.left->as<uint64_t>(), p.left_validity ? p.left_validity->as<uint64_t>() : nullptr, p.m,
        j->visited->as<unsigned long long>()); // :274-282
    SQ_HIP(hipGetLastError());
  }
  return gather_pairs(j, right, p); // :284-291
}

} // namespace sq

extern "C" {

int sqlrs_hash_join_create(sqlrs_ctx_t *ctx, int join_type, int num_keys,
                           const sqlrs_expr_t *left_keys, const sqlrs_expr_t *right_keys,
                           const sqlrs_expr_t *filter, int num_right_columns,
                           const int32_t *right_dtypes, sqlrs_hash_join_t **out) {
  return guard(ctx, [&] {
    if (num_keys < 1) fail(SQLRS_ERR_INTERNAL, "HashJoin must has on condition"); // :132
    if (join_type < SQLRS_JOIN_INNER || join_type > SQLRS_JOIN_FULL)
      fail(SQLRS_ERR_INTERNAL, "bad join type");
    auto j = std::unique_ptr<sqlrs_hash_join>(new sqlrs_hash_join());
    j->ctx = ctx;
    j->join_type = join_type;
    for (int i = 0; i < num_keys; i++) {
      j->lkeys.push_back(expr_from_abi(&left_keys[i]));
      j->rkeys.push_back(expr_from_abi(&right_keys[i]));
    }
    if (filter && filter->num_nodes > 0) {
      j->has_filter = true;
      j->filter = expr_from_abi(filter);
    }
    if (num_right_columns > 0) j->right_dtypes.assign(right_dtypes, right_dtypes + num_right_columns);
    *out = j.release();
  });
}

static int hash_join_build_push_device(sqlrs_hash_join_t *j, const sqlrs_batch_t *left);
static int hash_join_flush_host(sqlrs_hash_join_t *j) {
  if (!j->hstage.has_schema) return SQLRS_OK;
  sqlrs_batch_t *dev = nullptr;
  int st = guard(j->ctx, [&] {
    SQ_HIP(hipSetDevice(j->ctx->device));
    dev = j->hstage.take();
  });
  if (st != SQLRS_OK) return st;
  st = hash_join_build_push_device(j, dev);
  sqlrs_batch_release(dev);
  return st;
}
int sqlrs_hash_join_build_push(sqlrs_hash_join_t *j, const sqlrs_batch_t *left) {
  j->hstage.ctx = j->ctx;
  if (!j->finished && j->hstage.accepts(left)) {
    int st = guard(j->ctx, [&] { j->hstage.append(left); });
    if (st != SQLRS_OK || j->hstage.rows < HOST_STAGE_FLUSH_ROWS) return st;
    return hash_join_flush_host(j);
  }
  int st = hash_join_flush_host(j);
  return st != SQLRS_OK ? st : hash_join_build_push_device(j, left);
}
static int hash_join_build_push_device(sqlrs_hash_join_t *j, const sqlrs_batch_t *left) {
  return guard(j->ctx, [&] {
    SQ_HIP(hipSetDevice(j->ctx->device));
    if (j->finished) fail(SQLRS_ERR_INTERNAL, "build_push after build_finish");
    InBatch ib(j->ctx, left);
    DBatch b = ib.materialize(true);
    auto colfn = [&](int i) -> const DCol & {
      if (i < 0 || (size_t)i >= b.cols.size()) fail(SQLRS_ERR_INTERNAL, "input ref out of range");
      return b.cols[(size_t)i];
    };
    // (the evaluated key columns are kept only for the opt-in composite key, build_table: 8 B x rows x keys of device memory
    //  otherwise held until build_finish for nothing)
    const char *ck_e = std::getenv("SQLRS_JOIN_COMPOSITE");
    if (j->lkeys.size() >= 2 && j->lkeys.size() <= 4 && !j->lazy_table && ck_e && std::atoi(ck_e) == 1) {
      std::vector<DCol> kc = eval_key_cols(j->ctx, j->lkeys, colfn, b.rows);
      j->left_key_parts.push_back(normalize_keys(j->ctx, kc, b.rows));
      j->left_keycol_parts.push_back(std::move(kc));
    } else
      j->left_key_parts.push_back(eval_keys(j->ctx, j->lkeys, colfn, b.rows));
    j->left_batches.push_back(std::move(b));
    j->empty_build = false;
  });
}

int sqlrs_hash_join_build_finish(sqlrs_hash_join_t *j) {
  int stf = hash_join_flush_host(j);
  if (stf != SQLRS_OK) return stf;
  return guard(j->ctx, [&] {
    SQ_HIP(hipSetDevice(j->ctx->device));
    if (j->finished) return;
    j->finished = true;
    if (j->empty_build) return; // the join emits nothing (:183-185)
    Ctx *ctx = j->ctx;
    size_t nc = j->left_batches[0].cols.size();
    for (size_t c = 0; c < nc; c++) {
      std::vector<const DCol *> parts;
      for (DBatch &b : j->left_batches) {
        if (b.cols.size() != nc) fail(SQLRS_ERR_ARROW, "concat_batches: schema mismatch");
        parts.push_back(&b.cols[c]);
      }
      j->left.cols.push_back(concat_columns(ctx, parts));
    }
    for (DBatch &b : j->left_batches) j->nB += b.rows;
    j->left.rows = j->nB;
    if (j->nB > 0xffffffffll) fail(SQLRS_ERR_INTERNAL, "build side larger than 2^32 rows");
    build_table(j);
    j->left_batches.clear();
    j->left_key_parts.clear();
    if (j->join_type == SQLRS_JOIN_LEFT || j->join_type == SQLRS_JOIN_FULL)
      j->visited = ctx->alloc_zero(bitmap_bytes(std::max<int64_t>(j->nB, 1)));
  });
}

int sqlrs_hash_join_probe_push(sqlrs_hash_join_t *j, const sqlrs_batch_t *right, int out_mem,
                               sqlrs_batch_t **out) {
  return guard(j->ctx, [&] {
    SQ_HIP(hipSetDevice(j->ctx->device));
    if (!j->finished) fail(SQLRS_ERR_INTERNAL, "probe before build_finish");
    *out = nullptr;
    if (j->empty_build) return;
    InBatch ib(j->ctx, right);
    DBatch r = probe_batch(j, ib, nullptr);
    *out = emit_batch(j->ctx, std::move(r), out_mem);
  });
}

} // extern "C"

namespace sq {
// ---- one small HOST probe batch, one launch, no copy call (small_async.hpp) ------------------------------------------
// Inner join over unique build keys: one lookup per probe row (the direct-address table or the 16-byte-slot table), the
// rows with a partner compacted in probe-row order (= the reference's pair order, hash_join.rs:225-234), the build
// columns gathered by the build row, the probe columns copied — the joined batch (build_batch, hash_join.rs:25-45)
// written straight into the pinned slot.
struct SaProbeParams {
  SaLayout lay;
  int nleft, key_col, key_is32, dense;
  const void *lvals[SA_MAX_COLS];
  const uint64_t *lvalid[SA_MAX_COLS];
  const Slot *table;
  uint64_t mask;
  DenseTable dt;
  const uint8_t *in;
  uint8_t *out;
  unsigned long long seq;
};
__global__ __launch_bounds__(1024) void sa_probe_kernel(SaGroup<SaProbeParams> grp) {
  const SaProbeParams &p = grp.p[blockIdx.x]; // (one workgroup per batch of the group)
  __shared__ uint32_t s_w[17], s_nulls[SA_MAX_COLS];
  __shared__ uint8_t s_v[SA_MAX_ROWS];
  if (threadIdx.x < SA_MAX_COLS) s_nulls[threadIdx.x] = 0;
  const SaCol &kc = p.lay.c[p.nleft + p.key_col];
  uint32_t m[4] = {DENSE_EMPTY, DENSE_EMPTY, DENSE_EMPTY, DENSE_EMPTY}, pos[4], total;
  const uint32_t bits = sa_positions(
      p.lay.rows,
      [&](uint32_t r, int t) {
        const uint64_t key = p.key_is32 ? (uint64_t)(int64_t)((const int32_t *)(p.in + kc.in_off))[r] : ((const uint64_t *)(p.in + kc.in_off))[r];
        uint32_t h = DENSE_EMPTY;
        if (p.dense) {
          const uint64_t d = key - p.dt.kmin;
          h = dense_get(p.dt, d < p.dt.range ? d : p.dt.range + 1);
        } else {
          const Slot sl = probe_slot(p.table, p.mask, key, false);
          if (sl.count) h = sl.head;
        }
        m[t] = h;
        return h != DENSE_EMPTY;
      },
      pos, s_w, &total);
  for (int c = 0; c < p.lay.ncols; c++) {
    const SaCol &col = p.lay.c[c];
    const bool left = c < p.nleft;
    const uint8_t *rvalid = !left && col.in_voff != SA_NONE ? p.in + col.in_voff : nullptr;
    const uint64_t *lvalid = left ? p.lvalid[c] : nullptr;
    const uint8_t *src = left ? (const uint8_t *)p.lvals[c] : p.in + col.in_off;
#pragma unroll
    for (int t = 0; t < 4; t++) {
      if (!((bits >> t) & 1)) continue;
      const uint32_t r = left ? m[t] : (uint32_t)t * 1024u + threadIdx.x;
      if (col.width == 8) ((uint64_t *)(p.out + col.out_off))[pos[t]] = ((const uint64_t *)src)[r];
      else ((uint32_t *)(p.out + col.out_off))[pos[t]] = ((const uint32_t *)src)[r];
      if (lvalid) s_v[pos[t]] = (uint8_t)((lvalid[r >> 6] >> (r & 63)) & 1);
      else if (rvalid) s_v[pos[t]] = (rvalid[r >> 3] >> (r & 7)) & 1;
    }
    if (lvalid || rvalid) sa_pack_validity(s_v, total, p.out + col.out_voff, &s_nulls[c]);
  }
  sa_publish((SaHeader *)p.out, p.seq, total, s_nulls, p.lay.ncols);
}
static void sa_probe_launch(SaRing *r, Ctx *ctx) {
  SaGroup<SaProbeParams> g;
  for (int i = 0; i < r->pend_n; i++) std::memcpy(&g.p[i], r->pend_buf + (size_t)i * SA_PARAM_MAX, sizeof(SaProbeParams));
  sa_probe_kernel<<<dim3((unsigned)r->pend_n), dim3(1024), 0, r->stream_of(r->pend_first_slot)>>>(g);
  SQ_HIP(hipGetLastError());
}
// true = the kernel above was queued for `right` and *t describes its slot
static bool sa_probe_try(sqlrs_hash_join *j, const sqlrs_batch_t *right, sqlrs_ticket *t) {
  Ctx *ctx = j->ctx;
  const char *off_e = hook("SQLRS_ASYNC_FAST"); // test hook, read per call: 0 = every batch through the synchronous operator
  if (off_e && off_e[0] == '0') return false;
  if (j->join_type != SQLRS_JOIN_INNER || j->has_filter || !j->exact || j->comp.on || j->lkeys.size() != 1 || j->rkeys[0].nodes.size() != 1 ||
      j->rkeys[0].nodes[0].op != SQLRS_EXPR_INPUT_REF || !right)
    return false;
  const int kc = j->rkeys[0].nodes[0].index;
  if (kc < 0 || kc >= right->num_columns) return false;
  const sqlrs_column_t &kcol = right->columns[kc];
  if (kcol.dtype != j->key_dtype || (kcol.validity && kcol.null_count != 0)) return false; // (NULL probe keys: the general route)
  if (kcol.dtype != SQLRS_INT64 && kcol.dtype != SQLRS_FLOAT64 && kcol.dtype != SQLRS_INT32) return false;
  const int nleft = (int)j->left.cols.size();
  if (nleft + right->num_columns > SA_MAX_COLS) return false;
  int32_t ldt[SA_MAX_COLS];
  for (int c = 0; c < nleft; c++) {
    const DCol &lc = j->left.cols[(size_t)c];
    if ((lc.dtype != SQLRS_INT32 && lc.dtype != SQLRS_INT64 && lc.dtype != SQLRS_FLOAT64) || lc.stride == 0) return false;
    ldt[c] = lc.dtype;
  }
  dense_resolve(j);
  if (!j->dense) hash_join_ensure_table(j);
  if (!j->unique || (!j->dense && !j->table)) return false;
  SaRing *r = sa_ring(ctx);
  const int slot = r ? sa_take_slot(r) : -1;
  if (slot < 0) return false;
  SaProbeParams p;
  if (!sa_stage_input(right, r->in_area(slot), &p.lay, nleft, ldt)) {
    r->busy[slot] = false;
    return false;
  }
  p.nleft = nleft;
  p.key_col = kc;
  p.key_is32 = kcol.dtype == SQLRS_INT32;
  p.dense = j->dense ? 1 : 0;
  for (int c = 0; c < SA_MAX_COLS; c++) {
    p.lvals[c] = c < nleft ? j->left.cols[(size_t)c].values : nullptr;
    p.lvalid[c] = c < nleft && j->left.cols[(size_t)c].has_nulls() ? j->left.cols[(size_t)c].validity : nullptr;
  }
  p.table = j->table ? j->table->as<Slot>() : nullptr;
  p.mask = j->mask;
  p.dt = dense_table_of(j);
  p.in = r->in_area(slot);
  p.out = r->out_area(slot);
  p.seq = ++r->seq;
  if (!j->async_ordered) { // the table and the build columns were queued on the ctx stream: the side streams wait for them, once
    sa_order_after_ctx(ctx, r);
    j->async_ordered = true;
  }
  sa_enqueue(ctx, r, j, sa_probe_launch, p, slot);
  t->slot = slot;
  t->seq = p.seq;
  t->lay = p.lay;
  return true;
}
} // namespace sq

extern "C" {
// sqlrs_hash_join_probe_push without the wait (small_async.hpp): *ticket stands for the HOST batch
// sqlrs_hash_join_probe_push(j, right, SQLRS_MEM_HOST, ..) would return — NULL for an empty build side.
int sqlrs_hash_join_probe_push_async(sqlrs_hash_join_t *j, const sqlrs_batch_t *right, sqlrs_ticket_t **ticket) {
  if (ticket) *ticket = nullptr;
  return guard(j->ctx, [&] {
    Ctx *ctx = j->ctx;
    if (!ticket) fail(SQLRS_ERR_INTERNAL, "push_async: null ticket");
    SQ_HIP(hipSetDevice(ctx->device));
    if (!j->finished) fail(SQLRS_ERR_INTERNAL, "probe before build_finish");
    auto t = std::unique_ptr<sqlrs_ticket>(new sqlrs_ticket());
    t->ctx = ctx;
    if (!j->empty_build && !sa_probe_try(j, right, t.get())) {
      sa_flush(ctx); // (tickets complete in issue order)
      InBatch ib(ctx, right);
      DBatch r = probe_batch(j, ib, nullptr);
      t->done = emit_batch(ctx, std::move(r), SQLRS_MEM_HOST);
    }
    *ticket = t.release();
  });
}
} // extern "C"

namespace sq {
// cut[i] = pairs whose probe row (right[], ascending: pairs are probe-row major) lies before bounds[i]
__global__ void pairs_before_kernel(const uint32_t *__restrict__ right, int64_t m, const int64_t *__restrict__ bounds, int64_t n,
                                    int64_t *__restrict__ cut) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t b = bounds[i];
  int64_t lo = 0, hi = m; // first pair with right >= b
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)right[mid] < b) lo = mid + 1;
    else hi = mid;
  }
  cut[i] = lo;
}
} // namespace sq

extern "C" {

// n probe batches in one call: out[i] is what sqlrs_hash_join_probe_push(right[i]) returns [ref: hash_join.rs:207-292: one
// joined batch per probe batch] — for the reference's batch shape, 1024-row HOST batches (storage/csv.rs:105), where one
// upload + probe + download per batch is ~80 us a call (12 Mrows/s).  Inner / Left joins without a join filter whose
// batches are small HOST batches of fixed-width columns (and whose joined columns are fixed width): the batches are
// uploaded together and probed as ONE batch — the pairs come out probe-row major (hash_join.rs:225-234), so input batch
// i's joined rows are one contiguous range, found by searching the pairs' probe rows for the batch boundaries — and
// every range is handed out as a HOST batch of its own.  Anything else runs batch by batch.
int sqlrs_hash_join_probe_push_many(sqlrs_hash_join_t *j, int n, const sqlrs_batch_t *const *right, int out_mem,
                                    sqlrs_batch_t **out) {
  return guard(j->ctx, [&] {
    Ctx *ctx = j->ctx;
    SQ_HIP(hipSetDevice(ctx->device));
    if (!j->finished) fail(SQLRS_ERR_INTERNAL, "probe before build_finish");
    for (int i = 0; i < n; i++) out[i] = nullptr;
    if (n <= 0 || j->empty_build) return;
    if (!j->probe_stage) {
      j->probe_stage.reset(new HostStage());
      j->probe_stage->ctx = ctx;
    }
    HostStage &st = *j->probe_stage;
    bool stageable = out_mem == SQLRS_MEM_HOST && n > 1 && !j->has_filter && !st.has_schema &&
                     (j->join_type == SQLRS_JOIN_INNER || j->join_type == SQLRS_JOIN_LEFT) && all_fixed_width(j->left);
    int64_t total_rows = 0;
    for (int i = 0; i < n && stageable; i++) {
      stageable = st.accepts(right[i]) && right[i]->num_columns == right[0]->num_columns;
      for (int c = 0; c < right[i]->num_columns && stageable; c++) stageable = right[i]->columns[c].dtype == right[0]->columns[c].dtype;
      total_rows += right[i] ? right[i]->num_rows : 0;
    }
    if (!stageable || total_rows == 0 || total_rows > (1ll << 30)) {
      int i = 0;
      try {
        for (; i < n; i++) {
          InBatch ib(ctx, right[i]);
          out[i] = emit_batch(ctx, probe_batch(j, ib, nullptr), out_mem);
        }
      } catch (...) {
        for (int k = 0; k < i; k++) {
          sqlrs_batch_release(out[k]);
          out[k] = nullptr;
        }
        throw;
      }
      return;
    }
    std::vector<int64_t> bounds((size_t)n + 1, 0);
    try {
      for (int i = 0; i < n; i++) {
        st.append(right[i]);
        bounds[(size_t)i + 1] = bounds[(size_t)i] + right[i]->num_rows;
      }
    } catch (...) {
      j->probe_stage.reset(); // (a half-staged call must not leave its schema and rows behind: the staged path would stay off)
      throw;
    }
    sqlrs_batch_t *dev = st.take();
    struct Rel {
      sqlrs_batch_t *b;
      ~Rel() { if (b) sqlrs_batch_release(b); }
    } rel{dev};
    Pairs p;
    DBatch o;
    {
      InBatch ib(ctx, dev);
      o = probe_batch(j, ib, nullptr, &p);
    }
    std::vector<int64_t> cut = bounds; // every probe row matched exactly once: pair i = probe row i
    if (!p.right_identity) {
      BufP dbounds = ctx->alloc(8 * ((size_t)n + 1)), dcut = ctx->alloc(8 * ((size_t)n + 1));
      SQ_HIP(hipMemcpyAsync(dbounds->p, bounds.data(), 8 * ((size_t)n + 1), hipMemcpyHostToDevice, ctx->stream));
      pairs_before_kernel<<<dim3((unsigned)ceil_div(n + 1, 256)), dim3(256), 0, ctx->stream>>>(
          p.right->as<uint32_t>(), p.m, dbounds->as<int64_t>(), n + 1, dcut->as<int64_t>());
      SQ_HIP(hipGetLastError());
      SQ_HIP(hipMemcpyAsync(cut.data(), dcut->p, 8 * ((size_t)n + 1), hipMemcpyDeviceToHost, ctx->stream));
      ctx->sync();
    }
    if (!all_fixed_width(o)) fail(SQLRS_ERR_INTERNAL, "probe_push_many: joined columns are fixed width");
    split_rows_to_host(ctx, o, cut, &j->pin_out, &j->pin_cap, n, out);
  });
}

int sqlrs_hash_join_probe_indices(sqlrs_hash_join_t *j, const sqlrs_batch_t *right, int out_mem,
                                  sqlrs_batch_t **out) {
  return guard(j->ctx, [&] {
    SQ_HIP(hipSetDevice(j->ctx->device));
    if (!j->finished) fail(SQLRS_ERR_INTERNAL, "probe before build_finish");
    *out = nullptr;
    if (j->empty_build) return;
    InBatch ib(j->ctx, right);
    Pairs p;
    probe_batch(j, ib, &p);
    DBatch b;
    b.rows = p.m;
    DCol l, r;
    l.dtype = SQLRS_UINT64;
    l.length = p.m;
    l.values = p.left->p;
    l.own_values = p.left;
    if (p.left_validity) {
      l.validity = p.left_validity->as<uint64_t>();
      l.own_validity = p.left_validity;
      l.null_count = -1;
    }
    r.dtype = SQLRS_UINT32;
    r.length = p.m;
    r.values = p.right->p;
    r.own_values = p.right;
    b.cols.push_back(std::move(l));
    b.cols.push_back(std::move(r));
    *out = emit_batch(j->ctx, std::move(b), out_mem);
  });
}

int sqlrs_hash_join_finish(sqlrs_hash_join_t *j, int out_mem, sqlrs_batch_t **out) {
  return guard(j->ctx, [&] {
    SQ_HIP(hipSetDevice(j->ctx->device));
    if (!j->finished) fail(SQLRS_ERR_INTERNAL, "finish before build_finish");
    *out = nullptr;
    if (j->empty_build) return;
    if (j->join_type != SQLRS_JOIN_LEFT && j->join_type != SQLRS_JOIN_FULL) return;
    Ctx *ctx = j->ctx;
    Selection sel = selection_from_clear_bits(ctx, j->visited->as<uint64_t>(), j->nB); // :298-301
    DBatch b;
    b.rows = sel.count;
    for (const DCol &c : j->left.cols) b.cols.push_back(compact_column(ctx, c, sel));
    for (int32_t dt : j->right_dtypes) b.cols.push_back(make_null_column(ctx, dt, sel.count));
    *out = emit_batch(ctx, std::move(b), out_mem);
  });
}

void sqlrs_hash_join_destroy(sqlrs_hash_join_t *j) {
  if (j && j->async_ordered) sa_drain(j->ctx); // (probe kernels of the async path may still read the table on a side stream)
  delete j;
}

} // extern "C"
