// join_lds.hip — HashJoinExecutor, general keys on LDS bucket tables (hash_join.rs:172-177, 207-292): the blocked partitioned join
// of join.hip's probe for build sides beyond an L2-resident table.  Kernels (partition of the probe rows range by range, per-bucket
// LDS probe, un-permute + compaction per range, un-permute into {run, pairs} for duplicate keys and outer joins, uniqueness of the
// build keys on the same tables, the distinct keys of the general table) and the host side that prepares a build side for the
// route and matches one probe batch (lds_build_first, lds_join_match, lds_join_restore, lds_join_unpermute: join_state.hpp).
// Split from join.hip in round 6 (review r05: files at the edge of reviewability); nothing here is a new design.
#include <cstdlib>

#include "common.hpp"
#include "device_utils.hpp"
#include "prims.hpp"
#include "join_kernels.hpp"
#include "join_state.hpp"

namespace sq {

// ---- general keys on LDS tables (blocked partitioned join) -------------------------------------------------
// For build sides too large for an L2-resident table (sparse 64-bit keys: 16-byte slots at load <= 2/3 leave one
// XCD's 4 MiB L2 at ~1.3e5 keys) the probe of the global table runs at the random-access rate of the memory
// system behind L2 (56-66 G lookups/s: 2.6 ms per 1e8 probe rows against 1e6 build keys).  Radix-partitioned
// instead (north_star: LDS-staged open addressing for the partitioned hash join):
//   build  : the build keys hash-partitioned into P = 512 buckets once (partition_rows, (key, build row) per bucket);
//   probe 1: the probe keys cut into RANGES of 2^15 consecutive rows, every range hash-partitioned into the same
//            512 buckets by one workgroup in one pass (lds_join_partition_kernel: (key, row) per (range, bucket),
//            "slivers" of ~64 rows);
//   probe 2: one workgroup per (bucket, group of ranges): the bucket's build keys go into an open-addressing table
//            in LDS (slot claimed with one 32-bit ds CAS on the row word; the keys are unique, so an insert never
//            compares keys), then every sliver of the bucket is probed there, a wave per sliver, and the build row
//            (or "none") is stored AT THE ROW'S POSITION IN THE PARTITIONED ORDER: coalesced loads and stores only;
//   probe 3: one workgroup per range: its 32768 (row, match) entries are contiguous in the partitioned order; they
//            are un-permuted through a 128 KiB match array in LDS (scattered ds writes) and compacted from there
//            into (left_idx, right_idx) pairs in probe-row order — the reference's pair order
//            (hash_join.rs:225-234) — with the decoupled look-back every compaction here uses.
// The output order is restored by BLOCKING (a range is what one LDS array holds), not by re-sorting: a global
// scatter of 1e8 row ids runs at the random-store rate (89 G/s), and even confined to 4 MiB windows of a global
// match array it cost 2.2 ms (measured: partial-line write-backs), more than the direct probe it replaces.
// 1024-thread workgroups (16 waves share one table, two workgroups per CU): the sliver loop is a chain of dependent
// latencies (sliver bounds -> keys -> LDS probes -> store), so it runs at the number of waves in flight — with
// 256-thread workgroups (12 waves per CU next to three 48 KiB tables) the kernel took 1.9 ms, 1.0 of it without any
// lookup at all.  16-byte slots {key, build row + 1}: one ds_read_b128 per probe step instead of two dependent reads.
constexpr int LJ_WG = 1024, LJ_WAVES = LJ_WG / 64, LJ_RANGE_LOG2 = 15, LJ_RANGE = 1 << LJ_RANGE_LOG2;
#ifndef LJ_Q_N
#define LJ_Q_N 8
#endif
constexpr int LJ_Q = LJ_Q_N; // slivers a wave probes at a time
#ifndef LJ_DBG
#define LJ_DBG 0
#endif
struct LjSlot {
  uint64_t key;
  uint32_t row1, pad; // build row + 1, 0 = empty
};
__device__ __forceinline__ uint32_t lj_bucket(uint64_t key, uint32_t P) { // = radix_part's rp_bucket (the build side's)
  return (uint32_t)__umul64hi(mix64(key), (uint64_t)P);
}

// probe 1: one workgroup partitions ONE range of 2^15 consecutive probe rows into the P buckets, in one pass.
// A range's rows occupy exactly rows [range base, + len) of the partitioned order, so nothing depends on another
// range: no global histogram, no scan, no second read of the keys (a counting multi-split over all ranges cost
// hist 0.14 + scan 0.06 + scatter 0.84 ms per 1e8 rows — 12-row runs per (tile, bucket)).  The 32 keys of a thread
// stay in registers; an LDS atomic per row gives its rank inside its bucket, a scan of the P counters the bucket
// starts (also the sliver table the next two kernels read), and the rows leave through an LDS staging area a
// quarter of the range at a time, so the stores are contiguous (12 B per row: key + original row).
// 1.10 ms per 1e8 rows — not at bandwidth (2.0 GB): at 1024 threads (128 VGPRs) the compiler spills 23 of the 32 keys
// right behind their loads, and one workgroup per CU overlaps nothing.  A second form that STREAMS the keys three
// times instead (ranks, positions, staged output; 512-thread workgroups, 80 VGPRs, three per CU) was built and
// measured at 1.16 ms: its 64 + 64 + 8 x 64 row steps per thread are a chain of dependent L2 latencies.  (Lesson kept
// from it: the unrolled phases share their 64 row addresses, which the compiler keeps live from one phase to the next
// — or hoists out of the staging loop — and spills; an opaque `asm volatile("" : "+v"(tid))` per phase / per group
// brought 628 bytes of scratch per lane down to 60.)
constexpr int LP_ROWS = LJ_RANGE / LJ_WG; // 32 rows per thread
constexpr int LP_STAGE = LJ_RANGE / 4;    // rows staged per round (96 KiB)
__global__ __launch_bounds__(LJ_WG) void lds_join_partition_kernel(const uint64_t *__restrict__ keys, int64_t n, uint32_t P,
                                                                   uint64_t *__restrict__ okey, uint16_t *__restrict__ oidx,
                                                                   uint32_t *__restrict__ pbs, uint32_t nrs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lp_smem[];
  uint64_t *skey = (uint64_t *)lp_smem;            // [LP_STAGE]
  uint16_t *sidx = (uint16_t *)(skey + LP_STAGE);  // [LP_STAGE] row inside the range (< 2^15: 16 bits; round 6 — 32-bit row ids before)
  __shared__ uint32_t cnt[512], start[512 + 1];
  __shared__ uint32_t s_wsum[8];
  const int64_t rbase = (int64_t)blockIdx.x * LJ_RANGE;
  const uint32_t len = (uint32_t)min<int64_t>(LJ_RANGE, n - rbase);
  uint64_t k[LP_ROWS];
  uint32_t rk2[LP_ROWS / 2]; // two 16-bit values per word: the row's rank inside its bucket, then its position inside
                             // the range's output (0xffff = no row); the bucket is recomputed from the key (registers)
#pragma unroll
  for (int j = 0; j < LP_ROWS; j++) // unconditional loads (rows past the end re-read the last row)
    k[j] = __builtin_nontemporal_load(keys + rbase + min((uint32_t)(j * LJ_WG) + threadIdx.x, len - 1));
  if (threadIdx.x < 512) cnt[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < LP_ROWS; j++) {
    uint32_t r = 0xffffu;
    if ((uint32_t)(j * LJ_WG) + threadIdx.x < len) r = atomicAdd(&cnt[lj_bucket(k[j], P)], 1u); // (< 2^15: the range's rows)
    rk2[j >> 1] = (j & 1) ? (rk2[j >> 1] | (r << 16)) : r;
  }
  __syncthreads();
  if (threadIdx.x < 512) { // exclusive scan of the P <= 512 counters (8 waves)
    const uint32_t c = threadIdx.x < P ? cnt[threadIdx.x] : 0;
    const uint32_t inc = wave_iscan_u32(c);
    if (lane_id() == 63) s_wsum[wave_id()] = inc;
    cnt[threadIdx.x] = inc - c; // (wave-local exclusive prefix)
  }
  __syncthreads();
  if (threadIdx.x < 512) {
    uint32_t wb = 0;
    for (int w = 0; w < wave_id(); w++) wb += s_wsum[w];
    const uint32_t st = cnt[threadIdx.x] + wb;
    start[threadIdx.x] = st;
    // (bucket-major: the probe pass of bucket b reads the starts of 64 consecutive ranges with one coalesced load)
    if (threadIdx.x < P) pbs[(size_t)threadIdx.x * nrs + blockIdx.x] = (uint32_t)rbase + st;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < LP_ROWS; j++) {
    const uint32_t r = (rk2[j >> 1] >> ((j & 1) * 16)) & 0xffffu;
    const uint32_t pos = r == 0xffffu ? 0xffffu : start[lj_bucket(k[j], P)] + r; // (< 2^15)
    rk2[j >> 1] = (j & 1) ? ((rk2[j >> 1] & 0xffffu) | (pos << 16)) : ((rk2[j >> 1] & 0xffff0000u) | pos);
  }
  for (uint32_t q0 = 0; q0 < len; q0 += LP_STAGE) { // (uniform trip count)
    // (thread id and packed positions through an opaque asm: the 32 row ids and the 32 unpacked positions of the
    //  unrolled body are loop invariants otherwise, get hoisted out of this loop and push the keys into scratch)
    uint32_t tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
#pragma unroll
    for (int j = 0; j < LP_ROWS; j++) {
      uint32_t w = rk2[j >> 1];
      asm volatile("" : "+v"(w));
      const uint32_t p = ((w >> ((j & 1) * 16)) & 0xffffu) - q0; // (0xffff - q0 stays >= LP_STAGE)
      if (p < (uint32_t)LP_STAGE) {
        skey[p] = k[j];
        sidx[p] = (uint16_t)((uint32_t)(j * LJ_WG) + tid);
      }
    }
    __syncthreads();
    const uint32_t m = min((uint32_t)LP_STAGE, len - q0);
    for (uint32_t p = threadIdx.x; p < m; p += LJ_WG) __builtin_nontemporal_store(skey[p], okey + rbase + q0 + p);
    for (uint32_t p = 2 * threadIdx.x; p < m; p += 2 * LJ_WG) { // (two 16-bit row numbers per store; rbase + q0 is even)
      const uint32_t w2 = (uint32_t)sidx[p] | ((p + 1 < m ? (uint32_t)sidx[p + 1] : 0u) << 16);
      if (p + 1 < m) __builtin_nontemporal_store(w2, (uint32_t *)(oidx + rbase + q0 + p));
      else oidx[rbase + q0 + p] = (uint16_t)w2;
    }
    __syncthreads();
  }
}
template <int Q> // slivers a wave probes per trip
__global__ __launch_bounds__(LJ_WG) void lds_join_probe_kernel(
    const uint64_t *__restrict__ bkey, const uint32_t *__restrict__ brow, const uint32_t *__restrict__ bbstart,
    const uint64_t *__restrict__ pkey, const uint32_t *__restrict__ pbs, uint32_t nrs, uint32_t pn, uint32_t P, uint32_t nranges,
    uint32_t ranges_per_item, uint32_t slots, uint32_t *__restrict__ mpart) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lj_smem[];
  LjSlot *tab = (LjSlot *)lj_smem;
  const uint32_t b = blockIdx.x % P, g = blockIdx.x / P, mask = slots - 1;
  for (uint32_t s = threadIdx.x; s < slots; s += LJ_WG) tab[s].row1 = 0;
  __syncthreads();
  {
    const uint32_t b0 = bbstart[b], b1 = bbstart[b + 1];
    constexpr int BU = 4; // build rows in flight per lane
    for (uint32_t base = b0 + threadIdx.x; base < b1; base += LJ_WG * BU) {
      uint64_t bk[BU];
      uint32_t br[BU];
#pragma unroll
      for (int u = 0; u < BU; u++) {
        const uint32_t i = min(base + u * LJ_WG, b1 - 1);
        bk[u] = bkey[i];
        br[u] = brow[i];
      }
#pragma unroll
      for (int u = 0; u < BU; u++) {
        if (base + u * LJ_WG >= b1) continue;
        uint32_t s = (uint32_t)mix64(bk[u]) & mask; // (the bucket was chosen by the HIGH bits of mix64)
        while (atomicCAS(&tab[s].row1, 0u, br[u] + 1u) != 0u) s = (s + 1) & mask; // (load <= 1/2: terminates)
        tab[s].key = bk[u];
      }
    }
  }
  __syncthreads();
  auto lookup = [&](uint64_t key) {
    uint32_t s = (uint32_t)mix64(key) & mask;
    while (true) {
      const uint4 sl = *(const uint4 *)&tab[s]; // one 16-byte LDS read: key + row word
      if (!sl.z) return DENSE_EMPTY;
      if ((((uint64_t)sl.y << 32) | sl.x) == key) return sl.z - 1;
      s = (s + 1) & mask;
    }
  };
  const int lane = lane_id();
  const uint32_t r1 = min(nranges, (g + 1) * ranges_per_item);
  // Round 6.  A wave takes the bucket's slivers of 64 CONSECUTIVE ranges per round: their bounds arrive with two coalesced
  // loads (bucket-major starts; before: two dependent 4-byte loads per sliver, each its own cache line) and are handed out
  // with v_readlane.  The slivers are probed Q at a time, software-pipelined: the keys of trip t + 1 (first 128 rows of
  // each sliver, 2 Q loads per lane) are in flight while trip t is looked up and stored — the loop was a chain of dependent
  // latencies (bounds -> keys -> LDS probes -> stores) on 16 waves per CU.  The few slivers longer than 128 rows finish in
  // a tail loop.
  for (uint32_t rb = g * ranges_per_item + 64 * wave_id(); rb < r1; rb += 64 * LJ_WAVES) {
    const uint32_t ns = min(64u, r1 - rb); // (uniform)
    const uint32_t rr = rb + min((uint32_t)lane, ns - 1);
    const uint32_t lo_l = pbs[(size_t)b * nrs + rr];
    const uint32_t hi_l = b + 1 < P ? pbs[(size_t)(b + 1) * nrs + rr] : (uint32_t)min<uint64_t>((uint64_t)(rr + 1) << LJ_RANGE_LOG2, pn);
    auto bounds = [&](uint32_t sv, uint32_t &lo, uint32_t &hi) { // sliver sv of the round (uniform); past the end: empty
      const uint32_t sc = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(sv, ns - 1));
      lo = (uint32_t)__builtin_amdgcn_readlane((int)lo_l, (int)sc);
      hi = sv < ns ? (uint32_t)__builtin_amdgcn_readlane((int)hi_l, (int)sc) : lo;
    };
    auto load = [&](uint32_t c, uint64_t(&k)[2 * Q]) {
#pragma unroll
      for (int q = 0; q < Q; q++) {
        uint32_t lo, hi;
        bounds(c * Q + q, lo, hi);
#pragma unroll
        for (int h = 0; h < 2; h++) { // unconditional loads: lanes past the sliver re-read its first row (or row 0)
          const uint32_t i = lo + h * 64 + lane;
          k[2 * q + h] = __builtin_nontemporal_load(pkey + (i < hi ? i : min(lo, pn - 1)));
        }
      }
    };
    // (the lookups of a trip batched — all first slot reads in flight, unresolved keys advancing one step per round — were
    //  measured SLOWER, 0.73 against 0.54 ms: the resolved keys' re-reads and the registers cost more than the chain saves)
    auto work = [&](uint32_t c, const uint64_t(&k)[2 * Q]) {
#pragma unroll
      for (int q = 0; q < Q; q++) {
        uint32_t lo, hi;
        bounds(c * Q + q, lo, hi);
        // (round 6, measured and dropped: the sliver's two 64-row halves looked up in lockstep, two probe sequences per lane in
        //  flight — 0.63 against 0.535 ms; neither that nor the batched form beats one sequence at a time)
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const uint32_t i = lo + h * 64 + lane;
#if LJ_DBG == 1 // (measurement: no LDS lookups)
          if (i < hi) __builtin_nontemporal_store((uint32_t)k[2 * q + h], mpart + i);
#elif LJ_DBG == 2 // (measurement: no stores)
          if (i < hi && lookup(k[2 * q + h]) == 12345u) mpart[i] = 0;
#else
          if (i < hi) __builtin_nontemporal_store(lookup(k[2 * q + h]), mpart + i);
#endif
        }
        for (uint32_t i = lo + 128 + lane; i < hi; i += 64) mpart[i] = lookup(pkey[i]);
      }
    };
    const uint32_t nc = (ns + Q - 1) / Q;
    uint64_t ka[2 * Q], kb[2 * Q];
    load(0, ka);
    for (uint32_t c = 0; c < nc; c += 2) {
      if (c + 1 < nc) load(c + 1, kb);
      work(c, ka);
      if (c + 2 < nc) load(c + 2, ka);
      if (c + 1 < nc) work(c + 1, kb);
    }
  }
}

// Uniqueness of the build keys, bucket by bucket, on the same LDS tables (round 6): a build side that takes this route
// never needs the global 16-byte-slot table (32 MiB memset + 1e6 random CAS inserts = 0.13 ms for 1e6 keys, a twelfth of
// the C3 sparse-key join) — but `unique` was a by-product of building it.  One workgroup per bucket inserts the bucket's
// keys exactly as the probe kernel does and then looks every key up again: the FIRST slot of a key's probe sequence that
// holds an equal key is the same for all rows that carry the key, so of two rows with one key at least one finds a row
// other than itself.
__global__ __launch_bounds__(LJ_WG) void lds_join_unique_kernel(const uint64_t *__restrict__ bkey, const uint32_t *__restrict__ brow,
                                                                const uint32_t *__restrict__ bbstart, uint32_t slots,
                                                                unsigned int *__restrict__ dup) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lu_smem[];
  LjSlot *tab = (LjSlot *)lu_smem;
  const uint32_t b = blockIdx.x, mask = slots - 1;
  for (uint32_t s = threadIdx.x; s < slots; s += LJ_WG) tab[s].row1 = 0;
  __syncthreads();
  const uint32_t b0 = bbstart[b], b1 = bbstart[b + 1];
  for (uint32_t i = b0 + threadIdx.x; i < b1; i += LJ_WG) {
    const uint64_t k = bkey[i];
    uint32_t s = (uint32_t)mix64(k) & mask;
    while (atomicCAS(&tab[s].row1, 0u, brow[i] + 1u) != 0u) s = (s + 1) & mask;
    tab[s].key = k;
  }
  __syncthreads();
  bool bad = false;
  for (uint32_t i = b0 + threadIdx.x; i < b1; i += LJ_WG) {
    const uint64_t k = bkey[i];
    uint32_t s = (uint32_t)mix64(k) & mask;
    while (!(tab[s].row1 && tab[s].key == k)) s = (s + 1) & mask; // (the key is in the table: terminates)
    bad |= tab[s].row1 != brow[i] + 1u;
  }
  if (__ballot(bad) && lane_id() == 0) atomicOr(dup, 1u);
}

// probe 3: un-permute one range through LDS and compact it (see above).  8 worker waves + the scan wave.
constexpr int LR_WAVES = 8, LR_BLOCK = (LR_WAVES + 1) * 64, LR_PER_WAVE = LJ_RANGE / LR_WAVES / 64; // 64 chunks of 64 rows per wave
__global__ __launch_bounds__(LR_BLOCK) void lds_join_restore_kernel(
    const uint16_t *__restrict__ pidx, const uint32_t *__restrict__ mpart, int64_t n, int64_t num_tiles,
    uint64_t *__restrict__ left_idx, uint32_t *__restrict__ right_idx, uint64_t *desc, unsigned *ticket, uint64_t *total,
    int use_ticket) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lr_smem[];
  uint32_t *m = (uint32_t *)lr_smem; // [LJ_RANGE]
  unsigned *timeout = use_ticket ? nullptr : ticket + 1;
  __shared__ int64_t s_tile;
  __shared__ uint32_t s_wave[LR_WAVES];
  __shared__ uint64_t s_excl;
  int64_t tile = blockIdx.x;
  if (use_ticket) {
    if (threadIdx.x == 0) s_tile = (int64_t)atomicAdd(ticket, 1u);
    __syncthreads();
    tile = s_tile;
  }
  const int lane = lane_id(), w = wave_id();
  const int64_t rbase = tile * LJ_RANGE;
  const uint32_t len = (uint32_t)min<int64_t>(LJ_RANGE, n - rbase);
  if (w < LR_WAVES) { // the range's entries are rows [rbase, rbase + len) of the partitioned order, in bucket order
    constexpr int U = 8;
    for (uint32_t base = w * 64 + lane; base < len; base += LR_WAVES * 64 * U) {
      uint32_t id[U], mv[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t i = min(base + u * LR_WAVES * 64, len - 1);
        id[u] = __builtin_nontemporal_load(pidx + rbase + i);
        mv[u] = __builtin_nontemporal_load(mpart + rbase + i);
      }
#pragma unroll
      for (int u = 0; u < U; u++)
        if (base + u * LR_WAVES * 64 < len) m[id[u]] = mv[u];
    }
  }
  __syncthreads(); // (0) the match array of the range is complete
  if (w == LR_WAVES) { // ---- scan wave
    __syncthreads(); // (1) the workers' counts are in s_wave
    uint32_t c = lane < LR_WAVES ? s_wave[lane] : 0;
    uint64_t agg = wave_sum_u32(c);
    uint64_t excl = lookback_wave(desc, tile, agg, timeout);
    if (lane == 0) {
      s_excl = excl;
      if (tile == num_tiles - 1) *total = excl + agg;
    }
    __syncthreads(); // (2)
    return;
  }
  const uint32_t wbase = w * (LR_PER_WAVE * 64);
  uint64_t mine = 0; // lane j keeps the hit mask of chunk j
  uint32_t wave_cnt = 0;
#pragma unroll 8
  for (int j = 0; j < LR_PER_WAVE; j++) {
    const uint32_t r = wbase + j * 64 + lane;
    const uint64_t bm = __ballot(r < len && m[r] != DENSE_EMPTY);
    mine = (lane == j) ? bm : mine;
    wave_cnt += (uint32_t)__popcll(bm);
  }
  if (lane == 0) s_wave[w] = wave_cnt;
  __syncthreads(); // (1)
  __syncthreads(); // (2) the scan wave has published the range's offset
  uint64_t pos = s_excl;
  for (int q = 0; q < w; q++) pos += s_wave[q];
  const uint32_t mlo = (uint32_t)mine, mhi = (uint32_t)(mine >> 32);
#pragma unroll 8
  for (int j = 0; j < LR_PER_WAVE; j++) {
    const uint64_t bm = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)mhi, j) << 32) |
                        (uint32_t)__builtin_amdgcn_readlane((int)mlo, j);
    const uint32_t r = wbase + j * 64 + lane;
    if ((bm >> lane) & 1) {
      const uint64_t o = pos + mbcnt(bm);
      __builtin_nontemporal_store((uint64_t)m[r], &left_idx[o]);
      __builtin_nontemporal_store((uint32_t)(rbase + r), &right_idx[o]);
    }
    pos += (uint32_t)__popcll(bm);
  }
}

// probe 3 for everything that is not "unique build keys, Inner / Left" (round 6): duplicate build keys (every probe row emits
// the RUN of build rows that carry its key, hash_join.rs:225-234) and Right / Full joins (an unmatched probe row emits (NULL,
// row), :235-248).  The range is un-permuted through LDS as above, but nothing is compacted: row r of the batch gets
// match[r] = {first entry of its key's run, rows of the run} and counts[r] = the pairs it emits — exactly what
// join_count_kernel leaves after probing the general table at the random-access rate of the memory behind L2 (2.6 ms per 1e8
// probe rows against 0.5 + 0.5 + 0.4 ms for partition + LDS probe + this pass) — and the scan + join_fill_expand_kernel
// go on from there.  `dmatch` null: unique build keys (the matched "row" is the build row, a run of one).
__global__ __launch_bounds__(1024) void lds_join_unpermute_kernel(const uint16_t *__restrict__ pidx, const uint32_t *__restrict__ mpart,
                                                                  int64_t n, const uint2 *__restrict__ dmatch, int outer_right,
                                                                  uint2 *__restrict__ match, uint32_t *__restrict__ counts, int grouped) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lu2_smem[];
  uint32_t *m = (uint32_t *)lu2_smem; // [LJ_RANGE]
  const int64_t rbase = (int64_t)blockIdx.x * LJ_RANGE;
  const uint32_t len = (uint32_t)min<int64_t>(LJ_RANGE, n - rbase);
  constexpr int U = 8;
  for (uint32_t base = threadIdx.x; base < len; base += 1024 * U) {
    uint32_t id[U], mv[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint32_t i = min(base + u * 1024, len - 1);
      id[u] = __builtin_nontemporal_load(pidx + rbase + i);
      mv[u] = __builtin_nontemporal_load(mpart + rbase + i);
    }
#pragma unroll
    for (int u = 0; u < U; u++)
      if (base + u * 1024 < len) m[id[u]] = mv[u];
  }
  __syncthreads();
  for (uint32_t r0 = 0; r0 < len; r0 += 1024) { // (uniform trip count: the group sums are wave reductions)
    const uint32_t r = r0 + threadIdx.x;
    uint32_t c = 0;
    if (r < len) {
      const uint32_t d = m[r];
      uint2 mt = make_uint2(0u, 0u);
      if (d != DENSE_EMPTY) mt = dmatch ? dmatch[d] : make_uint2(d, 1u);
      __builtin_nontemporal_store(((uint64_t)mt.y << 32) | mt.x, (uint64_t *)(match + rbase + r));
      c = (outer_right && mt.y == 0) ? 1u : mt.y;
      if (!grouped) __builtin_nontemporal_store(c, counts + rbase + r);
    }
    if (grouped) {
      const uint32_t wsum = wave_sum_u32(c);
      if (lane_id() == 0 && (r & ~63u) < len) counts[(rbase + r) >> 6] = wsum;
    }
  }
}
// the distinct keys of the general table (non-empty slots; the slot of the key whose value is the table's "empty" word
// included, the NULL keys' slot not: this route takes no NULL keys) with their runs, in no particular order
__global__ __launch_bounds__(1024) void distinct_slots_kernel(const Slot *__restrict__ t, int64_t cap, uint64_t *__restrict__ dkey,
                                                              uint2 *__restrict__ dmatch, unsigned int *__restrict__ counter) {
  __shared__ uint32_t s_w[16], s_base;
  const int64_t i = blockIdx.x * 1024ll + threadIdx.x;
  Slot sl;
  sl.count = 0;
  if (i <= cap + 1 && i != cap) sl = load_slot(&t[i]);
  const bool has = sl.count != 0;
  const uint64_t bm = __ballot(has);
  if (lane_id() == 0) s_w[wave_id()] = (uint32_t)__popcll(bm);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (int w = 0; w < 16; w++) {
      const uint32_t c = s_w[w];
      s_w[w] = tot;
      tot += c;
    }
    s_base = tot ? atomicAdd(counter, tot) : 0u;
  }
  __syncthreads();
  if (has) {
    const uint32_t d = s_base + s_w[wave_id()] + mbcnt(bm);
    dkey[d] = i == cap + 1 ? EMPTY_KEY : sl.key;
    dmatch[d] = make_uint2(sl.head, sl.count);
  }
}

// General keys on LDS tables, probe 1 + 2 (see lds_join_probe_kernel): the probe rows in (range, bucket) order with the
// build row of every row's key (or DENSE_EMPTY) beside them; `ok` = false: route not taken.  Unique build keys without
// NULLs, exactly-compared keys, probe keys without NULLs; the build side large enough that its global table has left
// one XCD's L2 and small enough for 512 LDS tables of <= 8192 slots.
// the build keys in bucket order + the LDS table size of the fullest bucket, once per join (lds_slots = 0: not this route)
constexpr uint32_t LJ_P = 512;
// `keys` (n of them, no NULLs) in bucket order; returns the LDS table size of the fullest bucket (0: the route does not apply)
static uint32_t lds_partition_keys(Ctx *ctx, const uint64_t *keys, int64_t n, PartitionedRows *pr) {
  const uint32_t P = LJ_P;
  PartitionInput bin;
  bin.keys = keys;
  bin.n = n;
  bin.nv = 0;
  bin.build_side = true;
  if (!partition_rows(ctx, bin, P, pr) || pr->P != P || !pr->idx || !pr->key || pr->pack.kbits) return 0;
  uint32_t maxb = 0;
  for (uint32_t bkt = 0; bkt < P; bkt++) maxb = std::max(maxb, pr->bstart_host[bkt + 1] - pr->bstart_host[bkt]);
  // load <= 1/2 in the fullest bucket.  (Round 6 tried <= 0.7, which puts C3's 1e6 build keys — 1953 per bucket, the fullest
  // ~2090 — on 4096-slot tables, two workgroups per CU instead of one: probe pass 0.54 -> 0.81 ms.  A wave's lookup takes as
  // long as the longest probe sequence among its 64 lanes, and that length, not the number of resident waves, sets the pace:
  // without any lookup the pass takes 0.38 ms.  SQLRS_LJ_LOAD = percent, read once per join.)
  const char *ld_e = hook("SQLRS_LJ_LOAD");
  const uint32_t pct = ld_e ? (uint32_t)std::min(90, std::max(10, std::atoi(ld_e))) : 50;
  uint32_t slots = 1024;
  while ((uint64_t)slots * pct < 100ull * maxb) slots <<= 1;
  return slots <= 8192 ? slots : 0; // (8192 x 16 B = 128 KiB: one workgroup per CU)
}
static void lds_join_prepare(sqlrs_hash_join *j) {
  if (j->lds_build) return;
  auto pr = std::make_shared<PartitionedRows>();
  j->lds_slots = 0;
  j->lds_build = pr; // (remembered either way: do not try again)
  j->lds_slots = lds_partition_keys(j->ctx, j->bkeys->as<uint64_t>(), j->nB, pr.get());
}
// ... of a build side with duplicate keys: the distinct keys of its general table (built by now: rows_by_slot holds the runs)
static void lds_join_prepare_distinct(sqlrs_hash_join *j) {
  if (j->lds_distinct) return;
  Ctx *ctx = j->ctx;
  auto pr = std::make_shared<PartitionedRows>();
  j->lds_dslots = 0;
  j->lds_distinct = pr;
  if (!j->table || !j->rows_by_slot) return;
  const int64_t cap = (int64_t)j->mask + 1;
  j->lds_dkeys = ctx->alloc(8 * (size_t)j->nB);
  j->lds_dmatch = ctx->alloc(8 * (size_t)j->nB);
  BufP counter = ctx->alloc_zero(8);
  {
    ProfScope ps(ctx, "join_build_lds_distinct");
    distinct_slots_kernel<<<dim3((unsigned)ceil_div(cap + 2, 1024)), dim3(1024), 0, ctx->stream>>>(
        j->table->as<Slot>(), cap, j->lds_dkeys->as<uint64_t>(), j->lds_dmatch->as<uint2>(), counter->as<unsigned int>());
    SQ_HIP(hipGetLastError());
  }
  const int64_t D = (int64_t)ctx->fetch_value(counter->as<unsigned int>());
  if (D < 2) return;
  j->lds_dslots = lds_partition_keys(ctx, j->lds_dkeys->as<uint64_t>(), D, pr.get());
}
// A build side that will be probed on LDS tables establishes `unique` there (lds_join_unique_kernel) and leaves the global
// table unbuilt (`table_built` stays false: hash_join_ensure_table builds it when a probe cannot take the route — a small
// batch, NULL probe keys, a Right / Full join never get here).  false: not such a build side, or its keys are not unique.
// SQLRS_LDS_FIRST=0 (read per call): the table first, as before round 6.
bool lds_build_first(sqlrs_hash_join *j) {
  Ctx *ctx = j->ctx;
  const char *env_e = hook("SQLRS_LDS_JOIN"), *first_e = hook("SQLRS_LDS_FIRST");
  const int env = env_e ? std::atoi(env_e) : -1;
  const bool outer_right = j->join_type == SQLRS_JOIN_RIGHT || j->join_type == SQLRS_JOIN_FULL;
  if (env == 0 || (first_e && std::atoi(first_e) == 0) || outer_right || j->lazy_table || j->bkeys_validity || !j->bkeys ||
      j->nB < 2 || j->nB > (1ll << 24))
    return false;
  if (j->unique_known && !j->unique) return false; // (the direct-address build has already seen duplicates: nothing to establish)
  if (env != 1 && j->nB < (1 << 18)) return false;
  lds_join_prepare(j);
  if (!j->lds_slots || !j->lds_build->key) return false;
  const PartitionedRows &bp = *j->lds_build;
  BufP dup = ctx->alloc_zero(8);
  {
    ProfScope ps(ctx, "join_build_lds_unique");
    allow_big_lds(ctx, lds_join_unique_kernel, 136 * 1024);
    lds_join_unique_kernel<<<dim3(LJ_P), dim3(LJ_WG), (size_t)j->lds_slots * sizeof(LjSlot), ctx->stream>>>(
        bp.key->as<uint64_t>(), bp.idx->as<uint32_t>(), bp.bstart->as<uint32_t>(), j->lds_slots, dup->as<unsigned int>());
    SQ_HIP(hipGetLastError());
  }
  if (ctx->fetch_value(dup->as<unsigned int>()) != 0) return false; // duplicates: the general table (CSR chains)
  j->unique = true;
  j->unique_known = true;
  j->lds_first = true;
  return true;
}
// `distinct`: the build side has duplicate keys — the tables hold its DISTINCT keys, a match is an index into lds_dmatch
LdsJoinMatch lds_join_match(sqlrs_hash_join *j, const NKeys &pk, bool distinct) {
  Ctx *ctx = j->ctx;
  LdsJoinMatch out;
  const int64_t n = pk.rows, nB = j->nB;
  const char *env_e = hook("SQLRS_LDS_JOIN"); // test / tuning hook, read per call: 0 = never, 1 = whenever the shapes allow
  const int env = env_e ? std::atoi(env_e) : -1;
  // (hashed keys — several key columns, Utf8 — are matched by their 64-bit hash alone, the reference's rule, hash_join.rs:222-232:
  //  the hash IS the key of this route, compared exactly; they carry no validity: a NULL leaves the hash unchanged)
  if (env == 0 || pk.validity || j->bkeys_validity || !j->bkeys || nB < 2 || n > 0xffffffffll) return out;
  if (env != 1 && (nB < (1 << 18) || n < (1 << 22) || n < 8 * nB)) return out;
  if (distinct) lds_join_prepare_distinct(j);
  else lds_join_prepare(j);
  const uint32_t P = LJ_P;
  const uint32_t lds_slots = distinct ? j->lds_dslots : j->lds_slots;
  if (!lds_slots) return out;
  const PartitionedRows &bp = distinct ? *j->lds_distinct : *j->lds_build;
  const uint32_t nranges = (uint32_t)ceil_div(n, LJ_RANGE);
  const uint32_t nrs = (uint32_t)round_up((size_t)nranges, 64) + 64; // row stride of the bucket-major sliver starts
  BufP pkey = ctx->alloc(8 * (size_t)n + 16), pbstart = ctx->alloc(4 * (size_t)nrs * P);
  out.idx = ctx->alloc(2 * (size_t)n + 16);
  {
    ProfScope ps(ctx, "join_partition_lds");
    allow_big_lds(ctx, lds_join_partition_kernel, 112 * 1024);
    lds_join_partition_kernel<<<dim3(nranges), dim3(LJ_WG), (size_t)LP_STAGE * 12, ctx->stream>>>(
        pk.keys->as<uint64_t>(), n, P, pkey->as<uint64_t>(), out.idx->as<uint16_t>(), pbstart->as<uint32_t>(), nrs);
    SQ_HIP(hipGetLastError());
  }
  // ranges per work item: one LDS table build (~2 K inserts) per rpi x ~64 probe rows
  const char *rpi_e = hook("SQLRS_LJ_RPI");
  const uint32_t rpi = rpi_e ? (uint32_t)std::max(1, std::atoi(rpi_e)) : 1024;
  const uint32_t ngroups = (uint32_t)ceil_div((int64_t)nranges, (int64_t)rpi);
  out.mpart = ctx->alloc(4 * (size_t)n + 16);
  {
    ProfScope ps(ctx, "join_probe_lds");
    const size_t lds = (size_t)lds_slots * sizeof(LjSlot);
    allow_big_lds(ctx, lds_join_probe_kernel<LJ_Q>, 136 * 1024);
    lds_join_probe_kernel<LJ_Q><<<dim3(P * ngroups), dim3(LJ_WG), lds, ctx->stream>>>(
        bp.key->as<uint64_t>(), bp.idx->as<uint32_t>(), bp.bstart->as<uint32_t>(), pkey->as<uint64_t>(),
        pbstart->as<uint32_t>(), nrs, (uint32_t)n, P, nranges, rpi, lds_slots, out.mpart->as<uint32_t>());
    SQ_HIP(hipGetLastError());
  }
  out.ok = true;
  return out;
}

// probe 3 of a batch matched by lds_join_match: (left_idx, right_idx) pairs in probe-row order, compacted with the look-back of
// every compaction here (`desc`: `tiles` zeroed words; ticket / total: the caller's, fetched by it)
int64_t lds_join_tiles(int64_t n) { return ceil_div(n, (int64_t)LJ_RANGE); }
void lds_join_restore(Ctx *ctx, const LdsJoinMatch &lm, int64_t n, uint64_t *left_idx, uint32_t *right_idx, uint64_t *desc, unsigned *ticket,
                      uint64_t *total, int use_ticket) {
  const int64_t tiles = lds_join_tiles(n);
  allow_big_lds(ctx, lds_join_restore_kernel, 4 * LJ_RANGE + 1024);
  lds_join_restore_kernel<<<dim3((unsigned)tiles), dim3(LR_BLOCK), 4 * (size_t)LJ_RANGE, ctx->stream>>>(
      lm.idx->as<uint16_t>(), lm.mpart->as<uint32_t>(), n, tiles, left_idx, right_idx, desc, ticket, total, use_ticket);
  SQ_HIP(hipGetLastError());
}
// ... or, for duplicate build keys / Right / Full joins: match[r] = {run, pairs} and the pair counts (per row, or one sum per
// 64-row group when `grouped`) of every probe row, in probe-row order
void lds_join_unpermute(sqlrs_hash_join *j, const LdsJoinMatch &lm, int64_t n, int outer_right, uint2 *match, uint32_t *counts, int grouped) {
  Ctx *ctx = j->ctx;
  allow_big_lds(ctx, lds_join_unpermute_kernel, 4 * LJ_RANGE + 1024);
  lds_join_unpermute_kernel<<<dim3((unsigned)lds_join_tiles(n)), dim3(1024), 4 * (size_t)LJ_RANGE, ctx->stream>>>(
      lm.idx->as<uint16_t>(), lm.mpart->as<uint32_t>(), n, j->unique ? nullptr : j->lds_dmatch->as<uint2>(), outer_right, match, counts, grouped);
  SQ_HIP(hipGetLastError());
}

} // namespace sq
