// radix_part_kernels.hpp — the device side of radix_part.hip (round 6: moved out of the file that plans and launches the levels; one
// translation unit as before): the counting multi-split of one level (histogram + LDS-staged scatter, packed / record / column rows),
// the chunked first level that needs no histogram (claimed chunks, optional fused row filter, level-2 counts on the way), the slim
// 12-byte-row forms of both levels and of the claimed single level, and the small kernels that turn chunk tables into tile lists,
// run lists and bucket starts.  What every kernel does and why is written at the kernel; the levels are described in radix_part.hip.
#pragma once
#include "device_utils.hpp"
#include "radix_part.hpp"

// stores of the partition passes: plain by default; -DRP_NT_STORES = non-temporal (A/B: tools/ab_two_builds.sh)
#ifdef RP_NT_STORES
#define RP_ST(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define RP_ST(ptr, val) (*(ptr) = (val))
#endif

namespace sq {

// rows per thread are a template parameter: tile = 512 * ROWS rows.  Run length (rows per digit
// per tile) matters more than workgroups per CU: 6144-row tiles with one workgroup per CU beat
// 3072- and 4096-row tiles with two (measured on C5, both before and after the kernel was made
// branch-free).

// s_waitcnt immediate for "at most N vector-memory operations outstanding" (gfx9 encoding: vmcnt[3:0] in bits 3:0,
// vmcnt[5:4] in bits 15:14; expcnt and lgkmcnt left at their maxima = not waited for)
constexpr int vmcnt_imm(int n) { return (n & 0xF) | ((n >> 4) << 14) | 0x0F70; }

struct Tile {
  int64_t start;
  uint32_t len;
  uint32_t stride; // tiles of this tile's segment (matrix stride between digits)
  int64_t mat;     // index of (digit 0, this tile) in the count matrix
};

// XCD-aware block -> tile mapping.  Workgroup b runs on XCD b % 8 (observed dispatch order,
// used for speed only).  The runs that neighbouring tiles write into one bucket are adjacent in
// memory; giving every XCD a CONTIGUOUS range of tiles makes those partial cache lines meet in
// one XCD's L2 instead of being written to HBM twice (rocprofv3 WRITE_SIZE showed 2.2x write
// amplification with the identity mapping).
__device__ __forceinline__ uint32_t xcd_tile(uint32_t b, uint32_t nb) {
  uint32_t q = nb >> 3, r = nb & 7, x = b & 7, i = b >> 3;
  return x * q + min(x, r) + i;
}

// global bucket of a row; digit of the current level
__device__ __forceinline__ uint32_t rp_bucket(uint64_t key, bool valid, uint32_t P) {
  return valid ? (uint32_t)__umul64hi(mix64(key), (uint64_t)P) : 0u;
}
// `key` is the key the row carries after packing (packed_clamp / packed_key); the branch is uniform
__device__ __forceinline__ uint32_t rp_bucket(const KeyPack &kp, uint64_t key, bool valid, uint32_t P) {
  if (kp.dense) return (uint32_t)min((key - kp.kmin) >> kp.rbits, (uint64_t)(P - 1));
  return rp_bucket(key, valid, P);
}
__device__ __forceinline__ uint32_t rp_digit(uint32_t bucket, int level, uint32_t p2_bits) {
  return level == 1 ? (bucket >> p2_bits) : (bucket & ((1u << p2_bits) - 1));
}

// PLAIN: no nullable key (no bitmap, no flags column) — every lane loads unconditionally (rows
// past the end of a ragged tile re-read its last row), so the RP_ROWS loads issue back to back.
template <int RP_WG, int RP_ROWS, bool PLAIN>
__global__ __launch_bounds__(RP_WG) void rp_hist_kernel(const uint64_t *__restrict__ keys,
                                                        const uint64_t *__restrict__ key_validity,
                                                        const uint8_t *__restrict__ flags,
                                                        const Tile *__restrict__ tiles, uint32_t P,
                                                        uint32_t p2_bits, int level, uint32_t digits,
                                                        uint32_t *__restrict__ mat, KeyPack kp) {
  __shared__ uint32_t h[512];
  const Tile t = tiles[xcd_tile(blockIdx.x, gridDim.x)];
  uint64_t k[RP_ROWS];
  if (PLAIN) {
#pragma unroll
    for (int j = 0; j < RP_ROWS; j++)
      k[j] = __builtin_nontemporal_load(keys + t.start + min((uint32_t)(j * RP_WG) + threadIdx.x, t.len - 1));
  }
  if (threadIdx.x < digits) h[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < RP_ROWS; j++) {
    uint32_t o = j * RP_WG + threadIdx.x;
    if (o < t.len) {
      int64_t r = t.start + o;
      bool valid = PLAIN ? true : (flags ? (flags[r] & 1) : (!key_validity || ((key_validity[r >> 6] >> (r & 63)) & 1)));
      uint64_t key = PLAIN ? k[j] : keys[r];
      if (kp.kbits) key = level == 1 ? packed_clamp(kp, key) : packed_key(kp, key); // packed partition
      atomicAdd(&h[rp_digit(rp_bucket(kp, key, valid, P), level, p2_bits)], 1u);
    }
  }
  __syncthreads();
  // tile-major: the tile's `digits` counts are one contiguous run (rp_transpose_kernel turns the
  // matrix digit-major for the scan; 256 scattered 4-byte writes per tile were a third of this
  // kernel's time at level 2)
  if (threadIdx.x < digits) mat[(int64_t)xcd_tile(blockIdx.x, gridDim.x) * digits + threadIdx.x] = h[threadIdx.x];
}

// count / offset matrix between its two layouts, 16 tiles at a time through LDS:
//   tile-major  T[g * digits + d]                (what one tile reads or writes: contiguous)
//   digit-major M[tiles[g].mat + d * tiles[g].stride]  (the order the exclusive scan must run in:
//                                                  all tiles of digit 0, then digit 1, ... per segment)
constexpr int RP_TB = 16;
template <bool TO_TILE_MAJOR>
__global__ __launch_bounds__(256) void rp_transpose_kernel(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst,
                                                           const Tile *__restrict__ tiles, uint32_t num_tiles,
                                                           uint32_t digits) {
  extern __shared__ uint32_t tbuf[]; // [RP_TB][digits + 1]
  const uint32_t g0 = blockIdx.x * RP_TB, pitch = digits + 1;
  for (uint32_t i = threadIdx.x; i < RP_TB * digits; i += 256) {
    uint32_t tl, d;
    if (TO_TILE_MAJOR) { d = i / RP_TB; tl = i % RP_TB; } else { tl = i / digits; d = i % digits; }
    uint32_t g = g0 + tl;
    if (g >= num_tiles) continue;
    tbuf[tl * pitch + d] = TO_TILE_MAJOR ? src[tiles[g].mat + (int64_t)d * tiles[g].stride] : src[(int64_t)g * digits + d];
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < RP_TB * digits; i += 256) {
    uint32_t tl, d;
    if (TO_TILE_MAJOR) { tl = i / digits; d = i % digits; } else { d = i / RP_TB; tl = i % RP_TB; }
    uint32_t g = g0 + tl;
    if (g >= num_tiles) continue;
    if (TO_TILE_MAJOR) dst[(int64_t)g * digits + d] = tbuf[tl * pitch + d];
    else dst[tiles[g].mat + (int64_t)d * tiles[g].stride] = tbuf[tl * pitch + d];
  }
}

struct RpIn {
  const uint64_t *key, *v0, *v1;
  const uint32_t *idx;     // null at level 1: row id = position
  const uint8_t *flags;    // null at level 1: built from the bitmaps
  const uint64_t *key_validity, *v0_validity, *v1_validity;
};
struct RpOut {
  uint64_t *key, *v0, *v1;
  uint32_t *idx;
  uint8_t *flags; // null when no column is nullable
  u64x2 *rec = nullptr; // REC kernels: {packed key|row word, value 0} records instead of key / v0
};

// How a level reads its rows (template parameter of the scatter kernel, so the unrolled load
// sequence has no branches: with run-time `if (in.idx)` / bitmap tests inside it the compiler put
// an s_waitcnt vmcnt(0) after every row's loads and the 12 rows of a thread were fetched one
// HBM latency after the other — 20 us per tile, i.e. the whole kernel):
//   RP_L1      level 1, no nullable column: row id = position, flags = 7
//   RP_L1_NULL level 1 with validity bitmaps (flags built from the bitmaps)
//   RP_LN      level >= 2: row id column, no flags column
//   RP_LN_FLAG level >= 2: row id column + flags column
enum { RP_L1 = 0, RP_L1_NULL = 1, RP_LN = 2, RP_LN_FLAG = 3 };

// rows of one tile held in registers (one struct per pipeline stage)
template <int NV, int RP_ROWS> struct TileRegs {
  uint64_t k[RP_ROWS], a0[NV >= 1 ? RP_ROWS : 1], a1[NV >= 2 ? RP_ROWS : 1];
  uint32_t id[RP_ROWS];
  uint8_t fl[RP_ROWS];
  uint32_t goff; // offs[] entry of (digit threadIdx.x, this tile)
};

template <int NV, int RP_WG, int RP_ROWS, int MODE, bool PACK>
__device__ __forceinline__ void rp_load_tile(const RpIn &in, const Tile &t, uint32_t tile_index,
                                             const uint32_t *__restrict__ offs, uint32_t digits,
                                             TileRegs<NV, RP_ROWS> &r) {
  // every lane loads (rows past the end of a ragged tile re-read its last row), so the loads of
  // all RP_ROWS rows are issued back to back
#pragma unroll
  for (int j = 0; j < RP_ROWS; j++) {
    uint32_t o = min((uint32_t)(j * RP_WG) + threadIdx.x, t.len - 1);
    int64_t row = t.start + o;
    r.k[j] = __builtin_nontemporal_load(in.key + row);
    if (NV >= 1) r.a0[j] = __builtin_nontemporal_load(in.v0 + row);
    if (NV >= 2) r.a1[j] = __builtin_nontemporal_load(in.v1 + row);
    if (PACK) r.id[j] = (uint32_t)row; // level 1: packed at staging time; level >= 2: unused
    else if (MODE == RP_LN || MODE == RP_LN_FLAG) r.id[j] = __builtin_nontemporal_load(in.idx + row);
    else r.id[j] = (uint32_t)row;
    if (MODE == RP_LN_FLAG) r.fl[j] = in.flags[row];
    else r.fl[j] = 7;
  }
  r.goff = offs[(int64_t)tile_index * digits + min(threadIdx.x, digits - 1)]; // tile-major offsets
  if (MODE == RP_L1_NULL) {
#pragma unroll
    for (int j = 0; j < RP_ROWS; j++) {
      int64_t row = t.start + min((uint32_t)(j * RP_WG) + threadIdx.x, t.len - 1);
      uint8_t f = 0;
      if (!in.key_validity || ((in.key_validity[row >> 6] >> (row & 63)) & 1)) f |= 1;
      if (!in.v0_validity || ((in.v0_validity[row >> 6] >> (row & 63)) & 1)) f |= 2;
      if (!in.v1_validity || ((in.v1_validity[row >> 6] >> (row & 63)) & 1)) f |= 4;
      r.fl[j] = f;
    }
  }
}

// Persistent workgroups: workgroup b owns the CONTIGUOUS tile range [b*tpw, (b+1)*tpw) (so the
// partial cache lines shared by neighbouring tiles' runs meet in one L2) and software-pipelines
// it: the rows of tile i+1 are loaded into a second register set before tile i goes through its
// LDS phases (rank with LDS atomics -> scan -> stage sorted -> coalesced stores), which hides the
// HBM latency that a 150 KiB-LDS kernel (one workgroup per CU) cannot hide with occupancy.
template <int NV, int RP_WG, int RP_ROWS, int MODE, bool PACK, bool REC = false>
__global__ __launch_bounds__(RP_WG, (RP_ROWS <= 6 || (PACK && RP_ROWS <= 8 && NV <= 1)) ? 2 : 1) void rp_scatter_kernel(RpIn in, RpOut out,
                                                           const Tile *__restrict__ tiles, uint32_t P,
                                                           uint32_t p2_bits, int level, uint32_t digits,
                                                           const uint32_t *__restrict__ offs,
                                                           uint32_t num_tiles, uint32_t tiles_per_wg,
                                                           int64_t sink, KeyPack kp) {
  constexpr int RP_TILE = RP_WG * RP_ROWS;
  constexpr bool FLAGS = MODE == RP_L1_NULL || MODE == RP_LN_FLAG;
#ifdef RP_NO_DRAIN
  constexpr bool DRAIN = false;
#else
  constexpr bool DRAIN = true;
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // staging area: key | v0 | v1 | row id | digit | flags  (packed mode: key|row words and values only,
  // the digit is recomputed from the key when the tile is written out)
  uint64_t *skey = (uint64_t *)smem;
  uint64_t *sv0 = skey + RP_TILE;
  uint64_t *sv1 = sv0 + (NV >= 1 ? RP_TILE : 0);
  uint32_t *sidx = (uint32_t *)(sv1 + (NV >= 2 ? RP_TILE : 0));
  uint16_t *sdig = (uint16_t *)(sidx + (PACK ? 0 : RP_TILE));
  uint8_t *sflag = (uint8_t *)(sdig + (PACK ? 0 : RP_TILE));
  uint32_t *cnt = (uint32_t *)(sflag + (PACK ? 0 : RP_TILE)); // [RP_WG]
  uint32_t *lstart = cnt + RP_WG;                // [RP_WG]
  int64_t *gbase = (int64_t *)(lstart + RP_WG);  // [RP_WG]
  __shared__ uint32_t s_wsum[RP_WG / 64];

  const uint32_t t0 = blockIdx.x * tiles_per_wg;
  const uint32_t t1 = min(num_tiles, t0 + tiles_per_wg);
  if (t0 >= t1) return;
  // Software pipeline over the workgroup's tiles.  While tile i sits sorted in the LDS staging
  // area and is written out, tile i+1 (already in registers) is ranked — the LDS atomics of the
  // rank fill the issue slots the stores leave while the memory pipe pushes back — and tile i+2 is
  // being loaded into the second register set.  Everything that would make the compiler wait
  // early is kept out of the loop (s_waitcnt vmcnt counts loads and stores together on gfx9, and
  // at a join of two paths the compiler assumes the one with FEWER younger operations):
  //  * loads are unconditional — ragged tiles re-read their last row, the last iterations
  //    re-read the last tile;
  //  * stores are unconditional — the lanes past the end of a ragged tile write to the
  //    `sink` rows behind the output columns;
  //  * the first tile is complete before the loop, like every later tile on the back edge.
  // A tile's stores are drained before the barrier that follows them (read / write phases per
  // CU: measured 12.9 -> 10.7 ms per C5 step).
  TileRegs<NV, RP_ROWS> cur, nxt;
  uint32_t dg[RP_ROWS], rk[RP_ROWS];
  auto rank_row = [&](int j, uint32_t len) {
    dg[j] = 0xffffffffu;
    if ((uint32_t)(j * RP_WG) + threadIdx.x < len) {
      const uint64_t key = !PACK ? cur.k[j] : (MODE == RP_LN ? packed_key(kp, cur.k[j]) : packed_clamp(kp, cur.k[j]));
      dg[j] = rp_digit(rp_bucket(kp, key, cur.fl[j] & 1, P), level, p2_bits);
      rk[j] = atomicAdd(&cnt[dg[j]], 1u);
    }
  };
  auto scan_and_stage = [&]() { // counters -> run starts; rows of `cur` -> staging area, sorted by digit
    uint32_t c = cnt[threadIdx.x];
    uint32_t inc = wave_iscan_u32(c);
    if (lane_id() == 63) s_wsum[wave_id()] = inc;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < wave_id(); w++) wbase += s_wsum[w];
    uint32_t ls = wbase + inc - c;
    lstart[threadIdx.x] = ls;
    gbase[threadIdx.x] = (int64_t)cur.goff - (int64_t)ls;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RP_ROWS; j++) {
      if (dg[j] == 0xffffffffu) continue;
      uint32_t p = lstart[dg[j]] + rk[j];
      skey[p] = (PACK && MODE == RP_L1) ? pack_key_row(kp, cur.k[j], cur.id[j]) : cur.k[j];
      if (NV >= 1) sv0[p] = cur.a0[j];
      if (NV >= 2) sv1[p] = cur.a1[j];
      if (!PACK) sidx[p] = cur.id[j];
      if (!PACK) sdig[p] = (uint16_t)dg[j];
      if (FLAGS) sflag[p] = cur.fl[j];
    }
    __syncthreads();
  };
  auto store_row = [&](int j, uint32_t len) { // row j of the staged tile -> its run in the output
    uint32_t p = j * RP_WG + threadIdx.x;
    const uint64_t kw = skey[p];
    const uint32_t d = PACK ? rp_digit(rp_bucket(kp, packed_key(kp, kw), true, P), level, p2_bits) : sdig[p];
    int64_t g = gbase[min(d, (uint32_t)(RP_WG - 1))] + p;
    if (p >= len) g = sink + threadIdx.x;
    if (REC) {
      u64x2 rec;
      rec.x = kw;
      rec.y = sv0[NV >= 1 ? p : 0];
      RP_ST(&out.rec[g], rec);
      return;
    }
    out.key[g] = kw;
    if (NV >= 1) out.v0[g] = sv0[p];
    if (NV >= 2) out.v1[g] = sv1[p];
    if (!PACK) out.idx[g] = sidx[p];
    if (FLAGS) out.flags[g] = sflag[p];
  };

  // first tile: rank, scan, stage
  Tile t = tiles[t0];
  rp_load_tile<NV, RP_WG, RP_ROWS, MODE, PACK>(in, t, t0, offs, digits, cur);
  __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
  cnt[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < RP_ROWS; j++) rank_row(j, t.len);
  __syncthreads();
  scan_and_stage();
  uint32_t staged_len = t.len;
  // second tile into `cur`
  t = tiles[min(t0 + 1, t1 - 1)];
  rp_load_tile<NV, RP_WG, RP_ROWS, MODE, PACK>(in, t, min(t0 + 1, t1 - 1), offs, digits, cur);
#ifndef RP_PREFETCH_LATE
  __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): same entry state as the back edge
#endif
  // loads of one rp_load_tile per thread (the prefetch left in flight behind the drained stores, see below)
  [[maybe_unused]] constexpr int NLOADS = RP_ROWS * (1 + (NV >= 1) + (NV >= 2) + ((!PACK && (MODE == RP_LN || MODE == RP_LN_FLAG)) ? 1 : 0) +
                                    (MODE == RP_LN_FLAG ? 1 : 0)) + 1;
  for (uint32_t ti = t0 + 1; ti < t1; ti++) {
    // staging area: tile ti-1 (sorted);  cur: tile ti;  nxt <- tile ti+1
    const uint32_t tnext = min(ti + 1, t1 - 1);
    Tile tn = tiles[tnext];
#ifndef RP_PREFETCH_LATE // (default; -DRP_PREFETCH_LATE = the A/B variant below, tools/ab_two_builds.sh)
    rp_load_tile<NV, RP_WG, RP_ROWS, MODE, PACK>(in, tn, tnext, offs, digits, nxt);
#endif
    cnt[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RP_ROWS; j++) {
      store_row(j, staged_len);
      rank_row(j, t.len);
    }
    // -DRP_PREFETCH_LATE (measured in round 3, not faster, kept as an A/B build): the next tile's loads issued BEHIND
    // this tile's stores.  vmcnt is one in-order counter: with the loads in front, every wait of the rank loop — and
    // the drain — also waits for prefetched rows it does not need yet (the ISA shows vmcnt(62) .. vmcnt(32) down the
    // unrolled loop: the counter saturates at 63 with 64 loads + stores in flight) and nothing is in flight during
    // scan_and_stage; behind the stores the drain waits for exactly the stores (the NLOADS youngest operations stay
    // outstanding) and the rows of tile ti+1 travel while tile ti is scanned and staged.  Two builds in one process
    // (tools/ab_two_builds.sh), C5: level 1 6.27 / 6.20 / 5.84 ms late vs 6.21 / 5.93 / 5.83 early, level 2 3.80 / 3.18
    // / 3.72 vs 3.04 / 2.91 / 3.70 — inside the spread that buffer placement alone causes: these loops are not bound by
    // where their loads sit.
#ifndef RP_PREFETCH_LATE
    if (DRAIN) __builtin_amdgcn_s_waitcnt(0x0F70);
#else
    rp_load_tile<NV, RP_WG, RP_ROWS, MODE, PACK>(in, tn, tnext, offs, digits, nxt);
    if (DRAIN) __builtin_amdgcn_s_waitcnt(vmcnt_imm(NLOADS < 63 ? NLOADS : 63));
#endif
    __syncthreads(); // the staging area is free, the counters are complete
    scan_and_stage();
    staged_len = t.len;
    cur = nxt;
    t = tn;
  }
  // last staged tile
#pragma unroll
  for (int j = 0; j < RP_ROWS; j++) store_row(j, staged_len);
}

// ------------------------------------------------------------ chunked first level --
// First level of a two-level partition WITHOUT a histogram pass, with an optional row filter fused in.
//
// A counting multi-split needs every (tile, digit) count before the first row can be written: one
// extra read of the keys, and — when a Filter sits below the operator — a filter pass that writes
// compacted copies of every column just so that they can be read again (C5: filter 12 + compact 12 +
// hist 4 + scatter 16 GB).  Here the output of level 1 is not one contiguous run per digit but a list
// of fixed-size CHUNKS per digit (linked-bucket partitioning): workgroup b appends the rows of digit d
// to its own current chunk of that digit and takes a fresh chunk of its own arena when it is
// full, so a row's destination depends on nothing but the workgroup's own history.  One pass: every
// input column read once (the predicate evaluated on the way), every kept row written once.
//
// Level 2 treats every non-empty chunk as one input tile (chunk capacity = its tile size), so only
// its tile list changes (rp_chunk_plan_kernel); its output is contiguous per bucket as before.
//
// Chunk bookkeeping per (workgroup, digit) lives in the registers of thread `digit`: current chunk and fill.
// Every workgroup owns an ARENA of consecutive chunks and hands them out in order from a counter in LDS (digit d
// starts in chunk d of the arena): no global atomic, no overflow possible — a workgroup that reads R rows closes
// at most R / CAP chunks and leaves at most `digits` partly filled, so arena = ceil(tiles_per_wg / tiles per chunk)
// + digits + 1 chunks always suffice — and the pool is half the size of the first scheme (two pre-assigned chunks
// per (workgroup, digit) + a global counter with a pre-fetched next chunk), which it replaced at equal level-1
// time and slightly better level-2 time (its chunks are read in a more regular order).
// Chunks are laid out RP_CHUNK_SKEW rows apart from a multiple of the tile size: with exact multiples of
// 48 KiB every workgroup of the next level starts its tile on one of four phases of the HBM channel
// interleave at the same moment.
constexpr uint32_t RP_CHUNK_SKEW = 32;
struct ChunkOut {
  uint64_t *key, *v0, *v1;
  uint32_t *idx;        // null when the row id is packed into the key word
  uint32_t *chunk_len;  // rows in chunk c (written when it is closed / at the end of its workgroup)
  uint32_t *chunk_dig;  // level-1 digit of chunk c
  unsigned int *counter; // [1] overflow flag ([0] stays 0: the plan kernels scan base_chunks + counter[0] chunks)
  uint32_t base_chunks, max_chunks; // both = workgroups * arena
  uint32_t cap; // rows per chunk: a multiple of the tile size
  uint32_t arena;     // chunks per workgroup: workgroup b takes chunks b * arena, b * arena + 1, ... in this order
  uint32_t *hist = nullptr; // H2 kernels: [chunk][next level's digits] rows of the chunk per digit of the NEXT level
};

// PSRC: where the predicate's operand comes from: -1 no filter, 1 = value column 0, 3 = its own column
template <int NV, int RP_ROWS, int PSRC> struct ChunkRegs {
  uint64_t k[RP_ROWS], a0[NV >= 1 ? RP_ROWS : 1], a1[NV >= 2 ? RP_ROWS : 1], pv[PSRC == 3 ? RP_ROWS : 1];
};

template <int NV, int RP_WG, int RP_ROWS, int PSRC>
__device__ __forceinline__ void rp_chunk_load(const uint64_t *__restrict__ key, const uint64_t *__restrict__ v0,
                                              const uint64_t *__restrict__ v1, const uint64_t *__restrict__ pcol,
                                              int64_t start, uint32_t len, ChunkRegs<NV, RP_ROWS, PSRC> &r) {
#pragma unroll
  for (int j = 0; j < RP_ROWS; j++) { // unconditional: rows past the end of a ragged tile re-read its last row
    const int64_t row = start + min((uint32_t)(j * RP_WG) + threadIdx.x, len - 1);
    r.k[j] = __builtin_nontemporal_load(key + row);
    if (NV >= 1) r.a0[j] = __builtin_nontemporal_load(v0 + row);
    if (NV >= 2) r.a1[j] = __builtin_nontemporal_load(v1 + row);
    if (PSRC == 3) r.pv[j] = __builtin_nontemporal_load(pcol + row);
  }
}

// H2: the kernel also counts, per chunk, its rows per digit of the NEXT level (the level-2 histogram pass read the
// key words of every chunk again for exactly these numbers).  Possible when all P buckets have a counter in LDS:
// h2[bucket] holds two 16-bit counts, low = rows of the digit's CURRENT chunk, high = rows of this tile that spill
// into its next chunk (rank >= room left); when a chunk is closed its low halves go to out.hist and the high halves
// take their place.
template <int NV, int RP_WG, int RP_ROWS, bool PACK, int PSRC, bool H2 = false>
__global__ __launch_bounds__(RP_WG, 1) void rp_chunk_scatter_kernel(
    const uint64_t *__restrict__ key, const uint64_t *__restrict__ v0, const uint64_t *__restrict__ v1, RowFilter flt,
    int64_t n, ChunkOut out, uint32_t P, uint32_t p2_bits, uint32_t digits, uint32_t num_tiles, uint32_t tiles_per_wg,
    int64_t sink, KeyPack kp) {
  constexpr uint32_t RP_TILE = RP_WG * RP_ROWS; // (a chunk holds out.cap = k * RP_TILE rows)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t *skey = (uint64_t *)smem;
  uint64_t *sv0 = skey + RP_TILE;
  uint64_t *sv1 = sv0 + (NV >= 1 ? RP_TILE : 0);
  uint32_t *sidx = (uint32_t *)(sv1 + (NV >= 2 ? RP_TILE : 0));
  uint16_t *sdig = (uint16_t *)(sidx + (PACK ? 0 : RP_TILE));
  uint32_t *cnt = (uint32_t *)(sdig + (PACK ? 0 : RP_TILE)); // [RP_WG]
  uint32_t *split = cnt + RP_WG;                              // [RP_WG] tile-local position where a digit's run changes chunk
  int64_t *gb0 = (int64_t *)(split + RP_WG);                  // [RP_WG] destination of position p: gb0[d] + p below the split,
  int64_t *gb1 = gb0 + RP_WG;                                 //         gb1[d] + p from it on
  uint32_t *room_s = (uint32_t *)(gb1 + RP_WG);               // H2: [RP_WG] rows the digit's current chunk can still take
  uint32_t *close_id = room_s + RP_WG;                        // H2: [RP_WG] chunk the digit closes in this tile, or ~0
  uint32_t *h2 = close_id + RP_WG;                            // H2: [P]
  __shared__ uint32_t s_wsum[RP_WG / 64];
  __shared__ uint32_t s_total;
  __shared__ uint32_t s_next; // next free chunk of this workgroup's arena
  const uint32_t d2n = 1u << p2_bits; // digits of the next level

  const uint32_t t0 = blockIdx.x * tiles_per_wg;
  const uint32_t t1 = min(num_tiles, t0 + tiles_per_wg);
  // chunk state of digit threadIdx.x
  uint32_t cur_id = blockIdx.x * out.arena + min(threadIdx.x, digits - 1), cfill = 0;
  if (threadIdx.x == 0) s_next = digits; // (barriers follow before the first allocation)
  const bool owner = threadIdx.x < digits;
  if (H2) {
    room_s[threadIdx.x] = out.cap;
    for (uint32_t i = threadIdx.x; i < digits * d2n; i += RP_WG) h2[i] = 0;
  }
  auto tile_start = [&](uint32_t t) { return (int64_t)t * RP_TILE; };
  auto tile_len = [&](uint32_t t) { return (uint32_t)min((int64_t)RP_TILE, n - (int64_t)t * RP_TILE); };

  ChunkRegs<NV, RP_ROWS, PSRC> cur, nxt;
  uint32_t dg[RP_ROWS], rk[RP_ROWS];
  int64_t cur_start = 0;
  auto rank_row = [&](int j, uint32_t len) {
    dg[j] = 0xffffffffu;
    bool keep = (uint32_t)(j * RP_WG) + threadIdx.x < len;
    if (PSRC == 1) keep = keep && row_passes(flt, cur.a0[NV >= 1 ? j : 0]);
    if (PSRC == 3) keep = keep && row_passes(flt, cur.pv[PSRC == 3 ? j : 0]);
    if (keep) {
      const uint64_t k = PACK ? packed_clamp(kp, cur.k[j]) : cur.k[j];
      const uint32_t bkt = rp_bucket(kp, k, true, P);
      dg[j] = rp_digit(bkt, 1, p2_bits);
      rk[j] = atomicAdd(&cnt[dg[j]], 1u);
      if (H2) atomicAdd(&h2[bkt], rk[j] < room_s[dg[j]] ? 1u : 0x10000u);
    }
  };
  auto flush_closed = [&]() { // H2: histograms of the chunks named in close_id[] -> out.hist, spill counts move down
    for (uint32_t i = threadIdx.x; i < digits * d2n; i += RP_WG) {
      const uint32_t c = close_id[i >> p2_bits];
      if (c == 0xffffffffu) continue;
      const uint32_t v = h2[i];
      out.hist[(size_t)c * d2n + (i & (d2n - 1))] = v & 0xffffu;
      h2[i] = v >> 16;
    }
  };
  auto scan_and_stage = [&]() { // counters -> tile-local run starts + chunk destinations; rows of `cur` -> staging area
    const uint32_t c = cnt[threadIdx.x];
    const uint32_t inc = wave_iscan_u32(c);
    if (lane_id() == 63) s_wsum[wave_id()] = inc;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
    for (int w = 0; w < RP_WG / 64; w++) {
      if (w < wave_id()) wbase += s_wsum[w];
      tot += s_wsum[w];
    }
    const uint32_t ls = wbase + inc - c;
    if (threadIdx.x == 0) s_total = tot;
    // destination of this digit's run: the rest of the current chunk, then the next chunk of the arena
    const uint32_t room = min(c, out.cap - cfill);
    int64_t g0 = (int64_t)cur_id * (out.cap + RP_CHUNK_SKEW) + cfill, g1 = 0;
    if (H2) close_id[threadIdx.x] = (owner && c > room) ? cur_id : 0xffffffffu;
    if (owner && c > room) {
      out.chunk_len[cur_id] = out.cap; // closed
      out.chunk_dig[cur_id] = threadIdx.x;
      const uint32_t o = atomicAdd(&s_next, 1u); // next chunk of the workgroup's own arena (an LDS counter)
      if (o >= out.arena) out.counter[1] = 1;    // cannot happen (see the bound above); never out of bounds
      cur_id = blockIdx.x * out.arena + min(o, out.arena - 1);
      cfill = c - room;
      g1 = (int64_t)cur_id * (out.cap + RP_CHUNK_SKEW);
    } else {
      cfill += c;
    }
    cnt[threadIdx.x] = ls; // run start (rank_row's counters are consumed)
    split[threadIdx.x] = ls + room;
    gb0[threadIdx.x] = g0 - (int64_t)ls;
    gb1[threadIdx.x] = g1 - (int64_t)(ls + room);
    if (H2) room_s[threadIdx.x] = out.cap - cfill; // what the next tile's rows of this digit may still put into its chunk
    __syncthreads();
    if (H2) flush_closed(); // (this tile's counts are complete, the next tile's start after the barrier below)
#pragma unroll
    for (int j = 0; j < RP_ROWS; j++) {
      if (dg[j] == 0xffffffffu) continue;
      const uint32_t p = cnt[dg[j]] + rk[j];
      const uint32_t row = (uint32_t)(cur_start + (uint32_t)(j * RP_WG) + threadIdx.x);
      skey[p] = PACK ? pack_key_row(kp, cur.k[j], row) : cur.k[j];
      if (NV >= 1) sv0[p] = cur.a0[j];
      if (NV >= 2) sv1[p] = cur.a1[j];
      if (!PACK) sidx[p] = row;
      if (!PACK) sdig[p] = (uint16_t)dg[j];
    }
    __syncthreads();
  };
  auto store_row = [&](int j, uint32_t len) { // position p of the staged tile -> its chunk
    const uint32_t p = j * RP_WG + threadIdx.x;
    const uint64_t kw = skey[p];
    // (clamped, not masked: RP_WG = 768 is no power of two; the clamp only keeps the garbage digit of a position past the staged rows inside the arrays)
    const uint32_t d = min(PACK ? rp_digit(rp_bucket(kp, packed_key(kp, kw), true, P), 1, p2_bits) : (uint32_t)sdig[p], (uint32_t)(RP_WG - 1));
    int64_t g = (p < split[d] ? gb0[d] : gb1[d]) + p;
    if (p >= len) g = sink + (int64_t)blockIdx.x * RP_WG + threadIdx.x; // lanes past the staged rows (boundary slot): this workgroup's sink rows
    RP_ST(&out.key[g], kw);
    if (NV >= 1) RP_ST(&out.v0[g], sv0[p]);
    if (NV >= 2) RP_ST(&out.v1[g], sv1[p]);
    if (!PACK) out.idx[g] = sidx[p];
  };

  if (t0 < t1) {
    // same software pipeline as rp_scatter_kernel: while tile i-1 (staged, sorted) is written out, tile i
    // (in `cur`) is ranked and tile i+1 is in flight into `nxt`
    uint32_t len = tile_len(t0);
    cur_start = tile_start(t0);
    rp_chunk_load<NV, RP_WG, RP_ROWS, PSRC>(key, v0, v1, flt.col, cur_start, len, cur);
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
    cnt[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RP_ROWS; j++) rank_row(j, len);
    __syncthreads();
    scan_and_stage();
    uint32_t staged_len = s_total;
    uint32_t tcur = min(t0 + 1, t1 - 1);
    len = tile_len(tcur);
    cur_start = tile_start(tcur);
    rp_chunk_load<NV, RP_WG, RP_ROWS, PSRC>(key, v0, v1, flt.col, cur_start, len, cur);
#ifndef RP_PREFETCH_LATE
    __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
    [[maybe_unused]] constexpr int NLOADS = RP_ROWS * (1 + (NV >= 1) + (NV >= 2) + (PSRC == 3)); // loads of one rp_chunk_load per thread
    for (uint32_t ti = t0 + 1; ti < t1; ti++) {
      const uint32_t tnext = min(ti + 1, t1 - 1);
      const uint32_t nlen = tile_len(tnext);
      const int64_t nstart = tile_start(tnext);
#ifndef RP_PREFETCH_LATE
      rp_chunk_load<NV, RP_WG, RP_ROWS, PSRC>(key, v0, v1, flt.col, nstart, nlen, nxt);
#endif
      cnt[threadIdx.x] = 0; // (the run starts it held were consumed before scan_and_stage's last barrier)
      __syncthreads();
#pragma unroll
      for (int j = 0; j < RP_ROWS; j++) {
        // a filter leaves the staged tile partly empty: slots whose 512 positions are all past its end are
        // skipped (uniform branch), only the boundary slot stores to the sink rows
        if ((uint32_t)(j * RP_WG) < staged_len) store_row(j, staged_len);
        rank_row(j, len);
      }
#ifndef RP_PREFETCH_LATE
      __builtin_amdgcn_s_waitcnt(0x0F70);
#else
      rp_chunk_load<NV, RP_WG, RP_ROWS, PSRC>(key, v0, v1, flt.col, nstart, nlen, nxt);
      __builtin_amdgcn_s_waitcnt(vmcnt_imm(NLOADS < 63 ? NLOADS : 63));
#endif
      __syncthreads();
      scan_and_stage();
      staged_len = s_total;
      cur = nxt;
      len = nlen;
      cur_start = nstart;
    }
#pragma unroll
    for (int j = 0; j < RP_ROWS; j++)
      if ((uint32_t)(j * RP_WG) < staged_len) store_row(j, staged_len);
  }
  if (owner) { // publish what is left open
    out.chunk_len[cur_id] = cfill; // (chunks never taken keep the zero length of the table's memset)
    out.chunk_dig[cur_id] = threadIdx.x;
  }
  if (H2) { // ... and the histograms of the open chunks (no spill counts are left after the last tile)
    __syncthreads();
    close_id[threadIdx.x] = owner ? cur_id : 0xffffffffu;
    __syncthreads();
    flush_closed();
  }
}

// ------------------------------------------------------------ slim records (12 bytes per row) --
// The packed two-level partition above moves {key|row word, value} = 16 bytes per kept row through level 1, level 2
// and the bucket pass: 4 x 16 B.  The row id (30 bits at C5) is only there for the first-seen order of the groups
// (hash_agg.rs:87-99), and the key bits a level has already consumed travel on for nothing.  The slim form carries
//   level 1 -> level 2:  value (8 B) + u32 { key offset inside the level-1 digit | row inside its level-1 TILE << kshift }
//   level 2 -> buckets:  value (8 B) + u32 { slot inside the bucket | row inside its tile << rbits | tile DELTA << (rbits + 13) }
// and rebuilds the row id where it is needed: row = (base tile of the chunk + tile delta) * TILE + row inside the tile.
//  * Level 1: workgroup b reads CONSECUTIVE tiles and appends digit d's rows to its open chunk of d, so a chunk is a
//    sequence of runs, one per tile: the owner thread of the digit writes cstart[chunk][tile - base tile] = fill at the
//    start of the run (a 2-byte store per tile and digit) and base[chunk].  A chunk is closed early when the next tile
//    would be more than 127 tiles after its base (the delta has 7 bits).
//  * Level 2 reads a chunk as one input tile: from cstart[] it builds, in LDS, one bit per position where a non-empty
//    run starts + the prefix count of those bits per 64-position group + the delta of the k-th non-empty run; a row at
//    position p then finds its delta with two wave-uniform LDS reads, a popcount and one byte read.
//  * The bucket pass needs the base tile of the CHUNK a row came from: bucket b is the concatenation of the runs
//    (input tile i of its segment, digit) in tile order, whose starts are the scanned count matrix's column; a small
//    kernel (rp_slim_runs_kernel) compacts the non-empty runs of every column to {start, base tile} lists, and the
//    bucket pass does at bucket scale what level 2 does at chunk scale (agg_partition.hip, lds_agg_dense_slim_kernel).
// Rows whose key lies outside the dense range (fused join: no build partner) are dropped by level 1 instead of
// travelling to the last bucket.  C5: 48 -> 40 GB per step.
constexpr uint32_t SLIM_RUNS = 128;      // cstart entries per chunk = largest tile delta + 1
constexpr uint32_t SLIM_LOCAL_BITS = 13; // row inside a level-1 tile (tiles of <= 8192 rows)
struct SlimChunkOut {
  SlimRowsView rows;    // chunk c = rows [c * (cap + RP_CHUNK_SKEW), + cap)
  uint32_t *chunk_len;  // rows in chunk c (0 = never used)
  uint32_t *chunk_dig;  // level-1 digit of chunk c
  uint32_t *chunk_base; // level-1 tile of the chunk's first run
  uint16_t *cstart;     // [chunk][SLIM_RUNS] fill at the start of the run of tile base + i (0xffff = no such run)
  unsigned int *counter; // [1] arena overflow flag
  uint32_t max_chunks, cap, arena;
  uint32_t *hist;       // [chunk][next level's digits]
  uint32_t kshift;      // rbits + p2_bits: bits of the key offset inside a level-1 digit
  uint32_t max_delta;   // a chunk holds runs of tiles base .. base + max_delta (< SLIM_RUNS; smaller only as a test hook)
  uint32_t conc_eighths; // a tile with >= this many eighths of its row slots in ONE digit makes the next tile try the one-lane rank (9 = never)
};

// rows of the tile being ranked / staged: the key as its 32-bit offset in the dense range, ~0 = the row does not take
// part (past the end of a ragged tile, fails the predicate, no bucket).  Half the registers of the loaded form, and the
// predicate is evaluated once, when the prefetched rows become the current ones.
template <int RP_ROWS> struct SlimCur {
  uint32_t off[RP_ROWS];
  uint64_t a0[RP_ROWS];
};

template <int RP_WG, int RP_ROWS, int PSRC>
__global__ __launch_bounds__(RP_WG, 1) void rp_chunk_scatter_slim_kernel(
    const uint64_t *__restrict__ key, const uint64_t *__restrict__ v0, RowFilter flt, int64_t n, SlimChunkOut out,
    uint32_t P, uint32_t p2_bits, uint32_t digits, uint32_t num_tiles, uint32_t tiles_per_wg, int64_t sink, KeyPack kp) {
  constexpr uint32_t RP_TILE = RP_WG * RP_ROWS;
  static_assert(RP_TILE <= (1u << SLIM_LOCAL_BITS), "row inside a tile must fit SLIM_LOCAL_BITS");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t *sv0 = (uint64_t *)smem;
  uint32_t *sw = (uint32_t *)(sv0 + RP_TILE);
  uint16_t *sdig = (uint16_t *)(sw + RP_TILE);
  uint32_t *cnt = (uint32_t *)(sdig + RP_TILE); // [RP_WG]
  uint32_t *split = cnt + RP_WG;                // [RP_WG] tile-local position where a digit's run changes chunk
  int64_t *gb0 = (int64_t *)(split + RP_WG);    // [RP_WG] destination of position p: gb0[d] + p below the split,
  int64_t *gb1 = gb0 + RP_WG;                   //         gb1[d] + p from it on
  uint32_t *room_s = (uint32_t *)(gb1 + RP_WG); // [RP_WG] rows the digit's current chunk can still take
  uint32_t *close_id = room_s + RP_WG;          // [RP_WG] chunk the digit closes in this tile, or ~0
  uint32_t *h2 = close_id + RP_WG;              // [P] low half: rows of the digit's current chunk, high: spill of this tile
  __shared__ uint32_t s_wsum[RP_WG / 64];
  __shared__ uint32_t s_total;
  __shared__ uint32_t s_next; // next free chunk of this workgroup's arena
  const uint32_t d2n = 1u << p2_bits;
  const uint32_t inmask = (1u << out.kshift) - 1u;

  const uint32_t t0 = blockIdx.x * tiles_per_wg;
  const uint32_t t1 = min(num_tiles, t0 + tiles_per_wg);
  // chunk state of digit threadIdx.x
  uint32_t cur_id = blockIdx.x * out.arena + min(threadIdx.x, digits - 1), cfill = 0, cbase = 0;
  bool opened = false; // the current chunk has a first run (cbase is its tile)
  if (threadIdx.x == 0) s_next = digits;
  const bool owner = threadIdx.x < digits;
  room_s[threadIdx.x] = out.cap;
  for (uint32_t i = threadIdx.x; i < digits * d2n; i += RP_WG) h2[i] = 0;
  auto tile_start = [&](uint32_t t) { return (int64_t)t * RP_TILE; };
  auto tile_len = [&](uint32_t t) { return (uint32_t)min((int64_t)RP_TILE, n - (int64_t)t * RP_TILE); };

  ChunkRegs<1, RP_ROWS, PSRC> nxt;
  SlimCur<RP_ROWS> cur;
  uint32_t dr[RP_ROWS]; // digit << 16 | rank inside the tile's run of that digit; ~0 = row not kept
  uint32_t cur_tile = 0;
  // (`report`: false when the tile was loaded a second time — the pipeline re-reads a range's last tile instead of branching —
  //  so that a key outside the sampled range reaches the outlier list once)
  auto take_rows = [&](uint32_t len, uint32_t nxt_tile, bool report) { // nxt (as loaded: tile `nxt_tile`) -> cur
#pragma unroll
    for (int j = 0; j < RP_ROWS; j++) {
      bool keep = (uint32_t)(j * RP_WG) + threadIdx.x < len;
      if (PSRC == 1) keep = keep && row_passes(flt, nxt.a0[j]);
      if (PSRC == 3) keep = keep && row_passes(flt, nxt.pv[PSRC == 3 ? j : 0]);
      const uint64_t off = nxt.k[j] - kp.kmin;
      if (keep && off > kp.range) { // no bucket of the range partition holds this key: the row has no group / no partner
        // (optimistically sampled range: the row goes to the outlier list / the caller reruns with the exact range)
        if (kp.oob && report) key_out_of_range(kp, (uint32_t)((int64_t)nxt_tile * RP_TILE + (uint32_t)(j * RP_WG) + threadIdx.x));
        keep = false;
      }
      cur.off[j] = keep ? (uint32_t)off : 0xffffffffu;
      cur.a0[j] = nxt.a0[j];
    }
  };
  // (Rows ORDERED by key put all 64 rows of a wave into one bucket — 64 atomics on one LDS word for the rank and for the chunk
  //  histogram: bench.py, c5_variants.adversarial.sorted_fact, level 1 7.7 instead of 5.6 ms.  A wave-uniform fast path, one
  //  lane ranking the wave when all its kept rows share the bucket, was measured in one process against this form: sorted
  //  rows 7.75 -> 7.17 ms, random keys 5.65 -> 6.00 ms; tried only while it keeps succeeding — one test per tile on random
  //  keys — it was SLOWER on both, 5.45 -> 5.90 and 7.67 -> 8.07: the kernel sits at 256 VGPRs and the extra live values
  //  cost more than the atomics.  Not adopted; the headline's random keys decide.)
  // Round 5: the one-lane rank again, GATED like level 2's — tried only in a tile that follows a CONCENTRATED tile (one digit holds
  // >= 3/8 of the tile's row slots: ordered / clustered fact rows; random keys never take the test; SlimChunkOut::conc_eighths) — now that the 768-thread form
  // has the registers for it.  A wave whose kept rows all fall into ONE bucket is ranked by its first lane: one add on the digit's
  // counter, one on the bucket's chunk histogram (rows below the chunk's room / rows that spill, both halves in one add).
  __shared__ uint32_t s_conc;
  uint32_t conc_digit = 0xffffffffu; // (workgroup-uniform) the digit the previous tile was concentrated on, or none
  // The one-lane rank, once per WAVE and tile: while the previous tile was concentrated on one digit, the wave takes the
  // bucket b0 of its first row of that digit, counts its rows of b0 over all RP_ROWS slots, and ONE lane adds the total to
  // cnt / h2; the rows of b0 then take consecutive ranks from that base (slot by slot), every other row ranks itself.
  // (Per SLOT — two same-address LDS atomics per wave and slot, 192 per tile — ordered fact rows still cost level 1
  // 6.11 ms against 4.95 on random ones with the LDS pipe idle, profiles/r05x_sq_*.txt: the returning atomics of twelve
  // waves on one address are served one after the other.)
  uint32_t w_b0 = 0xffffffffu, w_base = 0; // (wave-uniform) the wave's bucket and the next free rank of its rows
  auto plan_wave = [&]() {
    w_b0 = 0xffffffffu;
    if (conc_digit == 0xffffffffu) return;
#pragma unroll
    for (int j = 0; j < RP_ROWS; j++) {
      const uint32_t bkt = cur.off[j] >> kp.rbits;
      const uint64_t m = __ballot(cur.off[j] != 0xffffffffu && (bkt >> p2_bits) == conc_digit);
      if (w_b0 == 0xffffffffu && m) w_b0 = (uint32_t)__builtin_amdgcn_readlane((int)bkt, __builtin_ctzll(m));
    }
    if (w_b0 == 0xffffffffu) return;
    uint32_t total = 0;
#pragma unroll
    for (int j = 0; j < RP_ROWS; j++)
      total += (uint32_t)__popcll(__ballot(cur.off[j] != 0xffffffffu && (cur.off[j] >> kp.rbits) == w_b0));
    if (total < 4u * RP_ROWS) { // a few rows only: every lane for itself
      w_b0 = 0xffffffffu;
      return;
    }
    const uint32_t d0 = w_b0 >> p2_bits;
    uint32_t base = 0;
    if (lane_id() == 0) base = atomicAdd(&cnt[d0], total);
    w_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    const uint32_t room = room_s[d0];
    const uint32_t nlo = room > w_base ? min(room - w_base, total) : 0u; // ranks w_base .. w_base + total - 1: those below the room
    if (lane_id() == 0) atomicAdd(&h2[w_b0], nlo + ((total - nlo) << 16));
  };
  auto rank_row = [&](int j) {
    dr[j] = 0xffffffffu;
    const bool in = cur.off[j] != 0xffffffffu;
    const uint32_t bkt = cur.off[j] >> kp.rbits;
    if (w_b0 != 0xffffffffu) {
      const bool peer = in && bkt == w_b0;
      const uint64_t same = __ballot(peer);
      if (peer) dr[j] = ((w_b0 >> p2_bits) << 16) | (w_base + (uint32_t)mbcnt(same));
      w_base += (uint32_t)__popcll(same);
      if (in && !peer) {
        const uint32_t d = bkt >> p2_bits;
        const uint32_t r = atomicAdd(&cnt[d], 1u);
        atomicAdd(&h2[bkt], r < room_s[d] ? 1u : 0x10000u);
        dr[j] = (d << 16) | r;
      }
      return;
    }
    if (in) {
      const uint32_t d = bkt >> p2_bits;
      const uint32_t r = atomicAdd(&cnt[d], 1u);
      atomicAdd(&h2[bkt], r < room_s[d] ? 1u : 0x10000u);
      dr[j] = (d << 16) | r;
    }
  };
  auto flush_closed = [&]() { // histograms of the chunks named in close_id[] -> out.hist, spill counts move down
    for (uint32_t i = threadIdx.x; i < digits * d2n; i += RP_WG) {
      const uint32_t c = close_id[i >> p2_bits];
      if (c == 0xffffffffu) continue;
      const uint32_t v = h2[i];
      out.hist[(size_t)c * d2n + (i & (d2n - 1))] = v & 0xffffu;
      h2[i] = v >> 16;
    }
  };
  auto take_chunk = [&]() { // next chunk of the workgroup's own arena (an LDS counter)
    const uint32_t o = atomicAdd(&s_next, 1u);
    if (o >= out.arena) out.counter[1] = 1; // cannot happen (the arena bound counts the early closes); never out of bounds
    return blockIdx.x * out.arena + min(o, out.arena - 1);
  };
  auto scan_and_stage = [&]() {
    const uint32_t c = cnt[threadIdx.x];
    if (c * 8u >= (uint32_t)RP_TILE * out.conc_eighths) s_conc = threadIdx.x + 1; // (this digit holds >= conc_eighths / 8 of the tile's row slots: the hint for the next tile)
    const uint32_t inc = wave_iscan_u32(c);
    if (lane_id() == 63) s_wsum[wave_id()] = inc;
    __syncthreads();
    conc_digit = s_conc - 1u; // (0 - 1 = none)
    uint32_t wbase = 0, tot = 0;
    for (int w = 0; w < RP_WG / 64; w++) {
      if (w < wave_id()) wbase += s_wsum[w];
      tot += s_wsum[w];
    }
    const uint32_t ls = wbase + inc - c;
    if (threadIdx.x == 0) s_total = tot;
    const uint32_t room = min(c, out.cap - cfill);
    int64_t g0 = (int64_t)cur_id * (out.cap + RP_CHUNK_SKEW) + cfill, g1 = 0;
    uint32_t closing = 0xffffffffu;
    if (owner && c > 0) {
      if (!opened) { // first run of this chunk
        opened = true;
        cbase = cur_tile;
        out.chunk_base[cur_id] = cur_tile;
      }
      out.cstart[(size_t)cur_id * SLIM_RUNS + (cur_tile - cbase)] = (uint16_t)cfill; // (delta <= 127: see the early close below)
    }
    if (owner && c > room) { // the run spills into the next chunk of the arena
      out.chunk_len[cur_id] = out.cap;
      out.chunk_dig[cur_id] = threadIdx.x;
      closing = cur_id;
      cur_id = take_chunk();
      cfill = c - room;
      cbase = cur_tile;
      out.chunk_base[cur_id] = cur_tile;
      out.cstart[(size_t)cur_id * SLIM_RUNS] = 0;
      g1 = (int64_t)cur_id * (out.cap + RP_CHUNK_SKEW);
    } else {
      cfill += c;
      if (owner && opened && cur_tile + 1 - cbase > out.max_delta) { // the next tile's delta would not fit: close early
        out.chunk_len[cur_id] = cfill;
        out.chunk_dig[cur_id] = threadIdx.x;
        closing = cur_id;
        cur_id = take_chunk();
        cfill = 0;
        opened = false;
      }
    }
    close_id[threadIdx.x] = closing;
    cnt[threadIdx.x] = ls;
    split[threadIdx.x] = ls + room;
    gb0[threadIdx.x] = g0 - (int64_t)ls;
    gb1[threadIdx.x] = g1 - (int64_t)(ls + room);
    room_s[threadIdx.x] = out.cap - cfill;
    __syncthreads();
    flush_closed();
#pragma unroll
    for (int j = 0; j < RP_ROWS; j++) {
      if (dr[j] == 0xffffffffu) continue;
      const uint32_t d = dr[j] >> 16;
      const uint32_t p = cnt[d] + (dr[j] & 0xffffu);
      const uint32_t local = (uint32_t)(j * RP_WG) + threadIdx.x;
      sv0[p] = cur.a0[j];
      sw[p] = (cur.off[j] & inmask) | (local << out.kshift);
      sdig[p] = (uint16_t)d;
    }
    __syncthreads();
  };
  auto store_row = [&](int j, uint32_t len) {
    const uint32_t p = j * RP_WG + threadIdx.x;
    const uint32_t d = min((uint32_t)sdig[p], (uint32_t)(RP_WG - 1)); // (clamped, not masked: RP_WG = 768 is no power of two)
    int64_t g = (p < split[d] ? gb0[d] : gb1[d]) + p;
    if (p >= len) g = sink + (int64_t)blockIdx.x * RP_WG + threadIdx.x;
    slim_store(out.rows, g, sw[p], sv0[p]);
  };

  if (t0 < t1) {
    uint32_t len = tile_len(t0);
    cur_tile = t0;
    rp_chunk_load<1, RP_WG, RP_ROWS, PSRC>(key, v0, nullptr, flt.col, tile_start(t0), len, nxt);
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
    take_rows(len, t0, true);
    cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_conc = 0;
    __syncthreads();
    plan_wave();
#pragma unroll
    for (int j = 0; j < RP_ROWS; j++) rank_row(j);
    __syncthreads();
    scan_and_stage();
    uint32_t staged_len = s_total;
    uint32_t tcur = min(t0 + 1, t1 - 1);
    len = tile_len(tcur);
    cur_tile = tcur;
    rp_chunk_load<1, RP_WG, RP_ROWS, PSRC>(key, v0, nullptr, flt.col, tile_start(tcur), len, nxt);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    take_rows(len, tcur, tcur != t0);
    for (uint32_t ti = t0 + 1; ti < t1; ti++) {
      const uint32_t tnext = min(ti + 1, t1 - 1);
      const uint32_t nlen = tile_len(tnext);
      rp_chunk_load<1, RP_WG, RP_ROWS, PSRC>(key, v0, nullptr, flt.col, tile_start(tnext), nlen, nxt);
      cnt[threadIdx.x] = 0;
      if (threadIdx.x == 0) s_conc = 0; // (its last reader passed the barriers inside scan_and_stage)
      __syncthreads();
      plan_wave();
#pragma unroll
      for (int j = 0; j < RP_ROWS; j++) {
        if ((uint32_t)(j * RP_WG) < staged_len) store_row(j, staged_len);
        rank_row(j);
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
      scan_and_stage();
      staged_len = s_total;
      take_rows(nlen, tnext, tnext != ti);
      cur_tile = tnext;
    }
#pragma unroll
    for (int j = 0; j < RP_ROWS; j++)
      if ((uint32_t)(j * RP_WG) < staged_len) store_row(j, staged_len);
  }
  if (owner) { // publish what is left open
    out.chunk_len[cur_id] = cfill;
    out.chunk_dig[cur_id] = threadIdx.x;
  }
  __syncthreads();
  close_id[threadIdx.x] = owner ? cur_id : 0xffffffffu;
  __syncthreads();
  flush_closed();
}

// Level 2 of the slim form: input tile = one chunk of level 1 (values + words), output = value / word columns in
// bucket order.  The pipeline is rp_scatter_kernel's; what is new is the tile-delta lookup (see the section header).
struct SlimIn {
  SlimRowsView rows;
  const uint32_t *tile_chunk;  // input tile -> chunk
  const uint16_t *cstart;      // [chunk][SLIM_RUNS]
};
struct SlimOut {
  SlimRowsView rows;
};
template <int RP_ROWS> struct SlimRegs {
  uint64_t a0[RP_ROWS];
  uint32_t w[RP_ROWS];
  uint32_t goff;      // offs[] entry of (digit threadIdx.x, this tile)
  uint16_t cs0, cs1;  // wave 0 only: cstart[lane], cstart[64 + lane] of the tile's chunk
};
template <int RP_WG, int RP_ROWS>
__device__ __forceinline__ void rp_slim_load(const SlimIn &in, const Tile &t, uint32_t tile_index,
                                             const uint32_t *__restrict__ offs, uint32_t digits, SlimRegs<RP_ROWS> &r) {
#pragma unroll
  for (int j = 0; j < RP_ROWS; j++) {
    const int64_t row = t.start + min((uint32_t)(j * RP_WG) + threadIdx.x, t.len - 1);
    slim_load_nt(in.rows, row, r.w[j], r.a0[j]);
  }
  r.goff = offs[(int64_t)tile_index * digits + min(threadIdx.x, digits - 1)];
  // (every wave loads — uniform control flow keeps the unrolled load sequence free of early waits; only wave 0 uses them)
  const uint16_t *cs = in.cstart + (size_t)in.tile_chunk[tile_index] * SLIM_RUNS;
  r.cs0 = cs[lane_id()];
  r.cs1 = cs[64 + lane_id()];
}

template <int RP_WG, int RP_ROWS>
__global__ __launch_bounds__(RP_WG, 1) void rp_scatter_slim_kernel(SlimIn in, SlimOut out, const Tile *__restrict__ tiles,
                                                                  uint32_t p2_bits, uint32_t digits,
                                                                  const uint32_t *__restrict__ offs, uint32_t num_tiles,
                                                                  uint32_t tiles_per_wg, int64_t sink, uint32_t kshift,
                                                                  uint32_t rbits) {
  constexpr int RP_TILE = RP_WG * RP_ROWS;
  static_assert(SLIM_RUNS == 128 && RP_TILE / 64 <= 128, "one mask word per 64 positions of a chunk");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t *sv0 = (uint64_t *)smem;
  unsigned long long *rmask = (unsigned long long *)(sv0 + RP_TILE); // [128] bit = a non-empty run starts at this position
  int64_t *gbase = (int64_t *)(rmask + 128);                         // [RP_WG]
  uint32_t *sw = (uint32_t *)(gbase + RP_WG);
  uint32_t *cnt = sw + RP_TILE;      // [RP_WG]
  uint32_t *lstart = cnt + RP_WG;    // [RP_WG]
  uint32_t *rpre = lstart + RP_WG;   // [128] non-empty runs that start before group g
  uint8_t *sdig = (uint8_t *)(rpre + 128);
  uint8_t *rdelta = sdig + RP_TILE;  // [128] tile delta of the k-th non-empty run
  __shared__ uint32_t s_wsum[RP_WG / 64];
  __shared__ uint32_t s_psum;
  const uint32_t d2mask = (1u << p2_bits) - 1u, slotmask = (1u << rbits) - 1u;
  const uint64_t le_mask = (2ull << lane_id()) - 1ull; // bits 0 .. lane

  // (a segment-synchronous schedule — every workgroup inside the same level-1 segment at any time, 64 instead of ~2400
  //  bucket regions written concurrently — was measured SLOWER in every one of six placements, 2.66-3.24 against 2.54-3.21 ms:
  //  the spread of this kernel between allocations is not a TLB-reach effect)
  const uint32_t t0 = blockIdx.x * tiles_per_wg;
  const uint32_t t1 = min(num_tiles, t0 + tiles_per_wg);
  if (t0 >= t1) return;
  auto tile_of = [&](uint32_t slot) { return slot; };
  SlimRegs<RP_ROWS> cur, nxt;
  uint32_t dr[RP_ROWS]; // digit << 16 | rank inside the tile's run of that digit; ~0 = no row
  // Rows ordered by key: the 64 rows of a wave share their digit, 64 atomics on one LDS word.  One lane ranks such a wave —
  // but only in a tile that follows a CONCENTRATED tile (>= 90 % of its rows in one digit, seen for free when the counters
  // are scanned): random keys never take the test (without the gate: 2.82 against 2.75 ms in one process, inside the
  // placement noise but not clearly free; sorted fact rows: this level 3.49 -> 2.63 ms).
  __shared__ uint32_t s_conc;
  bool conc_hint = false; // (workgroup-uniform) the previous tile was concentrated
  auto rank_row = [&](int j, uint32_t len) {
    dr[j] = 0xffffffffu;
    const bool in = (uint32_t)(j * RP_WG) + threadIdx.x < len;
    const uint32_t d = (cur.w[j] >> rbits) & d2mask;
    if (conc_hint) {
      const uint64_t m = __ballot(in);
      if (m) {
        const int first = __builtin_ctzll(m);
        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, first);
        const bool peer = in && d == d0; // (round 5: the first lane's digit by one lane, the few other lanes — a hot key's neighbours — by themselves)
        const uint64_t same = __ballot(peer);
        if (__popcll(same) >= 8) {
          uint32_t base = 0;
          if ((int)lane_id() == first) base = atomicAdd(&cnt[d0], (uint32_t)__popcll(same));
          base = (uint32_t)__builtin_amdgcn_readlane((int)base, first);
          if (peer) dr[j] = (d0 << 16) | (base + (uint32_t)mbcnt(same));
          else if (in) dr[j] = (d << 16) | atomicAdd(&cnt[d], 1u);
          return;
        }
      }
    }
    if (in) dr[j] = (d << 16) | atomicAdd(&cnt[d], 1u);
  };
  // wave 0: the chunk's run table -> rmask (bits), rdelta (k-th non-empty run -> tile delta).  A written cstart entry
  // below the chunk's length IS a non-empty run (level 1 writes an entry only for a tile that brings rows).
  auto build_runs = [&](uint32_t len) {
    if (wave_id() != 0) return;
    const bool f0 = cur.cs0 < len, f1 = cur.cs1 < len;
    const uint64_t b0 = __ballot(f0), b1 = __ballot(f1);
    const uint64_t lt = le_mask >> 1;
    if (f0) {
      rdelta[__popcll(b0 & lt)] = (uint8_t)lane_id();
      atomicOr(&rmask[cur.cs0 >> 6], 1ull << (cur.cs0 & 63));
    }
    if (f1) {
      rdelta[__popcll(b0) + __popcll(b1 & lt)] = (uint8_t)(64 + lane_id());
      atomicOr(&rmask[cur.cs1 >> 6], 1ull << (cur.cs1 & 63));
    }
  };
  auto scan_and_stage = [&](uint32_t tile_len) {
    uint32_t c = cnt[threadIdx.x];
    if (tile_len && c * 10u >= tile_len * 9u) s_conc = 1; // (one digit holds >= 90 % of the tile: the hint for the next tile)
    uint32_t inc = wave_iscan_u32(c);
    if (lane_id() == 63) s_wsum[wave_id()] = inc;
    // prefix count of the run bits per 64-position group (threads 0..127 = waves 0 and 1)
    uint32_t pc = 0, pinc = 0;
    if (threadIdx.x < 128) {
      pc = (uint32_t)__popcll(rmask[threadIdx.x]);
      pinc = wave_iscan_u32(pc);
      if (threadIdx.x == 63) s_psum = pinc;
    }
    __syncthreads();
    conc_hint = s_conc != 0;
    uint32_t wbase = 0;
    for (int w = 0; w < wave_id(); w++) wbase += s_wsum[w];
    uint32_t ls = wbase + inc - c;
    lstart[threadIdx.x] = ls;
    gbase[threadIdx.x] = (int64_t)cur.goff - (int64_t)ls;
    if (threadIdx.x < 128) rpre[threadIdx.x] = pinc - pc + (threadIdx.x >= 64 ? s_psum : 0u);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RP_ROWS; j++) {
      if (dr[j] == 0xffffffffu) continue;
      const uint32_t p = lstart[dr[j] >> 16] + (dr[j] & 0xffffu);
      const uint32_t g = (uint32_t)(j * (RP_WG / 64)) + (uint32_t)wave_id(); // 64-position group of input position j * RP_WG + tid
      const uint32_t k = rpre[g] + (uint32_t)__popcll(rmask[g] & le_mask) - 1u;
      const uint32_t delta = rdelta[k & (SLIM_RUNS - 1)];
      const uint32_t w = cur.w[j];
      sv0[p] = cur.a0[j];
      sw[p] = (w & slotmask) | (((w >> kshift) & ((1u << SLIM_LOCAL_BITS) - 1u)) << rbits) | (delta << (rbits + SLIM_LOCAL_BITS));
      sdig[p] = (uint8_t)(dr[j] >> 16);
    }
    __syncthreads();
  };
  auto store_row = [&](int j, uint32_t len) {
    const uint32_t p = j * RP_WG + threadIdx.x;
    int64_t g = gbase[sdig[p] & d2mask] + p;
    if (p >= len) g = sink + threadIdx.x;
    slim_store(out.rows, g, sw[p], sv0[p]);
  };

  uint32_t tix = tile_of(t0);
  Tile t = tiles[tix];
  rp_slim_load<RP_WG, RP_ROWS>(in, t, tix, offs, digits, cur);
  __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
  cnt[threadIdx.x] = 0;
  if (threadIdx.x < 128) rmask[threadIdx.x] = 0;
  if (threadIdx.x == 0) s_conc = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < RP_ROWS; j++) rank_row(j, t.len);
  build_runs(t.len);
  __syncthreads();
  scan_and_stage(t.len);
  uint32_t staged_len = t.len;
  tix = tile_of(min(t0 + 1, t1 - 1));
  t = tiles[tix];
  rp_slim_load<RP_WG, RP_ROWS>(in, t, tix, offs, digits, cur);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  for (uint32_t ti = t0 + 1; ti < t1; ti++) {
    const uint32_t tnext = tile_of(min(ti + 1, t1 - 1));
    Tile tn = tiles[tnext];
    rp_slim_load<RP_WG, RP_ROWS>(in, tn, tnext, offs, digits, nxt);
    cnt[threadIdx.x] = 0;
    if (threadIdx.x < 128) rmask[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_conc = 0; // (its last reader passed the barrier inside scan_and_stage)
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RP_ROWS; j++) {
      store_row(j, staged_len);
      rank_row(j, t.len);
    }
    build_runs(t.len);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    scan_and_stage(t.len);
    staged_len = t.len;
    cur = nxt;
    t = tn;
  }
#pragma unroll
  for (int j = 0; j < RP_ROWS; j++) store_row(j, staged_len);
}

// Non-empty runs of every bucket's column of the scanned (digit-major) count matrix, compacted in place:
// bucket b = (segment s, digit d) owns entries [col, col + tiles of s), col = seg_mat[s] + d * seg_tiles[s]; run i of the
// column is the rows input tile i of the segment sent to the bucket, offs[col + i] its first row in the output.
// nzstart[col + k] / nzbt[col + k] = start and BASE TILE (of the chunk the input tile was) of the column's k-th
// non-empty run, nzcount[b] their number, bcol[b] = col.  One workgroup per bucket.
__global__ __launch_bounds__(256) void rp_slim_runs_kernel(const uint32_t *__restrict__ offs, int64_t entries,
                                                           const uint64_t *__restrict__ total, const int64_t *__restrict__ seg_mat,
                                                           const uint32_t *__restrict__ seg_tiles,
                                                           const uint32_t *__restrict__ seg_tile_base,
                                                           const uint32_t *__restrict__ tile_chunk,
                                                           const uint32_t *__restrict__ chunk_base, uint32_t digits,
                                                           uint32_t *__restrict__ nzstart, uint32_t *__restrict__ nzbt,
                                                           uint32_t *__restrict__ nzcount, uint32_t *__restrict__ bcol) {
  __shared__ uint32_t s_w[4];
  const uint32_t b = blockIdx.x, s = b / digits, d = b % digits;
  const uint32_t nt = seg_tiles[s];
  const int64_t col = seg_mat[s] + (int64_t)d * nt;
  uint32_t done = 0;
  for (uint32_t i0 = 0; i0 < nt; i0 += 256) {
    const uint32_t i = i0 + threadIdx.x;
    uint32_t start = 0, next = 0;
    if (i < nt) {
      start = offs[col + i];
      next = col + i + 1 < entries ? offs[col + i + 1] : (uint32_t)*total;
    }
    const bool f = i < nt && next > start;
    const uint64_t bal = __ballot(f);
    if (lane_id() == 0) s_w[wave_id()] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t before = done, all = 0;
    for (int w = 0; w < 4; w++) {
      if (w < wave_id()) before += s_w[w];
      all += s_w[w];
    }
    if (f) {
      const uint32_t k = before + (uint32_t)mbcnt(bal);
      nzstart[col + k] = start;
      nzbt[col + k] = chunk_base[tile_chunk[seg_tile_base[s] + i]];
    }
    done += all;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    nzcount[b] = done;
    bcol[b] = (uint32_t)col;
  }
}

// ------------------------------------------------------------ claimed single level --
// A ONE-level partition (<= 512 buckets) without a histogram pass, with an optional row filter fused in: what the
// chunked first level is to two-level partitions.  The counting form reads every key twice (histogram, then
// scatter) and needs the (tile, digit) count matrix scanned in between (C4, 2e8 rows: 0.26 + 0.15 ms of 2.9).
//
// Here every bucket owns a REGION sized from a SAMPLE of the batch (rp_sample_hist_kernel: an eighth of every tile,
// the predicate applied; rp_region_plan_kernel: estimate x 9/8 + slack, prefix sum -> region starts), and a
// workgroup appends the rows of digit d to its own open BLOCK of that region — B rows, a power of two: every block
// starts on a cache-line boundary — taking the next block(s) with one atomic add on the region's cursor when the
// open one is full (`need` rows -> ceil(need / B) consecutive blocks in one claim, so a tile's run of one digit still
// has at most two destinations: the rest of the open block, then the new claim).  Per-tile claims of exactly the
// run's length would need no padding, but every run would then start and end inside a cache line shared with a run
// of another workgroup — another XCD's L2 — and partial lines are written to HBM from both (the 2.2x write
// amplification round 1 measured when neighbouring runs did not meet in one L2).
//
// What a workgroup leaves unfilled of its last block of every digit is filled with SENTINEL rows (all-ones key|row
// word = the packed form of "key outside the range", which the bucket pass already skips): bucket b = slots
// [start[b], cursor[b]) with holes of that kind, about wgs * B / 2 per bucket (B is chosen so that this is ~3 % of
// the rows).  A region that turns out too small (estimate off: clustered input whose clusters the sample missed)
// raises a flag and sends the rows of that digit to the sink; the caller then runs the counting level.
struct ClaimOut {
  uint64_t *key, *v0;    // column form (REC = false)
  u64x2 *rec;            // record form
  uint32_t *cursor;      // [P] next free slot of the bucket's region (starts at the region's first slot)
  const uint32_t *rend;  // [P] first slot behind the region
  unsigned int *flag;    // [0] set when a region overflowed
  unsigned long long *kept; // rows that passed the filter
  uint32_t B;            // block size in rows (power of two)
};

// est[d] += rows of digit d among the sampled rows that pass the filter; est[P] += sampled rows
__global__ __launch_bounds__(256) void rp_sample_hist_kernel(const uint64_t *__restrict__ key, RowFilter flt, int64_t n,
                                                             uint32_t tile_rows, uint32_t num_tiles, uint32_t P, KeyPack kp,
                                                             uint32_t sdiv, uint32_t *__restrict__ est) {
  __shared__ uint32_t h[512];
  __shared__ uint32_t s_seen;
  for (uint32_t i = threadIdx.x; i < 512; i += 256) h[i] = 0;
  if (threadIdx.x == 0) s_seen = 0;
  __syncthreads();
  // sampled rows per tile: the first tile_rows / sdiv rows (sdiv = 8, 16 or 32) of the (5 t mod 8)-th eighth of tile t
  const uint32_t S = tile_rows / sdiv;
  const int64_t total = (int64_t)num_tiles * S;
  uint32_t seen = 0;
  constexpr int U = 4;
  for (int64_t i0 = ((int64_t)blockIdx.x * U) * 256 + threadIdx.x; i0 < total; i0 += (int64_t)gridDim.x * U * 256) {
    uint64_t k[U], f[U];
    bool in[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = i0 + (int64_t)u * 256;
      const uint32_t t = (uint32_t)(i / S), o = (uint32_t)(i % S);
      const int64_t row = (int64_t)t * tile_rows + (int64_t)((t * 5u) & 7u) * (tile_rows / 8) + o;
      in[u] = i < total && row < n;
      const int64_t r = in[u] ? row : 0;
      k[u] = __builtin_nontemporal_load(key + r);
      f[u] = flt.col ? __builtin_nontemporal_load(flt.col + r) : 0ull;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (!in[u]) continue;
      seen++;
      if (flt.col && !row_passes(flt, f[u])) continue;
      atomicAdd(&h[rp_bucket(kp, packed_clamp(kp, k[u]), true, P)], 1u);
    }
  }
  atomicAdd(&s_seen, seen);
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < P; d += 256)
    if (h[d]) atomicAdd(&est[d], h[d]);
  if (threadIdx.x == 0 && s_seen) atomicAdd(&est[P], s_seen);
}

// regions from the sampled counts: cap[d] = est[d] * n / sampled * 9/8 + slack, rounded up to whole blocks
__global__ __launch_bounds__(512) void rp_region_plan_kernel(const uint32_t *__restrict__ est, uint32_t P, int64_t n, uint32_t slack,
                                                             uint32_t B, uint32_t *__restrict__ rstart, uint32_t *__restrict__ rend,
                                                             uint32_t *__restrict__ cursor) {
  __shared__ uint32_t s_wsum[8];
  const uint32_t d = threadIdx.x;
  const uint32_t seen = est[P];
  uint32_t cap = 0;
  if (d < P) {
    const uint64_t scaled = seen ? (uint64_t)((double)est[d] * (double)n / (double)seen) : 0ull;
    cap = (uint32_t)((scaled + scaled / 8 + slack + B - 1) & ~(uint64_t)(B - 1));
  }
  const uint32_t inc = wave_iscan_u32(cap);
  if (lane_id() == 63) s_wsum[wave_id()] = inc;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < wave_id(); w++) base += s_wsum[w];
  const uint32_t start = base + inc - cap;
  if (d < P) {
    rstart[d] = start;
    cursor[d] = start;
    rend[d] = start + cap;
  }
}

template <int NV, int RP_WG, int RP_ROWS, int PSRC, bool REC>
__global__ __launch_bounds__(RP_WG, 1) void rp_claim_scatter_kernel(
    const uint64_t *__restrict__ key, const uint64_t *__restrict__ v0, RowFilter flt, int64_t n, ClaimOut out, uint32_t P,
    uint32_t num_tiles, uint32_t tiles_per_wg, int64_t sink, KeyPack kp) {
  constexpr uint32_t RP_TILE = RP_WG * RP_ROWS;
  constexpr int64_t DEAD = -(1ll << 40); // destination base of a digit whose region overflowed: g < 0 -> the sink
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t *skey = (uint64_t *)smem;
  uint64_t *sv0 = skey + RP_TILE;
  uint32_t *cnt = (uint32_t *)(sv0 + (NV >= 1 ? RP_TILE : 0)); // [RP_WG]
  uint32_t *split = cnt + RP_WG;                                // [RP_WG] tile-local position where a digit's run changes block
  int64_t *gb0 = (int64_t *)(split + RP_WG);                    // [RP_WG] destination of position p: gb0[d] + p below the split,
  int64_t *gb1 = gb0 + RP_WG;                                   //         gb1[d] + p from it on
  __shared__ uint32_t s_wsum[RP_WG / 64];
  __shared__ uint32_t s_total;

  const uint32_t t0 = blockIdx.x * tiles_per_wg;
  const uint32_t t1 = min(num_tiles, t0 + tiles_per_wg);
  // open block of digit threadIdx.x: next free slot, slots left
  const bool owner = threadIdx.x < P;
  uint32_t pos = 0, room = 0;
  bool dead = false;
  const uint32_t my_end = owner ? out.rend[threadIdx.x] : 0u;
  const uint32_t Bm1 = out.B - 1;
  unsigned long long kept = 0;
  auto tile_start = [&](uint32_t t) { return (int64_t)t * RP_TILE; };
  auto tile_len = [&](uint32_t t) { return (uint32_t)min((int64_t)RP_TILE, n - (int64_t)t * RP_TILE); };

  ChunkRegs<NV, RP_ROWS, PSRC> cur, nxt;
  uint32_t dg[RP_ROWS], rk[RP_ROWS];
  int64_t cur_start = 0;
  auto rank_row = [&](int j, uint32_t len) {
    dg[j] = 0xffffffffu;
    bool keep = (uint32_t)(j * RP_WG) + threadIdx.x < len;
    if (PSRC == 1) keep = keep && row_passes(flt, cur.a0[NV >= 1 ? j : 0]);
    if (PSRC == 3) keep = keep && row_passes(flt, cur.pv[PSRC == 3 ? j : 0]);
    if (keep) {
      dg[j] = rp_bucket(kp, packed_clamp(kp, cur.k[j]), true, P);
      rk[j] = atomicAdd(&cnt[dg[j]], 1u);
    }
  };
  auto scan_and_stage = [&]() {
    const uint32_t c = cnt[threadIdx.x];
    // the claim first: its answer is needed only behind the staging loop
    const uint32_t used = min(c, room), need = c - used;
    const bool claim = owner && need > 0;
    const uint32_t k = (need + Bm1) & ~Bm1;
    uint32_t got = 0;
    if (claim && !dead) got = atomicAdd(&out.cursor[threadIdx.x], k);
    const uint32_t inc = wave_iscan_u32(c);
    if (lane_id() == 63) s_wsum[wave_id()] = inc;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
    for (int w = 0; w < RP_WG / 64; w++) {
      if (w < wave_id()) wbase += s_wsum[w];
      tot += s_wsum[w];
    }
    const uint32_t ls = wbase + inc - c;
    if (threadIdx.x == 0) {
      s_total = tot;
      kept += tot;
    }
    cnt[threadIdx.x] = ls; // run start (rank_row's counters are consumed)
    split[threadIdx.x] = ls + used;
    gb0[threadIdx.x] = (int64_t)pos - (int64_t)ls;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RP_ROWS; j++) {
      if (dg[j] == 0xffffffffu) continue;
      const uint32_t p = cnt[dg[j]] + rk[j];
      const uint32_t row = (uint32_t)(cur_start + (uint32_t)(j * RP_WG) + threadIdx.x);
      skey[p] = pack_key_row(kp, cur.k[j], row);
      if (NV >= 1) sv0[p] = cur.a0[j];
    }
    int64_t g1 = 0;
    if (claim) {
      if (!dead && (uint64_t)got + k > (uint64_t)my_end) {
        dead = true;
        *out.flag = 1u;
      }
      if (dead) {
        g1 = DEAD;
        pos = 0;
        room = 0;
      } else {
        g1 = (int64_t)got - (int64_t)(ls + used);
        pos = got + need;
        room = k - need;
      }
    } else {
      pos += c;
      room -= c;
    }
    gb1[threadIdx.x] = g1;
    __syncthreads();
  };
  auto store_row = [&](int j, uint32_t len) { // position p of the staged tile -> its block
    const uint32_t p = j * RP_WG + threadIdx.x;
    const uint64_t kw = skey[p];
    const uint32_t d = min(rp_bucket(kp, packed_key(kp, kw), true, P), (uint32_t)(RP_WG - 1));
    int64_t g = (p < split[d] ? gb0[d] : gb1[d]) + p;
    if (p >= len || g < 0) g = sink + (int64_t)blockIdx.x * RP_WG + threadIdx.x; // past the staged rows / overflowed region: sink rows
    if (REC) {
      u64x2 rec;
      rec.x = kw;
      rec.y = sv0[NV >= 1 ? p : 0];
      RP_ST(&out.rec[g], rec);
      return;
    }
    out.key[g] = kw;
    if (NV >= 1) out.v0[g] = sv0[p];
  };

  if (t0 < t1) {
    // the software pipeline of rp_chunk_scatter_kernel: while tile i-1 (staged, sorted) is written out, tile i
    // (in `cur`) is ranked and tile i+1 is in flight into `nxt`
    uint32_t len = tile_len(t0);
    cur_start = tile_start(t0);
    rp_chunk_load<NV, RP_WG, RP_ROWS, PSRC>(key, v0, nullptr, flt.col, cur_start, len, cur);
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
    cnt[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RP_ROWS; j++) rank_row(j, len);
    __syncthreads();
    scan_and_stage();
    uint32_t staged_len = s_total;
    uint32_t tcur = min(t0 + 1, t1 - 1);
    len = tile_len(tcur);
    cur_start = tile_start(tcur);
    rp_chunk_load<NV, RP_WG, RP_ROWS, PSRC>(key, v0, nullptr, flt.col, cur_start, len, cur);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (uint32_t ti = t0 + 1; ti < t1; ti++) {
      const uint32_t tnext = min(ti + 1, t1 - 1);
      const uint32_t nlen = tile_len(tnext);
      const int64_t nstart = tile_start(tnext);
      rp_chunk_load<NV, RP_WG, RP_ROWS, PSRC>(key, v0, nullptr, flt.col, nstart, nlen, nxt);
      cnt[threadIdx.x] = 0;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < RP_ROWS; j++) {
        if ((uint32_t)(j * RP_WG) < staged_len) store_row(j, staged_len);
        rank_row(j, len);
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
      scan_and_stage();
      staged_len = s_total;
      cur = nxt;
      len = nlen;
      cur_start = nstart;
    }
#pragma unroll
    for (int j = 0; j < RP_ROWS; j++)
      if ((uint32_t)(j * RP_WG) < staged_len) store_row(j, staged_len);
  }
  // what is left of every open block: sentinel rows
  if (owner && !dead) {
    for (uint32_t i = 0; i < room; i++) {
      if (REC) {
        u64x2 rec;
        rec.x = ~0ull;
        rec.y = 0;
        out.rec[pos + i] = rec;
      } else {
        out.key[pos + i] = ~0ull;
      }
    }
  }
  if (threadIdx.x == 0 && kept) atomicAdd(out.kept, kept);
}

// ---- the claimed single level in the SLIM form (round 5): 12 bytes per row out of the level instead of 16 ---------------
// What "slim records" above are to the two-level partition, for the claimed level (C4: 2e8 rows x 16 B in, the same out,
// the same again into the bucket pass).  A row leaves as value (8 B) + u32 { slot in the bucket | row inside its TILE <<
// rbits | tile DELTA << (rbits + 13) }; the bucket and the row id are rebuilt where they are needed: the bucket is the
// region the row lies in, and row = (base tile of the row's BLOCK + delta) * TILE + row inside the tile.  A block (B
// consecutive slots of a region, B-aligned) is filled by ONE workgroup with rows of consecutive tiles, so one u32 per block
// — the tile that claimed it, blk_bt[slot / B] — is all the bucket pass needs: a wave's 64 rows read one or two entries.
// The delta has 7 bits: a digit that sees a row only every few hundred tiles abandons what is left of its open block
// (sentinel rows) once the next tile would be more than 127 tiles after the block's first.  Sentinel = all ones: "row
// 8191 of its tile" does not exist in a tile of 6144 rows.
struct SlimClaimOut {
  SlimRowsView rows;
  uint32_t *cursor;      // [P] next free slot of the bucket's region
  const uint32_t *rend;  // [P] first slot behind the region
  uint32_t *blk_bt;      // [slots / B] tile of the claim that took the block
  unsigned int *flag;    // [0] set when a region overflowed
  unsigned long long *kept;
  uint32_t B, log_b;
  uint32_t max_delta;    // <= 127 (smaller only as a test hook)
};
constexpr uint32_t SLIM_SENTINEL = 0xffffffffu;
// (Round 6, review r05 #3c — fewer LDS instructions per row: the per-digit tables a row reads packed into ONE 16-byte entry per phase
//  ({run start, split, delta} while staging, {split, destination below / from the split on} while storing, 32-bit destinations) instead
//  of three 4 / 8-byte arrays, 13 -> 10 LDS instructions per row: 1.947-1.963 against 1.949-1.963 ms on one placement, alternating
//  in one process.  The kernel does not run at the pace of its LDS instructions; not kept.)

template <int RP_WG, int RP_ROWS, int PSRC>
__global__ __launch_bounds__(RP_WG, 1) void rp_claim_scatter_slim_kernel(
    const uint64_t *__restrict__ key, const uint64_t *__restrict__ v0, RowFilter flt, int64_t n, SlimClaimOut out, uint32_t P,
    uint32_t num_tiles, uint32_t tiles_per_wg, int64_t sink, KeyPack kp) {
  constexpr uint32_t RP_TILE = RP_WG * RP_ROWS;
  static_assert(RP_TILE < (1u << SLIM_LOCAL_BITS), "row 8191 of a tile is the sentinel");
  constexpr int64_t DEAD = -(1ll << 40); // destination base of a digit whose region overflowed: g < 0 -> the sink
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t *sv0 = (uint64_t *)smem;
  uint32_t *sw = (uint32_t *)(sv0 + RP_TILE);
  uint16_t *sdig = (uint16_t *)(sw + RP_TILE);
  uint32_t *cnt = (uint32_t *)(sdig + RP_TILE); // [RP_WG]
  uint32_t *split = cnt + RP_WG;                // [RP_WG] tile-local position where a digit's run changes block
  int64_t *gb0 = (int64_t *)(split + RP_WG);    // [RP_WG] destination of position p: gb0[d] + p below the split,
  int64_t *gb1 = gb0 + RP_WG;                   //         gb1[d] + p from it on
  uint32_t *dl0 = (uint32_t *)(gb1 + RP_WG);    // [RP_WG] tile delta (already shifted) of the rows below the split; 0 from it on
  __shared__ uint32_t s_wsum[RP_WG / 64];
  __shared__ uint32_t s_total;

  const uint32_t t0 = blockIdx.x * tiles_per_wg;
  const uint32_t t1 = min(num_tiles, t0 + tiles_per_wg);
  const bool owner = threadIdx.x < P;
  uint32_t pos = 0, room = 0, bbase = 0; // open block of digit threadIdx.x: next free slot, slots left, tile of its claim
  bool dead = false;
  const uint32_t my_end = owner ? out.rend[threadIdx.x] : 0u;
  const uint32_t Bm1 = out.B - 1;
  const uint32_t rbits = kp.rbits, smask = (1u << rbits) - 1u, dshift = rbits + SLIM_LOCAL_BITS;
  unsigned long long kept = 0;
  auto tile_start = [&](uint32_t t) { return (int64_t)t * RP_TILE; };
  auto tile_len = [&](uint32_t t) { return (uint32_t)min((int64_t)RP_TILE, n - (int64_t)t * RP_TILE); };

  ChunkRegs<1, RP_ROWS, PSRC> nxt;
  SlimCur<RP_ROWS> cur;
  uint32_t dr[RP_ROWS]; // digit << 16 | rank inside the tile's run of that digit; ~0 = row not kept
  uint32_t cur_tile = 0;
  // (`report`: false when the tile was loaded a second time — the pipeline re-reads a range's last tile instead of branching —
  //  so that a key outside the sampled range reaches the outlier list once)
  auto take_rows = [&](uint32_t len, uint32_t nxt_tile, bool report, SlimCur<RP_ROWS> &cur) { // nxt (as loaded: tile `nxt_tile`) -> cur
#pragma unroll
    for (int j = 0; j < RP_ROWS; j++) {
      bool keep = (uint32_t)(j * RP_WG) + threadIdx.x < len;
      if (PSRC == 1) keep = keep && row_passes(flt, nxt.a0[j]);
      if (PSRC == 3) keep = keep && row_passes(flt, nxt.pv[PSRC == 3 ? j : 0]);
      const uint64_t off = nxt.k[j] - kp.kmin;
      if (keep && off > kp.range) { // no bucket of the range partition holds this key (see rp_chunk_scatter_slim_kernel)
        if (kp.oob && report) key_out_of_range(kp, (uint32_t)((int64_t)nxt_tile * RP_TILE + (uint32_t)(j * RP_WG) + threadIdx.x));
        keep = false;
      }
      cur.off[j] = keep ? (uint32_t)off : 0xffffffffu;
      cur.a0[j] = nxt.a0[j];
    }
  };
  auto rank_row = [&](int j) {
    dr[j] = 0xffffffffu;
    if (cur.off[j] != 0xffffffffu) {
      const uint32_t d = min(cur.off[j] >> rbits, P - 1);
      dr[j] = (d << 16) | atomicAdd(&cnt[d], 1u);
    }
  };
  auto fill_sentinels = [&](uint32_t from, uint32_t count) {
    for (uint32_t i = 0; i < count; i++) slim_store(out.rows, (int64_t)from + i, SLIM_SENTINEL, 0ull);
  };
  auto scan_and_stage = [&]() {
    const uint32_t c = cnt[threadIdx.x];
    // what is left of an open block whose first tile is too far back for this tile's delta: sentinel rows
    if (owner && c > 0 && room > 0 && cur_tile - bbase > out.max_delta) {
      fill_sentinels(pos, room);
      room = 0;
    }
    // the claim first: its answer is needed only behind the staging loop
    const uint32_t used = min(c, room), need = c - used;
    const bool claim = owner && need > 0;
    const uint32_t k = (need + Bm1) & ~Bm1;
    uint32_t got = 0;
    if (claim && !dead) got = atomicAdd(&out.cursor[threadIdx.x], k);
    const uint32_t inc = wave_iscan_u32(c);
    if (lane_id() == 63) s_wsum[wave_id()] = inc;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
    for (int w = 0; w < RP_WG / 64; w++) {
      if (w < wave_id()) wbase += s_wsum[w];
      tot += s_wsum[w];
    }
    const uint32_t ls = wbase + inc - c;
    if (threadIdx.x == 0) {
      s_total = tot;
      kept += tot;
    }
    cnt[threadIdx.x] = ls; // run start (rank_row's counters are consumed)
    split[threadIdx.x] = ls + used;
    gb0[threadIdx.x] = (int64_t)pos - (int64_t)ls;
    dl0[threadIdx.x] = (cur_tile - bbase) << dshift; // (used > 0 only while the delta fits, see above)
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RP_ROWS; j++) {
      if (dr[j] == 0xffffffffu) continue;
      const uint32_t d = dr[j] >> 16;
      const uint32_t p = cnt[d] + (dr[j] & 0xffffu);
      const uint32_t local = (uint32_t)(j * RP_WG) + threadIdx.x;
      sv0[p] = cur.a0[j];
      sw[p] = (cur.off[j] & smask) | (local << rbits) | (p < split[d] ? dl0[d] : 0u);
      sdig[p] = (uint16_t)d;
    }
    int64_t g1 = 0;
    if (claim) {
      if (!dead && (uint64_t)got + k > (uint64_t)my_end) {
        dead = true;
        *out.flag = 1u;
      }
      if (dead) {
        g1 = DEAD;
        pos = 0;
        room = 0;
      } else {
        g1 = (int64_t)got - (int64_t)(ls + used);
        pos = got + need;
        room = k - need;
        bbase = cur_tile;
        for (uint32_t b = got >> out.log_b, be = (got + k) >> out.log_b; b < be; b++) out.blk_bt[b] = cur_tile;
      }
    } else {
      pos += c;
      room -= c;
    }
    gb1[threadIdx.x] = g1;
    __syncthreads();
  };
  auto store_row = [&](int j, uint32_t len) { // position p of the staged tile -> its block
    const uint32_t p = j * RP_WG + threadIdx.x;
    const uint32_t d = min((uint32_t)sdig[p], (uint32_t)(RP_WG - 1)); // (clamped, not masked: RP_WG = 768 is no power of two)
    int64_t g = (p < split[d] ? gb0[d] : gb1[d]) + p;
    if (p >= len || g < 0) g = sink + (int64_t)blockIdx.x * RP_WG + threadIdx.x; // past the staged rows / overflowed region: sink rows
    slim_store(out.rows, g, sw[p], sv0[p]);
  };

  if (t0 < t1) {
    uint32_t len = tile_len(t0);
    cur_tile = t0;
    rp_chunk_load<1, RP_WG, RP_ROWS, PSRC>(key, v0, nullptr, flt.col, tile_start(t0), len, nxt);
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
    take_rows(len, t0, true, cur);
    cnt[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RP_ROWS; j++) rank_row(j);
    __syncthreads();
    scan_and_stage();
    uint32_t staged_len = s_total;
    uint32_t tcur = min(t0 + 1, t1 - 1);
    len = tile_len(tcur);
    cur_tile = tcur;
    rp_chunk_load<1, RP_WG, RP_ROWS, PSRC>(key, v0, nullptr, flt.col, tile_start(tcur), len, nxt);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    take_rows(len, tcur, tcur != t0, cur);
    // the software pipeline of rp_chunk_scatter_slim_kernel: while tile i-1 (staged) is written out, tile i (in `cur`) is
    // ranked and tile i+1 is in flight into `nxt`.  (Measured and dropped, round 5: one tile MORE in flight — tile i+1
    // taken into a second compressed register set and the loads of tile i+2 issued before tile i is staged, 252 VGPRs, no
    // spill — so that the staging phase does not run with an idle memory pipe: 1.801 vs 1.804 ms for C4's 2e8 rows.  The
    // kernel is not waiting for its loads.)
    for (uint32_t ti = t0 + 1; ti < t1; ti++) {
      const uint32_t tnext = min(ti + 1, t1 - 1);
      const uint32_t nlen = tile_len(tnext);
      rp_chunk_load<1, RP_WG, RP_ROWS, PSRC>(key, v0, nullptr, flt.col, tile_start(tnext), nlen, nxt);
      cnt[threadIdx.x] = 0;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < RP_ROWS; j++) {
        if ((uint32_t)(j * RP_WG) < staged_len) store_row(j, staged_len);
        rank_row(j);
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
      scan_and_stage();
      staged_len = s_total;
      take_rows(nlen, tnext, tnext != ti, cur);
      cur_tile = tnext;
    }
#pragma unroll
    for (int j = 0; j < RP_ROWS; j++)
      if ((uint32_t)(j * RP_WG) < staged_len) store_row(j, staged_len);
  }
  if (owner && !dead) fill_sentinels(pos, room); // what is left of every open block
  if (threadIdx.x == 0 && kept) atomicAdd(out.kept, kept);
}

// Tile list of level 2 from the chunk table: chunk c of digit s contributes ceil(len / tile) tiles to
// segment s (any order of the chunks inside a segment).  Three small launches (count per digit -> prefix
// over <= 512 digits -> assign), <= a few hundred thousand chunks; a single workgroup doing all three took
// 165 us per C5 step.
// (The last tile of every (workgroup, digit) stream is partly filled; giving the level-2 workgroups tile
// ranges of equal ROW counts instead of equal tile counts was measured SLOWER, 4.24 -> 4.8 ms: a partly
// filled tile costs the pipeline as much as a full one.)
struct ChunkPlan {            // device scratch, zeroed before the count kernel
  uint32_t seg_tiles[512];    // tiles per digit
  uint32_t cursor[512];       // assign kernel: next free tile slot of the digit
  unsigned long long seg_rows[512];
};
__global__ __launch_bounds__(256) void rp_chunk_count_kernel(const uint32_t *__restrict__ chunk_len,
                                                             const uint32_t *__restrict__ chunk_dig,
                                                             const unsigned int *__restrict__ counter, uint32_t base_chunks,
                                                             uint32_t max_chunks, uint32_t tile, ChunkPlan *plan) {
  __shared__ uint32_t s_tiles[512];
  __shared__ unsigned long long s_rows[512];
  for (uint32_t i = threadIdx.x; i < 512; i += 256) {
    s_tiles[i] = 0;
    s_rows[i] = 0;
  }
  __syncthreads();
  const uint32_t nchunks = min(base_chunks + counter[0], max_chunks);
  for (uint32_t c = blockIdx.x * 256 + threadIdx.x; c < nchunks; c += gridDim.x * 256) {
    const uint32_t len = chunk_len[c];
    if (!len) continue;
    atomicAdd(&s_tiles[chunk_dig[c]], (len + tile - 1) / tile);
    atomicAdd(&s_rows[chunk_dig[c]], (unsigned long long)len);
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < 512; i += 256) {
    if (s_tiles[i]) atomicAdd(&plan->seg_tiles[i], s_tiles[i]);
    if (s_rows[i]) atomicAdd(&plan->seg_rows[i], s_rows[i]);
  }
}
__global__ __launch_bounds__(64) void rp_chunk_prefix_kernel(const ChunkPlan *plan, const unsigned int *__restrict__ counter,
                                                             uint32_t nseg, uint32_t digits2, int64_t *__restrict__ seg_start,
                                                             int64_t *__restrict__ seg_mat, uint32_t *__restrict__ seg_tiles,
                                                             uint32_t *__restrict__ seg_tile_base,
                                                             uint64_t *__restrict__ totals /* {tiles, rows, overflow} */) {
  if (threadIdx.x != 0) return;
  uint32_t tb = 0;
  int64_t rows = 0;
  for (uint32_t s = 0; s < nseg; s++) {
    seg_start[s] = rows;
    seg_tile_base[s] = tb;
    seg_mat[s] = (int64_t)tb * digits2;
    seg_tiles[s] = plan->seg_tiles[s];
    tb += plan->seg_tiles[s];
    rows += (int64_t)plan->seg_rows[s];
  }
  seg_start[nseg] = rows;
  totals[0] = tb;
  totals[1] = (uint64_t)rows;
  totals[2] = counter[1];
}
__global__ __launch_bounds__(256) void rp_chunk_assign_kernel(const uint32_t *__restrict__ chunk_len,
                                                              const uint32_t *__restrict__ chunk_dig,
                                                              const unsigned int *__restrict__ counter, uint32_t base_chunks,
                                                              uint32_t max_chunks, uint32_t digits2, uint32_t cap, uint32_t tile,
                                                              const uint32_t *__restrict__ seg_tiles,
                                                              const uint32_t *__restrict__ seg_tile_base, ChunkPlan *plan,
                                                              Tile *__restrict__ tiles, uint32_t *__restrict__ tile_chunk) {
  // a block claims the tile slots of its chunks with ONE global atomic per digit (its chunks rank themselves with
  // LDS atomics): one global atomic per chunk on <= 512 addresses took 96 us for the 9e4 chunks of a C5 step
  __shared__ uint32_t s_need[512], s_base[512];
  const uint32_t nchunks = min(base_chunks + counter[0], max_chunks);
  for (uint32_t c0 = blockIdx.x * 256; c0 < nchunks; c0 += gridDim.x * 256) {
    for (uint32_t i = threadIdx.x; i < 512; i += 256) s_need[i] = 0;
    __syncthreads();
    const uint32_t c = c0 + threadIdx.x;
    const uint32_t len = c < nchunks ? chunk_len[c] : 0;
    const uint32_t s = len ? chunk_dig[c] : 0, nt = (len + tile - 1) / tile;
    const uint32_t local = len ? atomicAdd(&s_need[s], nt) : 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 512; i += 256)
      if (s_need[i]) s_base[i] = atomicAdd(&plan->cursor[i], s_need[i]);
    __syncthreads();
    if (len) {
      const uint32_t base = seg_tile_base[s], i0 = s_base[s] + local;
      for (uint32_t q = 0; q < nt; q++) {
        Tile t;
        t.start = (int64_t)c * (cap + RP_CHUNK_SKEW) + (int64_t)q * tile;
        t.len = min(tile, len - q * tile);
        t.stride = seg_tiles[s];
        t.mat = (int64_t)base * digits2 + i0 + q;
        tiles[base + i0 + q] = t;
        if (tile_chunk) tile_chunk[base + i0 + q] = c; // (one tile per chunk when the chunk histograms are used)
      }
    }
    __syncthreads(); // (s_need / s_base are reused by the next trip)
  }
}
// count matrix of the next level (tile-major, what rp_hist_kernel writes) from the chunk histograms of an H2 level
__global__ void rp_hist_from_chunks_kernel(const uint32_t *__restrict__ hist, const uint32_t *__restrict__ tile_chunk,
                                           int64_t entries, uint32_t digits, uint32_t *__restrict__ mat) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= entries) return;
  const uint32_t t = (uint32_t)(i / digits), d = (uint32_t)(i % digits);
  mat[i] = hist[(size_t)tile_chunk[t] * digits + d];
}

// bucket b (level-1 digit d1 = b >> p2_bits ... ) start row, from the level's scanned matrix
__global__ void rp_bucket_starts_kernel(const uint32_t *__restrict__ offs,
                                        const int64_t *__restrict__ seg_mat,
                                        const uint32_t *__restrict__ seg_tiles,
                                        const int64_t *__restrict__ seg_start, uint32_t digits,
                                        uint32_t nseg, int64_t n, uint32_t *__restrict__ bstart) {
  int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t total = (int64_t)nseg * digits;
  if (b > total) return;
  if (b == total) {
    bstart[b] = (uint32_t)n;
    return;
  }
  uint32_t s = (uint32_t)(b / digits), d = (uint32_t)(b % digits);
  bstart[b] = seg_tiles[s] ? offs[seg_mat[s] + (int64_t)d * seg_tiles[s]] : (uint32_t)seg_start[s + 1];
}

// one block per segment: tile i of segment s
__global__ void rp_make_tiles_kernel(const int64_t *__restrict__ seg_start, const int64_t *__restrict__ seg_mat,
                                     const uint32_t *__restrict__ seg_tiles, const uint32_t *__restrict__ seg_tile_base,
                                     int rp_tile, Tile *__restrict__ tiles) {
  const uint32_t s = blockIdx.x, nt = seg_tiles[s], base = seg_tile_base[s];
  const int64_t start = seg_start[s], end = seg_start[s + 1], mat = seg_mat[s];
  for (uint32_t i = threadIdx.x; i < nt; i += blockDim.x) {
    Tile t;
    t.start = start + (int64_t)i * rp_tile;
    t.len = (uint32_t)min((int64_t)rp_tile, end - t.start);
    t.stride = nt;
    t.mat = mat + i;
    tiles[base + i] = t;
  }
}

} // namespace sq
