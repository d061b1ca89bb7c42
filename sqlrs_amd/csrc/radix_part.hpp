// radix_part.hpp — interface of the LDS-staged hash partitioner (radix_part.hip).
#pragma once

#include "common.hpp"
#include "prims.hpp"

namespace sq {

// Optional packing of (key, row id) into one 8-byte word: word = min(key - kmin, kmask) | row << kbits.
// Applies when the keys of interest span a known range [kmin, kmin + kmask) and kbits + bits(rows)
// <= 64; a key outside the range is stored as the sentinel offset kmask (no key of interest has
// it).  Saves the 4-byte row-id column in every partition pass and in the bucket pass.
//
// Dense keys (`dense` != 0, only together with packing): the keys of interest fill most of
// [kmin, kmin + range], so the partition is by key RANGE instead of by hash —
// bucket = min((key - kmin) >> rbits, P - 1) — and bucket b holds exactly the 2^rbits consecutive
// key offsets [b << rbits, (b + 1) << rbits): the bucket pass can address its LDS table directly
// by the low rbits of the offset (no key column in the table, no probing, 100 % fill).
struct KeyPack {
  uint64_t kmin = 0, kmask = 0;
  uint32_t kbits = 0; // 0 = not packed
  uint32_t dense = 0, rbits = 0;
  uint64_t range = 0; // dense: largest key offset of interest
  // Optimistic packing (range taken from a SAMPLE of the keys): the first pass that packs a row sets *oob when its
  // key lies outside [kmin, kmin + kmask) — the caller then discards the result and reruns with the exact range.
  // null = out-of-range keys are expected (fused join: no partner) or impossible (exact range).
  unsigned int *oob = nullptr;
  // ... unless the outliers are few: with `ov_rows` set, a row whose key lies outside the range is appended to that list
  // (local row ids; *ov_count rows so far, capacity ov_cap) and takes the caller's row route like a row that did not fit
  // its bucket table — an outlier the sample missed costs its own handling, not a second run; only a list that fills
  // up raises *oob
  unsigned long long *ov_count = nullptr;
  uint32_t *ov_rows = nullptr;
  uint32_t ov_cap = 0;
};
#if defined(__HIPCC__)
// a row whose key the (sampled) range does not cover: onto the outlier list, one atomic per wave; a full list -> *oob
__device__ __forceinline__ void key_out_of_range(const KeyPack &kp, uint32_t row) {
  if (!kp.ov_rows) {
    *kp.oob = 1u; // (a plain store, any number of writers)
    return;
  }
  const unsigned long long peers = __ballot(1); // the lanes that are here with an outlier of their own
  const int leader = __builtin_ctzll(peers);
  unsigned long long base = 0;
  if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(kp.ov_count, (unsigned long long)__popcll(peers));
  base = ((unsigned long long)(unsigned)__shfl((int)(base >> 32), leader, 64) << 32) | (unsigned)__shfl((int)base, leader, 64);
  const unsigned long long at = base + (unsigned long long)__popcll(peers & ((1ull << (threadIdx.x & 63)) - 1ull));
  if (at < kp.ov_cap) kp.ov_rows[at] = row;
  else *kp.oob = 1u;
}
__device__ __forceinline__ uint64_t pack_key_row(const KeyPack &kp, uint64_t key, uint32_t row) {
  uint64_t off = key - kp.kmin;
  // (uniform null test.  Range partitions: an offset between the range and the sentinel has no bucket of its own
  //  either — it would alias a slot of the last bucket)
  if (kp.oob && (off >= kp.kmask || (kp.dense && off > kp.range))) key_out_of_range(kp, row);
  return (off < kp.kmask ? off : kp.kmask) | ((uint64_t)row << kp.kbits);
}
__device__ __forceinline__ uint64_t packed_key(const KeyPack &kp, uint64_t w) { return (w & kp.kmask) + kp.kmin; }
// the key a row will carry once it is packed (out-of-range keys become the sentinel key): every
// pass of a packed partition must derive the row's bucket from THIS key, also the passes that
// still read the unpacked column — otherwise the first level counts and places a row by one key
// and writes it by another
__device__ __forceinline__ uint64_t packed_clamp(const KeyPack &kp, uint64_t key) {
  uint64_t off = key - kp.kmin;
  return (off < kp.kmask ? off : kp.kmask) + kp.kmin;
}
__device__ __forceinline__ uint64_t packed_off(const KeyPack &kp, uint64_t w) { return w & kp.kmask; }
__device__ __forceinline__ uint32_t packed_row(const KeyPack &kp, uint64_t w) { return (uint32_t)(w >> kp.kbits); }
#endif

// Rows of the slim form (see PartitionedRows::Slim): a value column + a 32-bit word column by default; -DSLIM_AOS =
// 12-byte {word, value} records, one dwordx3 access per row (compile-time A/B: tools/ab_two_builds.sh — measured: the
// partition levels gain 0.2 ms from the longer contiguous runs, the bucket pass loses 0.4 ms to the 12-byte lane stride)
struct SlimRec {
  uint32_t w, vlo, vhi;
};
struct SlimRowsView {
#ifdef SLIM_AOS
  SlimRec *rec = nullptr;
#else
  uint64_t *v = nullptr;
  uint32_t *w = nullptr;
#endif
};
#if defined(__HIPCC__)
__device__ __forceinline__ void slim_store(const SlimRowsView &r, int64_t g, uint32_t w, uint64_t v) {
#ifdef SLIM_AOS
  r.rec[g] = SlimRec{w, (uint32_t)v, (uint32_t)(v >> 32)};
#else
  r.v[g] = v;
  r.w[g] = w;
#endif
}
__device__ __forceinline__ void slim_load_nt(const SlimRowsView &r, int64_t i, uint32_t &w, uint64_t &v) {
#ifdef SLIM_AOS
  const SlimRec *p = r.rec + i;
  const uint32_t a = __builtin_nontemporal_load(&p->w), lo = __builtin_nontemporal_load(&p->vlo), hi = __builtin_nontemporal_load(&p->vhi);
  w = a;
  v = (uint64_t)lo | ((uint64_t)hi << 32);
#else
  v = __builtin_nontemporal_load(r.v + i);
  w = __builtin_nontemporal_load(r.w + i);
#endif
}
#endif

// A row predicate `col OP constant` evaluated by the FIRST partition pass itself (FilterExecutor
// directly below the partitioned operator, filter.rs:13-25): rows that fail are neither counted nor
// moved, so the filter's own read + compacted write of every column never happens.  `col` holds
// 8-byte values (int64 or float64, no NULLs); both sides are compared through their order-preserving
// u64 images (f64: IEEE total order, exactly like the stand-alone filter kernel of select.hip).
struct RowFilter {
  const uint64_t *col = nullptr; // nullptr = no filter
  uint32_t is_f64 = 0;
  uint32_t keep_mask = 7; // bit 0: keep rows with v < k, bit 1: v == k, bit 2: v > k
  uint64_t kord = 0;      // ordered image of the constant
};

#if defined(__HIPCC__)
} // namespace sq
#include "device_utils.hpp"
namespace sq {
__device__ __forceinline__ bool row_passes(const RowFilter &f, uint64_t bits) {
  const uint64_t o = f.is_f64 ? f64_to_ordered(__longlong_as_double((long long)bits)) : (bits ^ (1ull << 63));
  const uint32_t sel = o < f.kord ? 1u : (o == f.kord ? 2u : 4u);
  return (f.keep_mask & sel) != 0;
}
#endif
// `col OP constant` over an int64 / float64 column of `ib` without NULLs -> RowFilter; false = not that shape
// (hashagg_op.hip; shared by the fused join + aggregate and by sqlrs_hash_partition_filter)
bool fusable_row_filter(const Expr &e, InBatch &ib, RowFilter *rf);

struct PartitionInput {
  const uint64_t *keys = nullptr; // normalised keys (NKeys)
  const uint64_t *key_validity = nullptr;
  int64_t n = 0;
  int nv = 0; // 8-byte value columns carried along (0..2)
  const void *vals[2] = {nullptr, nullptr};
  const uint64_t *val_validity[2] = {nullptr, nullptr};
  bool build_side = false; // profiling label only
  KeyPack pack;            // used only when no column is nullable
  RowFilter filter;        // only taken by the chunked first level (partition_rows returns false otherwise)
};

// Rows in bucket order: bucket b = rows [bstart[b], bstart[b+1]).  `idx` = original row,
// `flags` (null when nothing is nullable) bit0 key valid, bit1 v0 valid, bit2 v1 valid.
// bucket(key) = mulhi(mix64(key), P); NULL keys -> bucket 0.  (Dense KeyPack: key range, see above.)
struct PartitionedRows {
  int64_t n = 0; // rows in bucket order (= input rows that passed PartitionInput::filter)
  uint32_t P = 0;
  BufP key, v0, v1, idx, flags;
  BufP rec;    // packed dense partitions with one value column: {key|row word, value} records; key / v0 are null then
  BufP bstart; // u32[P + 1]
  std::vector<uint32_t> bstart_host; // the same on the host
  // Claimed single level (radix_part.hip): bucket b = SLOTS [bstart_host[b], bend_host[b]) of its own region — rows
  // and sentinel rows (all-ones key|row word: a key outside the packed range, skipped by the bucket pass); the
  // regions are not adjacent.  Empty = the buckets are contiguous runs of rows (bucket b ends where b + 1 starts).
  std::vector<uint32_t> bend_host;
  uint32_t bucket_end(uint32_t b) const { return bend_host.empty() ? bstart_host[b + 1] : bend_host[b]; }
  KeyPack pack; // kbits != 0: `key` holds packed (key, row) words and `idx` is null
  // Slim form (radix_part.hip, "slim records"; dense packed partitions with one value column): 12 bytes per row —
  // an 8-byte value + a 32-bit word = slot in the bucket | row inside its level-1 tile << rbits | tile delta << (rbits + 13);
  // row id = (nzbt of the run the row sits in + tile delta) * tile + row inside the tile.  key / v0 / rec are null.
  struct Slim {
    bool on = false;
    BufP buf0, buf1;    // the rows in bucket order (values, words; SLIM_AOS: records, null)
    SlimRowsView rows;  // device view of buf0 / buf1
    BufP nzstart, nzbt; // u32 per run: first row / base tile of the k-th non-empty run of bucket b at [bcol[b] + k]
    BufP nzcount, bcol; // u32 [P]
    uint32_t tile = 0;  // rows per level-1 tile
    // Claimed single level in the slim form (rp_claim_scatter_slim_kernel): no run lists — the base tile of a row is that
    // of its BLOCK, blk_bt[slot >> log_b]; sentinel rows carry the word 0xffffffff.  Null = the run lists above.
    BufP blk_bt;
    uint32_t log_b = 0;
  } slim;
};

// P_wanted <= 65536; the actual bucket count (>= P_wanted) is returned in out->P.
bool partition_rows(Ctx *ctx, const PartitionInput &in, uint32_t P_wanted, PartitionedRows *out);

} // namespace sq
