// agg_partition.hip — LDS-partitioned pre-aggregation for large HashAgg batches.
//
// Why: MI355X executes ~24 G random global atomics/s but ~1500 G LDS atomics/s
// (profiles/r01_ubench_mi355x.txt), so a 2e8-row group-by done with global atomics is
// 60x off the HBM roofline.  Rows are therefore radix-partitioned by key hash into P buckets
// small enough that each bucket's groups fit one workgroup's LDS hash table:
//
//   key_stats : HyperLogLog (sampled) + min/max of the keys -> estimated group count -> P,
//               and whether (key, row id) can be packed into one word               ( 8 B/row read)
//   partition : radix_part.hip, one or two levels of LDS-staged multi-split: rows -> (key|row,
//               values) in bucket order                       (per level: 8 B hist + 32 B scatter)
//   lds_agg   : one workgroup per bucket: open-addressing table in LDS (64-bit ds CAS to claim a
//               slot, ds_add/ds_min/ds_max on the accumulator cells), then the occupied slots
//               are written out as (key, first row, accumulators)                  (16 B/row read)
//
// While nothing else has been pushed the batch's groups ARE the operator state (hashagg_op.hip
// emits them directly); otherwise they are merged into the global table by the ordinary resolve
// path (agg.hip) with explicit first-row ids and pre-aggregated weights: O(groups) atomics.
// Rows whose bucket table overflows (estimate too low) are returned to the caller and take
// the resolve path directly, so the result never depends on the estimate.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "agg_partition.hpp"
#include "device_utils.hpp"
#include "radix_part.hpp"

namespace sq {

#ifndef PART_WG_N
#define PART_WG_N 512
#endif
constexpr int PART_WG = PART_WG_N;
// The direct-addressed bucket passes (lds_agg_dense_kernel, lds_agg_dense_slim_kernel) run 1024-thread workgroups: their table
// fills the CU's LDS, so ONE workgroup is resident, and at 512 threads that is two waves per SIMD — too few to overlap a wave's
// load wait with another's LDS atomics.  Sixteen waves on the same table (60-81 VGPRs, no spill): C5 bucket pass 1.64 -> 1.42-1.48 ms,
// sorted fact rows 2.81 -> 2.38, C4 0.447 -> 0.432 (profiles/r05n_bucket_wg_ab.txt, r05o: one process, three builds).  The probing
// kernel (two 72 KiB tables per CU already) got SLOWER at 1024 (sparse-key C5: 2.52 -> 2.81 ms) and keeps 512.
#ifndef DENSE_WG_N
#define DENSE_WG_N 1024
#endif
constexpr int DENSE_WG = DENSE_WG_N;
#ifndef HOT_MIN_PEERS_N
#define HOT_MIN_PEERS_N 8
#endif
constexpr int HOT_MIN_PEERS = HOT_MIN_PEERS_N; // lanes of a wave on one slot from which they are reduced across the wave first
constexpr uint64_t LDS_EMPTY = ~0ull;

__device__ __forceinline__ uint32_t bucket_of(uint64_t h, uint32_t P) {
  return (uint32_t)__umul64hi(h, (uint64_t)P);
}

// ---------------------------------------------------------------- key statistics --
// HyperLogLog registers + min / max of the keys' signed-order image (hll[8192..8195] as two u64).
// Min / max see every row; the HyperLogLog registers only every 2^sample_shift-th 64-row group
// (the hash + LDS atomic per row is what made this kernel slower than a plain read).
// block_stride > 1: only every block_stride-th chunk of PART_WG * KU rows is read at all (plus the first and the
// last chunk: sorted / clustered keys have their extremes there) — min / max are then a SAMPLE's, see
// estimate_distinct.
// hll_bits: log2 of the number of registers in use (<= 12).  The block-sampled pass uses 1024: every block merges its
// registers with one global atomicMax each (24 G/s), and 512 blocks x 4096 registers cost more than reading the sample.
// TWO register sets, hll[0..4096) and hll[4096..8192), each fed by HALF of the sample (block-sampled pass: the chunks of
// the odd / the even workgroups; sample_shift > 0: the 64-row groups 0 and 2^(sample_shift-1) of every 2^sample_shift):
// the host estimates the union and one half, and a half that holds clearly fewer keys than the union says the sample has
// not seen every group yet — keys that arrive ORDERED or clustered, where a sample of the ROWS is no sample of the
// GROUPS (2e8 sorted rows of 1e6 groups: an eighth of the chunks holds an eighth of the groups; the tables were sized
// for 1.5e5 groups and the batch took the overflow path, 250 ms instead of 2.2 — round 5).
__global__ __launch_bounds__(PART_WG) void key_stats_kernel(const uint64_t *__restrict__ keys,
                                                            const uint64_t *__restrict__ validity,
                                                            int64_t n, int sample_shift, int block_stride, int hll_bits,
                                                            unsigned int *__restrict__ hll) {
  __shared__ unsigned int reg[2 * 4096];
  const int nreg = 1 << hll_bits;
  for (int i = threadIdx.x; i < 2 * 4096; i += PART_WG) reg[i] = 0;
  __syncthreads();
  uint64_t kmin = ~0ull, kmax = 0;
  const int64_t smask = (1ll << sample_shift) - 1, shalf = sample_shift ? 1ll << (sample_shift - 1) : -1;
  const unsigned blk_set = block_stride > 1 && (blockIdx.x & 1) ? 4096u : 0u; // (block-sampled: the workgroup's set)
  constexpr int KU = 12; // loads in flight per lane (the loaded HBM latency needs ~100 KiB per CU)
  const int64_t nchunks = (n + PART_WG * KU - 1) / (PART_WG * KU);
  const int64_t nsel = block_stride > 1 ? (nchunks + block_stride - 1) / block_stride + 1 : nchunks;
  for (int64_t ci = blockIdx.x; ci < nsel; ci += gridDim.x) {
    const int64_t chunk = block_stride > 1 ? min(ci * block_stride, nchunks - 1) : ci; // (the extra index = the last chunk)
    const int64_t base = chunk * (int64_t)(PART_WG * KU) + threadIdx.x;
    uint64_t k[KU];
#pragma unroll
    for (int u = 0; u < KU; u++) k[u] = __builtin_nontemporal_load(keys + min(base + u * PART_WG, n - 1));
#pragma unroll
    for (int u = 0; u < KU; u++) {
      const int64_t r = base + u * PART_WG;
      if (r >= n) continue;
      if (validity && !((validity[r >> 6] >> (r & 63)) & 1)) continue;
      const uint64_t o = k[u] ^ (1ull << 63);
      kmin = min(kmin, o);
      kmax = max(kmax, o);
      const int64_t grp = (r >> 6) & smask; // wave-uniform
      if (grp != 0 && grp != shalf) continue;
      uint64_t h = mix64(k[u] ^ 0x2545f4914f6cdd1dULL);
      unsigned idx = (unsigned)(h >> (64 - hll_bits)) + (grp == shalf ? 4096u : blk_set);
      unsigned rank = (unsigned)__builtin_clzll((h << hll_bits) | (1ull << (hll_bits - 1))) + 1;
      if (reg[idx] < rank) atomicMax(&reg[idx], rank);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nreg; i += PART_WG) {
    if (reg[i]) atomicMax(&hll[i], reg[i]);
    if (reg[4096 + i]) atomicMax(&hll[4096 + i], reg[4096 + i]);
  }
  // one atomic pair per BLOCK: atomics on one address are serialised in L2, and a pair per wave (8192 of them on two
  // addresses) was most of the sampled pass's 0.125 ms
  __shared__ unsigned long long s_mm[2][PART_WG / 64];
  kmin = wave_min_u64(kmin);
  kmax = wave_max_u64(kmax);
  if (lane_id() == 0) {
    s_mm[0][wave_id()] = kmin;
    s_mm[1][wave_id()] = kmax;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < PART_WG / 64; w++) {
      kmin = min(kmin, (uint64_t)s_mm[0][w]);
      kmax = max(kmax, (uint64_t)s_mm[1][w]);
    }
    unsigned long long *mm = (unsigned long long *)(hll + 8192);
    atomicMin(mm, (unsigned long long)kmin);
    atomicMax(mm + 1, (unsigned long long)kmax);
  }
}

// (returns the estimate over the whole sample; *half = the estimate over one half of it, see key_stats_kernel)
static double hll_pass(Ctx *ctx, const uint64_t *keys, const uint64_t *validity, int64_t n, int sample_shift,
                       uint64_t *omin, uint64_t *omax, int block_stride = 1, double *half = nullptr) {
  const int hll_bits = block_stride > 1 ? 10 : 12; // (1024 registers: 3 % standard error, the estimate only sizes tables)
  BufP hll = ctx->alloc_zero(8192 * 4 + 16);
  SQ_HIP(hipMemsetAsync(hll->as<uint8_t>() + 8192 * 4, 0xff, 8, ctx->stream)); // min starts at ~0
  {
    ProfScope ps(ctx, "key_stats");
    // two blocks per CU: every block ends with up to 4096 global atomicMax (24 G/s on MI355X), so
    // 2048 blocks spent 0.33 ms merging their registers — more than reading 1.6 GB of keys
    unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(n, (int64_t)PART_WG * 16 * block_stride), 2 * (int64_t)ctx->num_cus);
    key_stats_kernel<<<dim3(std::max(blocks, 1u)), dim3(PART_WG), 0, ctx->stream>>>(keys, validity, n, sample_shift,
                                                                                  block_stride, hll_bits, hll->as<unsigned int>());
    SQ_HIP(hipGetLastError());
  }
  // (through the pinned staging buffer: a copy into pageable memory cost ~50 us of host time before the partition)
  const unsigned int *hreg = (const unsigned int *)ctx->fetch(hll->p, 8192 * 4 + 16);
  if (omin) std::memcpy(omin, hreg + 8192, 8);
  if (omax) std::memcpy(omax, hreg + 8194, 8);
  const int nreg = 1 << hll_bits;
  const double m = (double)nreg;
  auto estimate = [&](bool both) {
    double sum = 0;
    int zeros = 0;
    for (int i = 0; i < nreg; i++) {
      const unsigned r = both ? std::max(hreg[i], hreg[4096 + i]) : hreg[i];
      sum += std::ldexp(1.0, -(int)r);
      zeros += (r == 0);
    }
    double e = (0.7213 / (1.0 + 1.079 / m)) * m * m / sum;
    if (e <= 2.5 * m && zeros) e = m * std::log(m / zeros);
    return e;
  };
  const double e_half = estimate(false), e = estimate(true);
  if (half) *half = e_half;
  return e;
}

// Distinct keys of the batch.  A sample of the 64-row groups (two of every eight) is enough while every
// group is seen several times in it; when the sample looks like mostly distinct keys — or one half of it holds
// clearly fewer keys than the whole: ordered / clustered keys, see key_stats_kernel — the full
// pass decides (the partition route is then usually rejected anyway).  A low estimate is not a
// correctness problem: rows that do not fit their bucket table take the overflow path.
//
// `sampled` (optimistic statistics, large batches without a NULL bitmap): only every eighth 6144-row chunk is READ
// (an eighth of the pass: C4 0.33 -> 0.05 ms), so min / max are the sample's — the caller widens them, packs with the
// widened range and lets the first partition pass report any key outside it (KeyPack::oob), which costs one rerun
// with the exact pass.  The estimate is scaled up a little: a sample misses groups that occur a handful of times.
double estimate_distinct(Ctx *ctx, const uint64_t *keys, const uint64_t *validity, int64_t n,
                         uint64_t *omin, uint64_t *omax, bool *sampled) {
  if (sampled) *sampled = false;
  static const int opt_env = [] { // test / tuning hook: 0 = always the exact pass
    const char *e = hook("SQLRS_SAMPLED_STATS");
    return e ? std::atoi(e) : 1;
  }();
  constexpr double SATURATED = 0.8; // half the sample holds >= this share of the sample's keys: every group has been seen
  // Between DISJOINT (ordered / clustered rows: a half holds half of the sample's keys) and SATURATED lie heavy-tailed keys —
  // Zipf(1.1) over 1e6 groups: the halves share the frequent keys and each has its own rare ones — where the sample is short
  // of the rare groups only: the estimate is scaled by the missing share instead of hashing every row (C4 Zipf: the exact pass
  // cost 0.34 of 2.97 ms); SQLRS_STATS_DISJOINT (read per call) moves the line
  double DISJOINT = 0.6;
  if (const char *dj = hook("SQLRS_STATS_DISJOINT")) DISJOINT = std::atof(dj);
  double half = 0;
  if (sampled && opt_env && !validity && n >= (1ll << 24)) {
    const int stride = 8;
    const double e = hll_pass(ctx, keys, validity, n, 0, omin, omax, stride, &half);
    if (e <= 0.2 * (double)(n / stride) && half >= DISJOINT * e) { // every group is seen several times in the sample (or only rare ones are missed): the estimate stands
      *sampled = true;
      return half >= SATURATED * e ? e * 1.25 : e * (1.25 + 2.5 * (SATURATED - half / e)); // (0.8 -> 1.25x ... 0.6 -> 1.75x)
    }
    if (half < DISJOINT * e) return hll_pass(ctx, keys, validity, n, 0, omin, omax); // ordered keys: every row decides
  }
  const int shift = n >= (1ll << 22) ? 3 : 0;
  double e = hll_pass(ctx, keys, validity, n, shift, omin, omax, 1, &half);
  if (shift && (e > 0.2 * (double)(n >> shift) || half < SATURATED * e)) e = hll_pass(ctx, keys, validity, n, 0, omin, omax);
  return e;
}

// ----------------------------------------------------------------- LDS aggregate --
// accumulator code = kind | (value column << 3)
enum AccKind { AK_COUNT = 0, AK_SUM_I64 = 1, AK_SUM_F64 = 2, AK_MIN_I64 = 3, AK_MIN_F64 = 4, AK_MAX_I64 = 5, AK_MAX_F64 = 6 };

struct LdsAggParams {
  int n_acc;
  int code[PART_MAX_ACC]; // AccKind | src << 3
  uint32_t cap;           // slots (power of two); +2 reserved slots follow
  // dense (direct-addressed) fused join whose build keys do not cover their whole range: bit (key offset) = the key
  // has a build partner; null = every key of the range has one
  const unsigned long long *partner_bits;
  // dense fused join over DUPLICATE build keys: build rows per key offset (0 = no partner); COUNT / SUM cells of a slot
  // are multiplied by it when the slot is emitted
  const unsigned int *partner_mult;
  int seg_off; // SQLRS_AGG_SEG=0 (read per call): no per-run adds for ordered rows (lds_agg_dense_slim_kernel; A/B)
};

// cell of accumulator `kind` for a key that has `m` build rows (every probe row = m joined rows)
__device__ __forceinline__ unsigned long long acc_times(int kind, unsigned long long cell, unsigned int m) {
  switch (kind) {
  case 0 /* AK_COUNT */:
  case 1 /* AK_SUM_I64 */: return cell * (unsigned long long)m; // (wrapping, like m additions)
  case 2 /* AK_SUM_F64 */: return (unsigned long long)__double_as_longlong(__longlong_as_double((long long)cell) * (double)m);
  default: return cell;
  }
}

__device__ __forceinline__ uint64_t acc_identity_cell(int kind) {
  return (kind == AK_MIN_I64 || kind == AK_MIN_F64) ? ~0ull : 0ull;
}

// slot hash inside a bucket.  The bucket itself is chosen by the high bits of mix64(key), so any
// cheap function of the key is independent of it; two 32-bit multiplies instead of mix64's two
// 64-bit ones (the kernel is bound by instruction issue, not by LDS or HBM).
__device__ __forceinline__ uint32_t slot_hash(uint64_t key) {
  uint32_t x = (uint32_t)key ^ ((uint32_t)(key >> 32) * 0x85ebca6bu);
  x *= 0x9e3779b1u;
  return x ^ (x >> 15);
}

__device__ __forceinline__ void acc_apply(int kind, unsigned long long *c, uint64_t v) {
  switch (kind) {
  case AK_COUNT: atomicAdd(c, 1ull); break;
  case AK_SUM_I64: atomicAdd(c, (unsigned long long)v); break;
  case AK_SUM_F64: unsafeAtomicAdd((double *)c, __longlong_as_double((long long)v)); break;
  case AK_MIN_I64: atomicMin(c, (unsigned long long)i64_to_ordered((int64_t)v)); break;
  case AK_MIN_F64: atomicMin(c, (unsigned long long)f64_to_ordered(__longlong_as_double((long long)v))); break;
  case AK_MAX_I64: atomicMax(c, (unsigned long long)i64_to_ordered((int64_t)v)); break;
  default: atomicMax(c, (unsigned long long)f64_to_ordered(__longlong_as_double((long long)v)));
  }
}

// rows of one trip (LDS_U rows per thread) held in registers
#ifndef LDS_U_ROWS
#define LDS_U_ROWS 4
#endif
constexpr int LDS_U = LDS_U_ROWS;
template <int NV> struct AggRows {
  uint64_t k[LDS_U], v0[NV >= 1 ? LDS_U : 1], v1[NV >= 2 ? LDS_U : 1];
  uint32_t id[LDS_U];
  uint8_t f[LDS_U];
};

// every lane loads (rows past `hi` re-read row hi-1), so all loads of a trip issue back to back
template <int NV, bool FLAGS, bool PACK, int WG = PART_WG>
__device__ __forceinline__ void lds_agg_load(const uint64_t *__restrict__ pk, const uint32_t *__restrict__ pi,
                                             const uint64_t *__restrict__ pv0, const uint64_t *__restrict__ pv1,
                                             const uint8_t *__restrict__ pf, int64_t i0, int64_t hi,
                                             AggRows<NV> &r) {
#pragma unroll
  for (int u = 0; u < LDS_U; u++) {
    int64_t i = min(i0 + (int64_t)u * WG, hi - 1);
    r.k[u] = __builtin_nontemporal_load(pk + i);
    if (!PACK) r.id[u] = pi ? __builtin_nontemporal_load(pi + i) : (uint32_t)i; // no id column: rows in place
    if (NV >= 1) r.v0[u] = __builtin_nontemporal_load(pv0 + i);
    if (NV >= 2) r.v1[u] = __builtin_nontemporal_load(pv1 + i);
    r.f[u] = FLAGS ? pf[i] : 7;
  }
}

// the same from {key|row word, value 0} records (radix_part.hpp PartitionedRows::rec): one 16-byte load per row
template <int NV, int WG = PART_WG>
__device__ __forceinline__ void lds_agg_load_rec(const uint64_t *__restrict__ prec, int64_t i0, int64_t hi, AggRows<NV> &r) {
#pragma unroll
  for (int u = 0; u < LDS_U; u++) {
    int64_t i = min(i0 + (int64_t)u * WG, hi - 1);
    const u64x2 t = __builtin_nontemporal_load((const u64x2 *)prec + i);
    r.k[u] = t.x;
    r.v0[0 + (NV >= 1 ? u : 0)] = t.y;
    r.f[u] = 7;
  }
}

// global tables of the split (skewed) buckets: nsplit tables of nslots = cap + 2 slots
struct SplitTables {
  unsigned long long *key = nullptr;  // [nsplit][nslots], LDS_EMPTY
  unsigned int *first = nullptr;      // [nsplit][nslots], ~0
  unsigned long long *acc = nullptr;  // [n_acc][nsplit][nslots], identity
  uint32_t nsplit = 0;
};

__global__ void split_init_kernel(SplitTables stb, int64_t total, int n_acc, LdsAggParams prm) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  stb.key[i] = LDS_EMPTY;
  stb.first[i] = 0xffffffffu;
  for (int a = 0; a < n_acc; a++) stb.acc[(size_t)a * total + i] = acc_identity_cell(prm.code[a] & 7);
}

// occupied slots of the split buckets' tables -> group list (same layout as lds_agg_kernel's output)
__global__ void split_emit_kernel(SplitTables stb, uint32_t nslots, uint32_t cap, int n_acc,
                                  unsigned long long *out_count, uint64_t *__restrict__ gkey,
                                  uint32_t *__restrict__ gfirst, uint8_t *__restrict__ gvalid,
                                  uint64_t *__restrict__ gacc, int64_t gcap) {
  const int64_t total = (int64_t)stb.nsplit * nslots;
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  unsigned int first = stb.first[i];
  if (first == 0xffffffffu) return;
  const uint32_t s = (uint32_t)(i % nslots);
  unsigned long long base = atomicAdd(out_count, 1ull);
  if ((int64_t)base >= gcap) return;
  gkey[base] = s == cap + 1 ? LDS_EMPTY : stb.key[i];
  gfirst[base] = first;
  if (gvalid) gvalid[base] = s == cap ? 0 : 1;
  for (int a = 0; a < n_acc; a++) gacc[(size_t)a * gcap + base] = stb.acc[(size_t)a * total + i];
}

// One workgroup per work item (a bucket, or a chunk of a skewed bucket).
//
// Table layout in LDS (structure of arrays, nslots = cap + 2; slot cap = NULL key, cap + 1 = the
// key whose value is the EMPTY marker):  key[nslots] u64 | acc[n_acc][nslots] u64 | first[nslots] u32.
// 64 lanes probing random slots touch 32 different bank pairs this way; with 32-byte
// array-of-struct slots they fell on 8 (SQ_LDS_BANK_CONFLICT was 60 % of the LDS cycles).
//
// NACC >= 0: the accumulator list is a compile-time constant (C0, C1 = codes of accumulators
// 0 and 1), which removes the per-row interpreter (loop + switch over prm.code); NACC < 0 reads
// it from `prm`.  JOIN: fused inner join, the bucket's build keys are inserted first and probe
// rows only accumulate into slots that exist.
// PACK: `pk` holds packed (key, row) words (radix_part.hpp KeyPack) and there is no `pi` column.
// REC (PACK, one value column, nothing nullable): `pk` holds {key|row word, value} records, pv0 is unused.
template <int NV, bool FLAGS, bool JOIN, int NACC, int C0, int C1, bool PACK, bool REC = false>
__global__ __launch_bounds__(PART_WG) void lds_agg_kernel(
    LdsAggParams prm, const uint64_t *__restrict__ pk, const uint32_t *__restrict__ pi,
    const uint64_t *__restrict__ pv0, const uint64_t *__restrict__ pv1,
    const uint8_t *__restrict__ pf, const uint32_t *__restrict__ work /* {bucket, lo, hi} triples */, uint32_t P,
    int64_t n, unsigned long long *out_count, uint64_t *__restrict__ gkey,
    uint32_t *__restrict__ gfirst, uint8_t *__restrict__ gvalid, uint64_t *__restrict__ gacc,
    int64_t gcap, unsigned long long *ov_count, uint32_t *__restrict__ ov_rows,
    const uint64_t *__restrict__ bk, const uint8_t *__restrict__ bf, const uint32_t *__restrict__ bbstart,
    KeyPack kp, SplitTables stb) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long tab[];
  __shared__ unsigned int s_cnt, s_join_fail;
  __shared__ unsigned long long s_base;
  const uint32_t b = work[4 * blockIdx.x];
  const int64_t lo = work[4 * blockIdx.x + 1];
  const int64_t hi = work[4 * blockIdx.x + 2];
  const uint32_t split = work[4 * blockIdx.x + 3]; // index of the bucket's global table, or ~0: not split
  const uint32_t cap = prm.cap, mask = cap - 1, nslots = cap + 2;
  const int n_acc = NACC >= 0 ? NACC : prm.n_acc;
  auto code_of = [&](int a) { return NACC >= 0 ? (a == 0 ? C0 : C1) : prm.code[a]; };
  unsigned long long *tkey = tab;
  unsigned long long *tacc = tab + nslots;
  unsigned int *tfirst = (unsigned int *)(tacc + (size_t)n_acc * nslots);
  AggRows<NV> cur, nxt;
  if (lo < hi) { // in flight during the set-up
    if (REC) lds_agg_load_rec<NV>(pk, lo + threadIdx.x, hi, cur);
    else lds_agg_load<NV, FLAGS, PACK>(pk, pi, pv0, pv1, pf, lo + threadIdx.x, hi, cur);
  }
  for (uint32_t s = threadIdx.x; s < nslots; s += PART_WG) {
    tkey[s] = LDS_EMPTY;
    tfirst[s] = 0xffffffffu;
#pragma unroll
    for (int a = 0; a < PART_MAX_ACC; a++) {
      if (a >= n_acc) break;
      tacc[(size_t)a * nslots + s] = acc_identity_cell(code_of(a) & 7);
    }
  }
  if (threadIdx.x == 0) {
    s_cnt = 0;
    s_join_fail = 0;
  }
  __syncthreads();
  if (JOIN) {
    // (a failed attempt is abandoned as a whole: once any workgroup has reported, the rest return at once)
    if (*(volatile unsigned long long *)(ov_count + 1) != 0) return;
    const int64_t blo = bbstart[b], bhi = bbstart[b + 1];
    for (int64_t i = blo + threadIdx.x; i < bhi; i += PART_WG) {
      uint64_t key = bk[i];
      bool valid = bf ? (bf[i] & 1) : true;
      // a key seen twice = duplicate build keys: the caller's pre-condition does not hold (it may not have been
      // established yet, join_state.hpp `unique_known`): flag it, the caller falls back to the composed route
      if (!valid) {
        if (atomicExch(&tkey[cap], 1ull) != LDS_EMPTY) s_join_fail = 1; // NULL build key present (NULL = NULL matches)
      } else if (key == LDS_EMPTY) {
        if (atomicExch(&tkey[cap + 1], 1ull) != LDS_EMPTY) s_join_fail = 1;
      } else {
        uint32_t s = slot_hash(key) & mask;
        uint32_t probes = 0;
        while (true) {
          unsigned long long prev = atomicCAS(&tkey[s], LDS_EMPTY, (unsigned long long)key);
          if (prev == key) s_join_fail = 1;
          if (prev == LDS_EMPTY || prev == key) break;
          s = (s + 1) & mask;
          if (++probes >= cap) {
            s_join_fail = 1; // table too small for the build side: caller falls back
            break;
          }
        }
      }
    }
    __syncthreads();
    // one report per workgroup (every duplicate used to exchange the global flag itself: 5e5 duplicate keys = 6 ms of
    // same-address atomics in an attempt that was going to be thrown away), and nothing more to do for this bucket
    if (s_join_fail) {
      if (threadIdx.x == 0) atomicExch(ov_count + 1, 1ull);
      return;
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): same entry state as the loop's back edge
  for (int64_t base = lo; base < hi; base += (int64_t)LDS_U * PART_WG) {
    const int64_t i0 = base + threadIdx.x;
    // next trip's rows (the last trip re-reads the last rows of the bucket)
    if (REC) lds_agg_load_rec<NV>(pk, i0 + (int64_t)LDS_U * PART_WG, hi, nxt);
    else lds_agg_load<NV, FLAGS, PACK>(pk, pi, pv0, pv1, pf, i0 + (int64_t)LDS_U * PART_WG, hi, nxt);
    // first probe of all LDS_U rows: the table reads are independent and issue together
    uint32_t slot[LDS_U];
    unsigned long long seen[LDS_U];
    uint32_t sentinel = 0; // bit u: row u carries the sentinel offset (its key lay outside the packed range: an outlier that
                           // went to the overflow list, radix_part.hpp key_out_of_range) — not a key of this bucket
    if (PACK) {
#pragma unroll
      for (int u = 0; u < LDS_U; u++) {
        if (!JOIN && packed_off(kp, cur.k[u]) == kp.kmask) sentinel |= 1u << u;
        cur.id[u] = packed_row(kp, cur.k[u]);
        cur.k[u] = packed_key(kp, cur.k[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < LDS_U; u++) {
      uint32_t s = slot_hash(cur.k[u]) & mask;
      if (FLAGS && !(cur.f[u] & 1)) s = cap;            // NULL keys: one group
      else if (cur.k[u] == LDS_EMPTY) s = cap + 1;
      slot[u] = s;
      seen[u] = tkey[s];
    }
#pragma unroll
    for (int u = 0; u < LDS_U; u++) {
      bool act = i0 + (int64_t)u * PART_WG < hi && !((sentinel >> u) & 1u); // this lane's row contributes to slot s
      const uint64_t key = cur.k[u];
      uint32_t s = slot[u];
      unsigned long long c = seen[u];
      if (act) {
        if (s >= cap) { // reserved slots: no probing
          if (JOIN && c == LDS_EMPTY) act = false;
        } else if (JOIN) {
          uint32_t probes = 0;
          while (c != key && c != LDS_EMPTY && ++probes < cap) {
            s = (s + 1) & mask;
            c = tkey[s];
          }
          if (c != key) act = false; // no build partner
        } else {
          uint32_t probes = 0;
          while (true) {
            if (c == key) break;
            if (c == LDS_EMPTY) {
              unsigned long long prev = atomicCAS(&tkey[s], LDS_EMPTY, (unsigned long long)key);
              if (prev == LDS_EMPTY || prev == key) break;
            }
            s = (s + 1) & mask;
            if (++probes >= cap) { // table full: hand the row back to the caller
              unsigned long long o = atomicAdd(ov_count, 1ull);
              ov_rows[o] = cur.id[u];
              act = false;
              break;
            }
            c = tkey[s];
          }
        }
      }
      // Hot keys: when many lanes of the wave hit the slot of the first active lane, they are
      // reduced across the wave and ONE lane issues the atomics — 64 lanes on one LDS address
      // serialise (Zipf(1.1) keys: 5.6 ms instead of 0.75 ms for the uniform 2e8-row C4 batch).
      const uint64_t actm = __ballot(act);
      if (actm) {
        const int first = __builtin_ctzll(actm);
        const uint32_t s0 = (uint32_t)__shfl((int)s, first, 64);
        const bool hot = act && s == s0;
        const uint64_t peers = __ballot(hot);
        if (__popcll(peers) >= HOT_MIN_PEERS) {
          const uint32_t idmin = wave_min_u32_dpp(hot ? cur.id[u] : 0xffffffffu);
          if (lane_id() == first) atomicMin(&tfirst[s0], idmin);
#pragma unroll
          for (int a = 0; a < PART_MAX_ACC; a++) {
            if (a >= n_acc) break;
            const int code = code_of(a), kind = code & 7, src = code >> 3;
            const bool in = hot && (!FLAGS || (cur.f[u] & (2 << src)));
            const uint64_t v = (NV >= 2 && src) ? cur.v1[u] : cur.v0[u];
            unsigned long long *cell = tacc + (size_t)a * nslots + s0;
            const uint64_t inm = __ballot(in);
            uint64_t red;
            switch (kind) {
            case AK_COUNT: red = (uint64_t)__popcll(inm); break;
            case AK_SUM_I64: red = wave_sum_u64_dpp(in ? v : 0ull); break;
            case AK_SUM_F64: red = (uint64_t)__double_as_longlong(wave_sum_f64_dpp(in ? __longlong_as_double((long long)v) : 0.0)); break;
            case AK_MIN_I64: red = wave_min_u64(in ? i64_to_ordered((int64_t)v) : ~0ull); break;
            case AK_MIN_F64: red = wave_min_u64(in ? f64_to_ordered(__longlong_as_double((long long)v)) : ~0ull); break;
            case AK_MAX_I64: red = wave_max_u64(in ? i64_to_ordered((int64_t)v) : 0ull); break;
            default: red = wave_max_u64(in ? f64_to_ordered(__longlong_as_double((long long)v)) : 0ull);
            }
            if (lane_id() == first && inm) {
              switch (kind) {
              case AK_COUNT:
              case AK_SUM_I64: atomicAdd(cell, (unsigned long long)red); break;
              case AK_SUM_F64: unsafeAtomicAdd((double *)cell, __longlong_as_double((long long)red)); break;
              case AK_MIN_I64:
              case AK_MIN_F64: atomicMin(cell, (unsigned long long)red); break;
              default: atomicMax(cell, (unsigned long long)red);
              }
            }
          }
          act = act && !hot;
        }
      }
      if (act) {
        if (cur.id[u] < tfirst[s]) atomicMin(&tfirst[s], cur.id[u]); // (see lds_agg_dense_kernel)
#pragma unroll
        for (int a = 0; a < PART_MAX_ACC; a++) {
          if (a >= n_acc) break;
          const int code = code_of(a);
          const int src = code >> 3;
          if (FLAGS && !(cur.f[u] & (2 << src))) continue; // NULL input is skipped by every accumulator
          uint64_t v = (NV >= 2 && src) ? cur.v1[u] : cur.v0[u];
          acc_apply(code & 7, tacc + (size_t)a * nslots + s, v);
        }
      }
    }
    cur = nxt;
  }
  __syncthreads();
  if (split != 0xffffffffu) {
    // chunk of a skewed bucket: its groups are merged into the bucket's small global table (a few
    // atomics per distinct key of the chunk), which split_emit_kernel appends to the group list once
    // — the chunks of a bucket never emit the same key twice
    unsigned long long *gk = stb.key + (size_t)split * nslots;
    unsigned int *gf = stb.first + (size_t)split * nslots;
    for (uint32_t s = threadIdx.x; s < nslots; s += PART_WG) {
      unsigned int first = tfirst[s];
      if (first == 0xffffffffu) continue;
      uint32_t g = s; // reserved slots keep their place
      bool placed = true;
      if (s < cap) {
        const unsigned long long key = tkey[s];
        g = slot_hash(key) & mask;
        uint32_t probes = 0;
        while (true) {
          unsigned long long prev = atomicCAS(&gk[g], LDS_EMPTY, key);
          if (prev == LDS_EMPTY || prev == key) break;
          g = (g + 1) & mask;
          if (++probes >= cap) { // more groups in this bucket than a table holds (estimate far too low)
            placed = false;
            break;
          }
        }
      }
      if (!placed) {
        // emitted as a partial group of its own; the caller is told that keys may repeat
        // (ov_count[3]) and merges the batch's groups through the global table
        unsigned long long o = atomicAdd(out_count, 1ull);
        if ((int64_t)o < gcap) {
          gkey[o] = tkey[s];
          gfirst[o] = first;
          if (gvalid) gvalid[o] = 1;
#pragma unroll
          for (int a = 0; a < PART_MAX_ACC; a++) {
            if (a >= n_acc) break;
            gacc[(size_t)a * gcap + o] = tacc[(size_t)a * nslots + s];
          }
        }
        atomicExch(ov_count + 2, 1ull);
        continue;
      }
      atomicMin(&gf[g], first);
#pragma unroll
      for (int a = 0; a < PART_MAX_ACC; a++) {
        if (a >= n_acc) break;
        unsigned long long *cell = stb.acc + ((size_t)a * stb.nsplit + split) * nslots + g;
        const unsigned long long v = tacc[(size_t)a * nslots + s];
        switch (code_of(a) & 7) {
        case AK_COUNT:
        case AK_SUM_I64: atomicAdd(cell, v); break;
        case AK_SUM_F64: unsafeAtomicAdd((double *)cell, __longlong_as_double((long long)v)); break;
        case AK_MIN_I64:
        case AK_MIN_F64: atomicMin(cell, v); break; // cells hold order-preserving images
        default: atomicMax(cell, v);
        }
      }
    }
    return;
  }
  // compact the occupied slots of this bucket into the global group list
  unsigned int mine = 0;
  for (uint32_t s = threadIdx.x; s < nslots; s += PART_WG) mine += tfirst[s] != 0xffffffffu;
  unsigned int my_off = atomicAdd(&s_cnt, mine);
  __syncthreads();
  if (threadIdx.x == 0) s_base = atomicAdd(out_count, (unsigned long long)s_cnt);
  __syncthreads();
  unsigned long long base = s_base + my_off;
  for (uint32_t s = threadIdx.x; s < nslots; s += PART_WG) {
    unsigned int first = tfirst[s];
    if (first == 0xffffffffu) continue;
    if ((int64_t)base < gcap) {
      gkey[base] = s == cap + 1 ? LDS_EMPTY : tkey[s];
      gfirst[base] = first;
      if (gvalid) gvalid[base] = s == cap ? 0 : 1;
#pragma unroll
      for (int a = 0; a < PART_MAX_ACC; a++) {
        if (a >= n_acc) break;
        gacc[(size_t)a * gcap + base] = tacc[(size_t)a * nslots + s];
      }
    }
    base++;
  }
}

// ---------------------------------------------------------- dense (direct-addressed) --
// Bucket pass of a range partition (KeyPack::dense): bucket b holds the 2^rbits consecutive key
// offsets [b << rbits, (b + 1) << rbits), so the LDS table is addressed by the low bits of the
// offset — acc[n_acc][R] u64 | first[R] u32, R = 2^rbits, no key column, no probing, and a slot
// is a group exactly when a row reached it (first != ~0).  Rows are packed words, nothing is
// nullable (the pre-conditions of packing).  JOIN: the build keys are the whole range
// [kmin, kmin + range] (unique and dense, checked by the host), so a probe row has a partner
// exactly when its offset is in range; no build-side partition is needed.
//
// split tables (skewed buckets): same direct addressing, first[nsplit][R] | acc[n_acc][nsplit][R].
// Dense route, split buckets: every chunk of a split bucket STORES its finished LDS table (first[R] | acc[n_acc][R],
// plain coalesced stores) as table `split` of `stb` (here: nsplit = number of CHUNK tables), and this kernel reduces the
// tables chunk_lo[t] .. chunk_lo[t + 1] of split bucket t slot by slot and emits the groups.  The first form merged
// every chunk into one global table per bucket with an atomic per slot and accumulator (12 K global atomics per chunk
// at 24 G/s): splitting was affordable for a few skewed buckets only, and a batch of Zipf keys — ~245 unequal buckets
// on 256 CUs, one work item each — ran at the pace of its slowest bucket (C4: bucket pass 1.58 ms against 0.63 for
// uniform keys).  With stores a chunk costs 80 KB of traffic, so EVERY bucket can be cut into several work items.
// (round 5: 16 slots x 16 CHUNK GROUPS per workgroup.  One thread per slot walking all chunk tables of its bucket was a chain
//  of dependent loads as long as the bucket has chunks — a bucket that is one hot key has ~1500: 1.3 ms for 4096 threads on 16
//  CUs, half of what `c5_variants.adversarial.hot_key` cost over the headline)
constexpr int SED_SLOTS = 16, SED_GROUPS = 16;
__global__ __launch_bounds__(SED_SLOTS * SED_GROUPS) void split_emit_dense_kernel(
    SplitTables stb, const uint32_t *__restrict__ split_bucket, const uint32_t *__restrict__ chunk_lo, uint32_t nbuckets_split,
    uint32_t R, KeyPack kp, LdsAggParams prm, int n_acc, unsigned long long *out_count, uint64_t *__restrict__ gkey,
    uint32_t *__restrict__ gfirst, uint64_t *__restrict__ gacc, int64_t gcap) {
  __shared__ unsigned int s_first[SED_GROUPS][SED_SLOTS];
  __shared__ unsigned long long s_acc[PART_MAX_ACC][SED_GROUPS][SED_SLOTS];
  const uint32_t tiles_per_bucket = (R + SED_SLOTS - 1) / SED_SLOTS;
  const uint32_t t = blockIdx.x / tiles_per_bucket, sl = threadIdx.x % SED_SLOTS, cg = threadIdx.x / SED_SLOTS;
  const uint32_t s = (blockIdx.x % tiles_per_bucket) * SED_SLOTS + sl;
  if (t >= nbuckets_split) return;
  const int64_t total = (int64_t)stb.nsplit * R;
  unsigned int first = 0xffffffffu;
  unsigned long long acc[PART_MAX_ACC];
  for (int a = 0; a < n_acc; a++) acc[a] = acc_identity_cell(prm.code[a] & 7);
  if (s < R) {
    const uint32_t c1 = chunk_lo[t + 1];
    for (uint32_t c = chunk_lo[t] + cg; c < c1; c += SED_GROUPS) {
      const unsigned int f = stb.first[(size_t)c * R + s];
      if (f == 0xffffffffu) continue; // (a slot no row of this chunk touched holds identities)
      first = min(first, f);
      for (int a = 0; a < n_acc; a++) {
        const unsigned long long v = stb.acc[(size_t)a * total + (size_t)c * R + s];
        switch (prm.code[a] & 7) {
        case AK_COUNT:
        case AK_SUM_I64: acc[a] += v; break;
        case AK_SUM_F64: acc[a] = (unsigned long long)__double_as_longlong(__longlong_as_double((long long)acc[a]) + __longlong_as_double((long long)v)); break;
        case AK_MIN_I64:
        case AK_MIN_F64: acc[a] = min(acc[a], v); break;
        default: acc[a] = max(acc[a], v);
        }
      }
    }
  }
  s_first[cg][sl] = first;
  for (int a = 0; a < n_acc; a++) s_acc[a][cg][sl] = acc[a];
  __syncthreads();
  if (cg != 0 || s >= R) return;
  for (int g = 1; g < SED_GROUPS; g++) { // (chunk groups in order: the sums add up in one fixed order per slot)
    const unsigned int f = s_first[g][sl];
    if (f == 0xffffffffu) continue;
    first = min(first, f);
    for (int a = 0; a < n_acc; a++) {
      const unsigned long long v = s_acc[a][g][sl];
      switch (prm.code[a] & 7) {
      case AK_COUNT:
      case AK_SUM_I64: acc[a] += v; break;
      case AK_SUM_F64: acc[a] = (unsigned long long)__double_as_longlong(__longlong_as_double((long long)acc[a]) + __longlong_as_double((long long)v)); break;
      case AK_MIN_I64:
      case AK_MIN_F64: acc[a] = min(acc[a], v); break;
      default: acc[a] = max(acc[a], v);
      }
    }
  }
  if (first == 0xffffffffu) return;
  unsigned int mlt = 1;
  if (prm.partner_mult) mlt = prm.partner_mult[((uint64_t)split_bucket[t] << kp.rbits) + s]; // (> 0: the chunk tables hold partners only)
  unsigned long long base = atomicAdd(out_count, 1ull);
  if ((int64_t)base >= gcap) return;
  gkey[base] = kp.kmin + ((uint64_t)split_bucket[t] << kp.rbits) + s;
  gfirst[base] = first;
  for (int a = 0; a < n_acc; a++) gacc[(size_t)a * gcap + base] = prm.partner_mult ? acc_times(prm.code[a] & 7, acc[a], mlt) : acc[a];
}

template <int NV, bool JOIN, int NACC, int C0, int C1, bool REC = false>
__global__ __launch_bounds__(DENSE_WG) void lds_agg_dense_kernel(
    LdsAggParams prm, const uint64_t *__restrict__ pk, const uint64_t *__restrict__ pv0,
    const uint32_t *__restrict__ work, unsigned long long *out_count, uint64_t *__restrict__ gkey,
    uint32_t *__restrict__ gfirst, uint64_t *__restrict__ gacc, int64_t gcap, KeyPack kp, SplitTables stb) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long tab[];
  __shared__ unsigned int s_cnt;
  __shared__ unsigned long long s_base;
  const uint32_t b = work[4 * blockIdx.x];
  const int64_t lo = work[4 * blockIdx.x + 1];
  const int64_t hi = work[4 * blockIdx.x + 2];
  const uint32_t split = work[4 * blockIdx.x + 3];
  const uint32_t R = prm.cap, mask = R - 1;
  const int n_acc = NACC >= 0 ? NACC : prm.n_acc;
  auto code_of = [&](int a) { return NACC >= 0 ? (a == 0 ? C0 : C1) : prm.code[a]; };
  unsigned long long *tacc = tab;
  unsigned int *tfirst = (unsigned int *)(tacc + (size_t)n_acc * R);
  AggRows<NV> cur, nxt;
  if (lo < hi) {
    if (REC) lds_agg_load_rec<NV, DENSE_WG>(pk, lo + threadIdx.x, hi, cur);
    else lds_agg_load<NV, false, true, DENSE_WG>(pk, nullptr, pv0, nullptr, nullptr, lo + threadIdx.x, hi, cur);
  }
  for (uint32_t s = threadIdx.x; s < R; s += DENSE_WG) {
    tfirst[s] = 0xffffffffu;
#pragma unroll
    for (int a = 0; a < PART_MAX_ACC; a++) {
      if (a >= n_acc) break;
      tacc[(size_t)a * R + s] = acc_identity_cell(code_of(a) & 7);
    }
  }
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  // Hot-slot CARRY (round 5): the wave-level reductions of the hot-key path used to reach the table with three LDS atomics per row
  // slot — and when a bucket is ONE hot key (30 % of the fact rows on a key: 1.5e8 rows in one bucket, cut into ~1500 chunks) all
  // sixteen waves queue on the same three addresses: the atomics of one address are serial in the LDS unit, ~90 cycles per 64 rows.
  // The wave keeps (slot, first row, accumulators) of its current hot slot in registers (wave-uniform values) across row slots and
  // adds them to the table when the slot changes and once at the end.
  uint32_t hc_slot = 0xffffffffu, hc_min = 0xffffffffu;
  uint64_t hc_acc[PART_MAX_ACC];
  auto hot_flush = [&]() {
    if (hc_slot == 0xffffffffu) return;
    if (lane_id() == 0) {
      atomicMin(&tfirst[hc_slot], hc_min);
#pragma unroll
      for (int a = 0; a < PART_MAX_ACC; a++) {
        if (a >= n_acc) break;
        unsigned long long *cell = tacc + (size_t)a * R + hc_slot;
        switch (code_of(a) & 7) {
        case AK_COUNT:
        case AK_SUM_I64: atomicAdd(cell, (unsigned long long)hc_acc[a]); break;
        case AK_SUM_F64: unsafeAtomicAdd((double *)cell, __longlong_as_double((long long)hc_acc[a])); break;
        case AK_MIN_I64:
        case AK_MIN_F64: atomicMin(cell, (unsigned long long)hc_acc[a]); break;
        default: atomicMax(cell, (unsigned long long)hc_acc[a]);
        }
      }
    }
    hc_slot = 0xffffffffu;
  };
  auto hot_take = [&](uint32_t s0, uint32_t idmin) { // the carry now belongs to slot s0
    if (hc_slot != s0) {
      hot_flush();
      hc_slot = s0;
      hc_min = 0xffffffffu;
#pragma unroll
      for (int a = 0; a < PART_MAX_ACC; a++) {
        if (a >= n_acc) break;
        hc_acc[a] = acc_identity_cell(code_of(a) & 7);
      }
    }
    hc_min = idmin < hc_min ? idmin : hc_min;
  };
  auto hot_add = [&](int a, int kind, uint64_t red) {
    switch (kind) {
    case AK_COUNT:
    case AK_SUM_I64: hc_acc[a] += red; break;
    case AK_SUM_F64: hc_acc[a] = (uint64_t)__double_as_longlong(__longlong_as_double((long long)hc_acc[a]) + __longlong_as_double((long long)red)); break;
    case AK_MIN_I64:
    case AK_MIN_F64: hc_acc[a] = red < hc_acc[a] ? red : hc_acc[a]; break;
    default: hc_acc[a] = red > hc_acc[a] ? red : hc_acc[a];
    }
  };
  __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): same entry state as the loop's back edge
  for (int64_t base = lo; base < hi; base += (int64_t)LDS_U * DENSE_WG) {
    const int64_t i0 = base + threadIdx.x;
    if (REC) lds_agg_load_rec<NV, DENSE_WG>(pk, i0 + (int64_t)LDS_U * DENSE_WG, hi, nxt);
    else lds_agg_load<NV, false, true, DENSE_WG>(pk, nullptr, pv0, nullptr, nullptr, i0 + (int64_t)LDS_U * DENSE_WG, hi, nxt);
#pragma unroll
    for (int u = 0; u < LDS_U; u++) {
      const uint64_t off = packed_off(kp, cur.k[u]);
      const uint32_t id = packed_row(kp, cur.k[u]);
      const uint32_t s = (uint32_t)off & mask;
      bool act = i0 + (int64_t)u * DENSE_WG < hi;
      // outside the range of the keys of interest: a probe row without partner (fused join), or a sentinel row of a
      // claimed partition (radix_part.hip; every real row of a plain aggregation lies inside the range)
      act = act && off <= kp.range;
      // hot keys: see lds_agg_kernel.  (Repeating the test for the next active lane's slot — up to four rounds per
      // row slot, for buckets with several hot keys — was measured SLOWER on Zipf(1.1) keys, 1.5 -> 1.8 ms for the C4
      // batch: a wave reduction costs more than the ~8-20 serialised LDS atomics it replaces.)
      const uint64_t actm = __ballot(act);
      if (actm) {
        const int first = __builtin_ctzll(actm);
        const uint32_t s0 = (uint32_t)__shfl((int)s, first, 64);
        const bool hot = act && s == s0;
        const uint64_t peers = __ballot(hot);
        if (__popcll(peers) >= HOT_MIN_PEERS) {
          const uint32_t idmin = wave_min_u32_dpp(hot ? id : 0xffffffffu);
          hot_take(s0, idmin);
#pragma unroll
          for (int a = 0; a < PART_MAX_ACC; a++) {
            if (a >= n_acc) break;
            const int kind = code_of(a) & 7;
            const uint64_t v = NV >= 1 ? cur.v0[u] : 0ull;
            uint64_t red;
            switch (kind) {
            case AK_COUNT: red = (uint64_t)__popcll(peers); break;
            case AK_SUM_I64: red = wave_sum_u64_dpp(hot ? v : 0ull); break;
            case AK_SUM_F64: red = (uint64_t)__double_as_longlong(wave_sum_f64_dpp(hot ? __longlong_as_double((long long)v) : 0.0)); break;
            case AK_MIN_I64: red = wave_min_u64(hot ? i64_to_ordered((int64_t)v) : ~0ull); break;
            case AK_MIN_F64: red = wave_min_u64(hot ? f64_to_ordered(__longlong_as_double((long long)v)) : ~0ull); break;
            case AK_MAX_I64: red = wave_max_u64(hot ? i64_to_ordered((int64_t)v) : 0ull); break;
            default: red = wave_max_u64(hot ? f64_to_ordered(__longlong_as_double((long long)v)) : 0ull);
            }
            hot_add(a, kind, red);
          }
          act = act && !hot;
        }
      }
      if (act) {
        // most rows are not their group's first row: a plain LDS read decides, the atomic (an order
        // of magnitude more expensive than a read on this LDS) runs for the few that lower the minimum
        if (id < tfirst[s]) atomicMin(&tfirst[s], id);
#pragma unroll
        for (int a = 0; a < PART_MAX_ACC; a++) {
          if (a >= n_acc) break;
          acc_apply(code_of(a) & 7, tacc + (size_t)a * R + s, NV >= 1 ? cur.v0[u] : 0ull);
        }
      }
    }
    cur = nxt;
  }
  hot_flush();
  __syncthreads();
  if (JOIN && (prm.partner_bits || prm.partner_mult)) { // build keys with gaps: a slot whose key has no build partner is not a group
    const uint64_t off0 = (uint64_t)b << kp.rbits; // (a multiple of 64 or R < 64: rbits >= 8)
    for (uint32_t s = threadIdx.x; s < R; s += DENSE_WG) { // (the slots this thread stores / counts below)
      if (tfirst[s] == 0xffffffffu) continue;
      const uint64_t o = off0 + s;
      const bool partner = prm.partner_mult ? prm.partner_mult[o] != 0 : (((prm.partner_bits[o >> 6] >> (o & 63)) & 1ull) != 0);
      if (!partner) tfirst[s] = 0xffffffffu;
    }
  }
  if (split != 0xffffffffu) { // chunk of a split bucket: its table goes out as chunk table `split` (split_emit_dense_kernel reduces)
    unsigned int *gf = stb.first + (size_t)split * R;
    for (uint32_t s = threadIdx.x; s < R; s += DENSE_WG) {
      gf[s] = tfirst[s];
#pragma unroll
      for (int a = 0; a < PART_MAX_ACC; a++) {
        if (a >= n_acc) break;
        stb.acc[((size_t)a * stb.nsplit + split) * R + s] = tacc[(size_t)a * R + s];
      }
    }
    return;
  }
  unsigned int mine = 0;
  for (uint32_t s = threadIdx.x; s < R; s += DENSE_WG) mine += tfirst[s] != 0xffffffffu;
  unsigned int my_off = atomicAdd(&s_cnt, mine);
  __syncthreads();
  if (threadIdx.x == 0) s_base = atomicAdd(out_count, (unsigned long long)s_cnt);
  __syncthreads();
  unsigned long long base = s_base + my_off;
  const uint64_t key0 = kp.kmin + ((uint64_t)b << kp.rbits);
  for (uint32_t s = threadIdx.x; s < R; s += DENSE_WG) {
    unsigned int first = tfirst[s];
    if (first == 0xffffffffu) continue;
    if ((int64_t)base < gcap) {
      gkey[base] = key0 + s;
      gfirst[base] = first;
      const unsigned int mlt = (JOIN && prm.partner_mult) ? prm.partner_mult[((uint64_t)b << kp.rbits) + s] : 1u;
#pragma unroll
      for (int a = 0; a < PART_MAX_ACC; a++) {
        if (a >= n_acc) break;
        const unsigned long long cell = tacc[(size_t)a * R + s];
        gacc[(size_t)a * gcap + base] = (JOIN && prm.partner_mult) ? acc_times(code_of(a) & 7, cell, mlt) : cell;
      }
    }
    base++;
  }
}

// Bucket pass over SLIM rows (radix_part.hpp PartitionedRows::Slim; radix_part.hip "slim records"): value + 32-bit word
// {slot | row inside its level-1 tile << rbits | tile delta << (rbits + 13)} per row, 12 instead of 16 bytes.  The word
// no longer holds the row id — only the first-seen order of the groups needs it (hash_agg.rs:87-99) — so it is rebuilt:
// row = (base tile of the run the row sits in + delta) * tile + row inside the tile.  The bucket is the concatenation
// of its non-empty RUNS (one per level-2 input tile that sent it rows) whose starts / base tiles rp_slim_runs_kernel
// listed; the prologue turns the starts inside this work item's range into one bit per position + a prefix count per
// 64-position group (LDS), and a row at position p finds its run with two wave-uniform LDS reads and a popcount.  The
// base tile is then ONE 4-byte load per row from a list a wave reads 1-3 consecutive entries of; it depends on the
// position only, so it is issued together with the prefetch of the row itself.
struct SlimBucketIn {
  SlimRowsView rows;
  const uint32_t *nzstart, *nzbt, *nzcount, *bcol;
  uint32_t tile;   // rows per level-1 tile
  uint32_t groups; // 64-position groups the LDS arrays hold (>= those of the largest work item)
  // BLK (rows of a claimed single level, rp_claim_scatter_slim_kernel): base tile of slot i = blk_bt[i >> log_b], the
  // word 0xffffffff marks a sentinel row; none of the run arrays above is used
  const uint32_t *blk_bt;
  uint32_t log_b;
};
constexpr uint32_t SLIM_BT_CAP = 4096; // base tiles of a work item's runs kept in LDS (16 KiB)
struct SlimAggRows {
  uint64_t v[LDS_U];
  uint32_t w[LDS_U], bt[LDS_U];
};

template <bool JOIN, int NACC, int C0, int C1, bool BLK = false>
__global__ __launch_bounds__(DENSE_WG) void lds_agg_dense_slim_kernel(
    LdsAggParams prm, SlimBucketIn in, const uint32_t *__restrict__ work, unsigned long long *out_count,
    uint64_t *__restrict__ gkey, uint32_t *__restrict__ gfirst, uint64_t *__restrict__ gacc, int64_t gcap, KeyPack kp,
    SplitTables stb) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long tab[];
  __shared__ unsigned int s_cnt, s_before, s_inside;
  __shared__ unsigned long long s_base;
  __shared__ uint32_t s_wsum[DENSE_WG / 64];
  const uint32_t b = work[4 * blockIdx.x];
  const int64_t lo = work[4 * blockIdx.x + 1];
  const int64_t hi = work[4 * blockIdx.x + 2];
  const uint32_t split = work[4 * blockIdx.x + 3];
  const uint32_t R = prm.cap, mask = R - 1;
  const int n_acc = NACC >= 0 ? NACC : prm.n_acc;
  auto code_of = [&](int a) { return NACC >= 0 ? (a == 0 ? C0 : C1) : prm.code[a]; };
  unsigned long long *tacc = tab;
  unsigned int *tfirst = (unsigned int *)(tacc + (size_t)n_acc * R);
  unsigned long long *gmask = (unsigned long long *)(tfirst + R); // [groups] bit = a run starts at this position (> lo)
  uint32_t *gpre = (uint32_t *)(gmask + in.groups);              // [groups] index of the run holding the group's first row
  uint32_t *gbt = gpre + in.groups;                              // [SLIM_BT_CAP] base tiles of the item's runs (when they fit)
  const uint32_t col = BLK ? 0u : in.bcol[b], nruns = BLK ? 0u : in.nzcount[b];
  const uint32_t ngroups = BLK ? 0u : (uint32_t)((hi - lo + 63) >> 6);
  const uint64_t le_mask = (2ull << lane_id()) - 1ull;
  const uint32_t rbits = kp.rbits, lmask = (1u << 13) - 1u;

  for (uint32_t s = threadIdx.x; s < R; s += DENSE_WG) {
    tfirst[s] = 0xffffffffu;
#pragma unroll
    for (int a = 0; a < PART_MAX_ACC; a++) {
      if (a >= n_acc) break;
      tacc[(size_t)a * R + s] = acc_identity_cell(code_of(a) & 7);
    }
  }
  for (uint32_t g = threadIdx.x; g < ngroups; g += DENSE_WG) gmask[g] = 0;
  if (threadIdx.x == 0) {
    s_cnt = 0;
    s_before = 0;
    s_inside = 0;
  }
  __syncthreads();
  if (!BLK) { // runs of the bucket: how many start at or before lo, one bit for every start inside (lo, hi)
    uint32_t before = 0, inside = 0;
    for (uint32_t k = threadIdx.x; k < nruns; k += DENSE_WG) {
      const int64_t st = in.nzstart[col + k];
      if (st <= lo) before++;
      else if (st < hi) {
        inside++;
        atomicOr(&gmask[(st - lo) >> 6], 1ull << ((st - lo) & 63));
      }
    }
    before = wave_sum_u32(before);
    inside = wave_sum_u32(inside);
    if (lane_id() == 0 && before) atomicAdd(&s_before, before);
    if (lane_id() == 0 && inside) atomicAdd(&s_inside, inside);
  }
  __syncthreads();
  // the base tiles of the runs this item touches (run s_before - 1 holds row lo, then the `inside` runs that start inside
  // the item) go to LDS when they fit: one LDS read per row instead of a global load (1.59 -> ~1.48 ms per C5 step)
  const uint32_t k_first = s_before - 1u, k_count = s_inside + 1u;
  const bool bt_lds = !BLK && k_count <= SLIM_BT_CAP;
  // Rows that arrived ORDERED by key reach the bucket as a few dozen long runs (random rows: one run per level-2 tile of
  // the segment, ~1500 short ones), and consecutive rows — the 64 lanes of a wave — carry two or three keys: 64 atomics on two
  // LDS cells.  Such an item adds a wave's rows per RUN OF EQUAL SLOTS instead: one inclusive scan of the values over the
  // wave, the run's last lane adds the difference of two prefix sums and the run's length (COUNT + SUM(double) only;
  // bench.py c5_variants.adversarial.sorted_fact: bucket pass 4.1 -> see DESIGN.md).  Workgroup-uniform.
  const bool seg_mode = NACC == 2 && ((C0 == AK_COUNT && C1 == AK_SUM_F64) || (C0 == AK_SUM_F64 && C1 == AK_COUNT)) &&
                        hi - lo >= 4096 && (int64_t)k_count * 2048 <= hi - lo && !prm.seg_off && !BLK; // (runs of >= 2048 rows on average; random rows: ~130)
  if (bt_lds)
    for (uint32_t k = threadIdx.x; k < k_count; k += DENSE_WG) gbt[k] = in.nzbt[col + min(k_first + k, nruns - 1)];
  if (!BLK) { // gpre[g] = (runs starting at or before lo) - 1 + bits of the groups before g
    constexpr uint32_t GPT = 8; // groups per thread per round
    uint32_t carry = s_before - 1u; // (the bucket's first run starts at its first row <= lo: s_before >= 1)
    for (uint32_t g0 = 0; g0 < ngroups; g0 += DENSE_WG * GPT) {
      const uint32_t gb = g0 + threadIdx.x * GPT;
      uint32_t pc[GPT], sum = 0;
#pragma unroll
      for (uint32_t i = 0; i < GPT; i++) {
        pc[i] = gb + i < ngroups ? (uint32_t)__popcll(gmask[gb + i]) : 0u;
        sum += pc[i];
      }
      const uint32_t inc = wave_iscan_u32(sum);
      if (lane_id() == 63) s_wsum[wave_id()] = inc;
      __syncthreads();
      uint32_t wbase = 0, tot = 0;
      for (int w = 0; w < DENSE_WG / 64; w++) {
        if (w < wave_id()) wbase += s_wsum[w];
        tot += s_wsum[w];
      }
      uint32_t run = carry + wbase + inc - sum;
#pragma unroll
      for (uint32_t i = 0; i < GPT; i++) {
        if (gb + i < ngroups) gpre[gb + i] = run;
        run += pc[i];
      }
      carry += tot;
      __syncthreads();
    }
  }
  // A wave's rows of one trip are CONSECUTIVE (LDS_U groups of 64, wave w from row w * LDS_U * 64 of the trip), not DENSE_WG
  // apart: rows that arrive ordered by key keep the wave on one key for several row slots, and the hot-slot carry below
  // reaches the table once per key instead of once per row slot (SLIM_WAVE_SPAN=0: the old mapping).  Every load is still
  // 64 consecutive slots per wave.
#ifndef SLIM_WAVE_SPAN
#define SLIM_WAVE_SPAN 1
#endif
  constexpr int64_t SLIM_USTEP = SLIM_WAVE_SPAN ? 64 : DENSE_WG;
  const int64_t slim_toff = SLIM_WAVE_SPAN ? (int64_t)wave_id() * (LDS_U * 64) + lane_id() : (int64_t)threadIdx.x;
  // rows i0 + u * SLIM_USTEP of this thread: value, word, and the base tile of their run
  auto load = [&](int64_t i0, SlimAggRows &r) {
#pragma unroll
    for (int u = 0; u < LDS_U; u++) {
      const int64_t i = min(i0 + (int64_t)u * SLIM_USTEP, hi - 1);
      slim_load_nt(in.rows, i, r.w[u], r.v[u]);
      if (BLK) { // (a wave's 64 consecutive slots lie in one or two blocks: one or two cache lines per load)
        r.bt[u] = in.blk_bt[i >> in.log_b];
        continue;
      }
      const uint32_t g = (uint32_t)((i - lo) >> 6);
      const uint32_t k = gpre[g] + (uint32_t)__popcll(gmask[g] & le_mask);
#if defined(SLIM_DBG) && (SLIM_DBG & 16)
      r.bt[u] = k; // timing experiment: no base-tile load (row ids wrong)
#else
      r.bt[u] = bt_lds ? gbt[min(k - k_first, k_count - 1)] : in.nzbt[col + min(k, nruns - 1)];
#endif
    }
  };
  // Hot-slot CARRY (round 5): the wave-level reductions of the hot-key path used to reach the table with three LDS atomics per row
  // slot — and when a bucket is ONE hot key (30 % of the fact rows on a key: 1.5e8 rows in one bucket, cut into ~1500 chunks) all
  // sixteen waves queue on the same three addresses: the atomics of one address are serial in the LDS unit, ~90 cycles per 64 rows.
  // The wave keeps (slot, first row, accumulators) of its current hot slot in registers (wave-uniform values) across row slots and
  // adds them to the table when the slot changes and once at the end.
  uint32_t hc_slot = 0xffffffffu, hc_min = 0xffffffffu;
  uint64_t hc_acc[PART_MAX_ACC];
  auto hot_flush = [&]() {
    if (hc_slot == 0xffffffffu) return;
    if (lane_id() == 0) {
      atomicMin(&tfirst[hc_slot], hc_min);
#pragma unroll
      for (int a = 0; a < PART_MAX_ACC; a++) {
        if (a >= n_acc) break;
        unsigned long long *cell = tacc + (size_t)a * R + hc_slot;
        switch (code_of(a) & 7) {
        case AK_COUNT:
        case AK_SUM_I64: atomicAdd(cell, (unsigned long long)hc_acc[a]); break;
        case AK_SUM_F64: unsafeAtomicAdd((double *)cell, __longlong_as_double((long long)hc_acc[a])); break;
        case AK_MIN_I64:
        case AK_MIN_F64: atomicMin(cell, (unsigned long long)hc_acc[a]); break;
        default: atomicMax(cell, (unsigned long long)hc_acc[a]);
        }
      }
    }
    hc_slot = 0xffffffffu;
  };
  auto hot_take = [&](uint32_t s0, uint32_t idmin) { // the carry now belongs to slot s0
    if (hc_slot != s0) {
      hot_flush();
      hc_slot = s0;
      hc_min = 0xffffffffu;
#pragma unroll
      for (int a = 0; a < PART_MAX_ACC; a++) {
        if (a >= n_acc) break;
        hc_acc[a] = acc_identity_cell(code_of(a) & 7);
      }
    }
    hc_min = idmin < hc_min ? idmin : hc_min;
  };
  auto hot_add = [&](int a, int kind, uint64_t red) {
    switch (kind) {
    case AK_COUNT:
    case AK_SUM_I64: hc_acc[a] += red; break;
    case AK_SUM_F64: hc_acc[a] = (uint64_t)__double_as_longlong(__longlong_as_double((long long)hc_acc[a]) + __longlong_as_double((long long)red)); break;
    case AK_MIN_I64:
    case AK_MIN_F64: hc_acc[a] = red < hc_acc[a] ? red : hc_acc[a]; break;
    default: hc_acc[a] = red > hc_acc[a] ? red : hc_acc[a];
    }
  };
  SlimAggRows cur, nxt;
  if (lo < hi) load(lo + slim_toff, cur);
  __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): same entry state as the loop's back edge
  for (int64_t base = lo; base < hi; base += (int64_t)LDS_U * DENSE_WG) {
    const int64_t i0 = base + slim_toff;
    load(i0 + (int64_t)LDS_U * DENSE_WG, nxt);
    // LDS operations complete in order: a read behind an atomic waits for it.  So the trip's reads (the slots' current
    // first rows) are all issued BEFORE its atomics — a stale (larger) first row only costs a redundant atomicMin —
    // and the hot-key test takes the first active lane's slot with v_readlane, not with a ds_bpermute shuffle.
    uint32_t sl[LDS_U], id[LDS_U], tf[LDS_U];
#pragma unroll
    for (int u = 0; u < LDS_U; u++) {
      const uint32_t w = cur.w[u];
      sl[u] = w & mask;
      id[u] = (cur.bt[u] + (w >> (rbits + 13))) * in.tile + ((w >> rbits) & lmask);
      tf[u] = tfirst[sl[u]];
    }
    // a wave's rows added per RUN OF EQUAL SLOTS (COUNT + SUM(double) only): the run's last lane adds the run's length, its
    // sum and its smallest row
    constexpr bool SEG_CAPABLE = NACC == 2 && ((C0 == AK_COUNT && C1 == AK_SUM_F64) || (C0 == AK_SUM_F64 && C1 == AK_COUNT));
    const int lane = (int)lane_id();
    auto seg_rows = [&](bool act, uint32_t sl_u, uint64_t v_u, uint32_t id_u, uint32_t tf_u) {
      const uint32_t key = act ? sl_u : (0x80000000u | (uint32_t)lane); // (rows past the end: runs of their own, never added)
      const uint32_t prevk = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x138, 0xf, 0xf, false); // wave_shr:1 (lane 0 is a run start anyway)
      const uint64_t bm = __ballot(lane == 0 || key != prevk);            // bit = a run starts at this lane
      const int headl = 63 - __builtin_clzll(bm & le_mask);               // first lane of this lane's run
      // SEGMENTED inclusive scan: only values of the same run are added (a difference of wave-wide prefix sums would let a
      // neighbouring key's magnitudes into this key's rounding)
      // ... and a segmented MIN of the row ids beside it: both slim partition levels rank a tile's rows of one digit in
      // LDS-atomic order, so two producer waves can interleave and a run's head lane need not hold its smallest row
      // (a later key whose first row falls into that gap would otherwise come out ahead in the first-seen order,
      // hash_agg.rs:98).
      double incl = act ? __longlong_as_double((long long)v_u) : 0.0;
      uint32_t idh = act ? id_u : 0xffffffffu;
      wave_seg_iscan_f64_min_u32(incl, idh, headl, lane); // (DPP; the __shfl_up form: SQLRS history, round 4)
      const bool tail = act && (lane == 63 || ((bm >> (lane + 1)) & 1ull));
      if (tail) {
        const uint32_t s = sl_u;
        if (idh < tf_u) atomicMin(&tfirst[s], idh);
        atomicAdd(tacc + (size_t)(C0 == AK_COUNT ? 0 : 1) * R + s, (unsigned long long)(lane - headl + 1));
        unsafeAtomicAdd((double *)(tacc + (size_t)(C0 == AK_COUNT ? 1 : 0) * R + s), incl);
      }
    };
    if (seg_mode) {
#pragma unroll
      for (int u = 0; u < LDS_U; u++) {
        const bool act = i0 + (int64_t)u * SLIM_USTEP < hi;
        { // the whole wave is ONE run (a hot key's bucket; long runs of ordered rows): a plain wave reduction into the hot-slot carry
          const uint32_t s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sl[u]);
          if (__ballot(act && sl[u] == s0) == ~0ull) {
            hot_take(s0, wave_min_u32_dpp(id[u]));
            hot_add(C0 == AK_COUNT ? 0 : 1, AK_COUNT, 64ull);
            hot_add(C0 == AK_COUNT ? 1 : 0, AK_SUM_F64, (uint64_t)__double_as_longlong(wave_sum_f64_dpp(__longlong_as_double((long long)cur.v[u]))));
            continue;
          }
        }
        seg_rows(act, sl[u], cur.v[u], id[u], tf[u]);
      }
      cur = nxt;
      continue;
    }
#pragma unroll
    for (int u = 0; u < LDS_U; u++) {
      const uint32_t s = sl[u];
      bool act = i0 + (int64_t)u * SLIM_USTEP < hi;
      if (BLK) act = act && cur.w[u] != 0xffffffffu; // sentinel rows of the claimed level
#ifdef SLIM_DBG
      uint64_t actm = (SLIM_DBG & 1) ? 0ull : __ballot(act); // timing experiments only (results invalid)
#else
      uint64_t actm = __ballot(act);
#endif
      // hot keys (see lds_agg_kernel): the rows on the slot of the first active lane, when there are enough of them, are
      // reduced over the wave into the carry — and then the same again for the lanes that are left, up to three slots
      // (rows ordered by key: a wave holds the tail of one key's run and the head of the next; the second key's lanes used
      // to queue on one LDS address, C4's bucket pass 1.39 ms on sorted rows against 0.45).  Random keys fail the first
      // test and leave.
      for (int it = 0; it < 3 && actm; it++) {
        const int first = __builtin_ctzll(actm);
        const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)s, first);
        const bool hot = act && s == s0;
        const uint64_t peers = __ballot(hot);
        if (__popcll(peers) < HOT_MIN_PEERS) break;
        // a hot slot on CONSECUTIVE lanes, and other rows behind it: runs of equal keys (rows ordered by key; a hot key among
        // random ones sits on scattered lanes and stays with the reduction below) — the whole wave per run
        if (SEG_CAPABLE && it == 0 && peers != actm && (((peers >> first) + 1) & (peers >> first)) == 0 && !prm.seg_off) {
          seg_rows(act, s, cur.v[u], id[u], tf[u]);
          act = false;
          break;
        }
        const uint32_t idmin = wave_min_u32_dpp(hot ? id[u] : 0xffffffffu);
        hot_take(s0, idmin);
#pragma unroll
        for (int a = 0; a < PART_MAX_ACC; a++) {
          if (a >= n_acc) break;
          const int kind = code_of(a) & 7;
          const uint64_t v = cur.v[u];
          uint64_t red;
          switch (kind) {
          case AK_COUNT: red = (uint64_t)__popcll(peers); break;
          case AK_SUM_I64: red = wave_sum_u64_dpp(hot ? v : 0ull); break;
          case AK_SUM_F64: red = (uint64_t)__double_as_longlong(wave_sum_f64_dpp(hot ? __longlong_as_double((long long)v) : 0.0)); break;
          case AK_MIN_I64: red = wave_min_u64(hot ? i64_to_ordered((int64_t)v) : ~0ull); break;
          case AK_MIN_F64: red = wave_min_u64(hot ? f64_to_ordered(__longlong_as_double((long long)v)) : ~0ull); break;
          case AK_MAX_I64: red = wave_max_u64(hot ? i64_to_ordered((int64_t)v) : 0ull); break;
          default: red = wave_max_u64(hot ? f64_to_ordered(__longlong_as_double((long long)v)) : 0ull);
          }
          hot_add(a, kind, red);
        }
        act = act && !hot;
        actm &= ~peers;
      }
      if (act) {
#ifdef SLIM_DBG
        if (!(SLIM_DBG & 8) && id[u] < tf[u]) atomicMin(&tfirst[s], id[u]);
#pragma unroll
        for (int a = 0; a < PART_MAX_ACC; a++) {
          if (a >= n_acc) break;
          if ((SLIM_DBG >> (1 + a)) & 1) continue;
          acc_apply(code_of(a) & 7, tacc + (size_t)a * R + s, cur.v[u]);
        }
#else
        if (id[u] < tf[u]) atomicMin(&tfirst[s], id[u]);
#pragma unroll
        for (int a = 0; a < PART_MAX_ACC; a++) {
          if (a >= n_acc) break;
          acc_apply(code_of(a) & 7, tacc + (size_t)a * R + s, cur.v[u]);
        }
#endif
      }
    }
    cur = nxt;
  }
  hot_flush();
  __syncthreads();
  if (JOIN && (prm.partner_bits || prm.partner_mult)) { // build keys with gaps: a slot whose key has no build partner is not a group
    const uint64_t off0 = (uint64_t)b << kp.rbits;
    for (uint32_t s = threadIdx.x; s < R; s += DENSE_WG) {
      if (tfirst[s] == 0xffffffffu) continue;
      const uint64_t o = off0 + s;
      const bool partner = prm.partner_mult ? prm.partner_mult[o] != 0 : (((prm.partner_bits[o >> 6] >> (o & 63)) & 1ull) != 0);
      if (!partner) tfirst[s] = 0xffffffffu;
    }
  }
  if (split != 0xffffffffu) { // chunk of a split bucket: its table goes out as chunk table `split` (split_emit_dense_kernel reduces)
    unsigned int *gf = stb.first + (size_t)split * R;
    for (uint32_t s = threadIdx.x; s < R; s += DENSE_WG) {
      gf[s] = tfirst[s];
#pragma unroll
      for (int a = 0; a < PART_MAX_ACC; a++) {
        if (a >= n_acc) break;
        stb.acc[((size_t)a * stb.nsplit + split) * R + s] = tacc[(size_t)a * R + s];
      }
    }
    return;
  }
  unsigned int mine = 0;
  for (uint32_t s = threadIdx.x; s < R; s += DENSE_WG) mine += tfirst[s] != 0xffffffffu;
  unsigned int my_off = atomicAdd(&s_cnt, mine);
  __syncthreads();
  if (threadIdx.x == 0) s_base = atomicAdd(out_count, (unsigned long long)s_cnt);
  __syncthreads();
  unsigned long long obase = s_base + my_off;
  const uint64_t key0 = kp.kmin + ((uint64_t)b << kp.rbits);
  for (uint32_t s = threadIdx.x; s < R; s += DENSE_WG) {
    unsigned int first = tfirst[s];
    if (first == 0xffffffffu) continue;
    if ((int64_t)obase < gcap) {
      gkey[obase] = key0 + s;
      gfirst[obase] = first;
      const unsigned int mlt = (JOIN && prm.partner_mult) ? prm.partner_mult[((uint64_t)b << kp.rbits) + s] : 1u;
#pragma unroll
      for (int a = 0; a < PART_MAX_ACC; a++) {
        if (a >= n_acc) break;
        const unsigned long long cell = tacc[(size_t)a * R + s];
        gacc[(size_t)a * gcap + obase] = (JOIN && prm.partner_mult) ? acc_times(code_of(a) & 7, cell, mlt) : cell;
      }
    }
    obase++;
  }
}

__global__ void first_to_rowid_kernel(const uint32_t *__restrict__ gfirst, int64_t n, uint64_t offset,
                                      uint64_t *__restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = offset + gfirst[i];
}
__global__ void bytes_pack_kernel(const uint8_t *__restrict__ bytes, int64_t n, uint64_t *__restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  bool b = (i < n) && bytes[i];
  uint64_t m = __ballot(b);
  if (lane_id() == 0 && i < n) out[i >> 6] = m;
}

// min / max of the signed-order image of `keys` (all valid)
__global__ __launch_bounds__(256) void key_range_kernel(const uint64_t *__restrict__ keys, int64_t n, unsigned long long *mm) {
  uint64_t kmin = ~0ull, kmax = 0;
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t o = keys[r] ^ (1ull << 63);
    kmin = min(kmin, o);
    kmax = max(kmax, o);
  }
  kmin = wave_min_u64(kmin);
  kmax = wave_max_u64(kmax);
  __shared__ unsigned long long s_lo[4], s_hi[4]; // 256 threads; one pair of atomics per block
  if (lane_id() == 0) {
    s_lo[wave_id()] = kmin;
    s_hi[wave_id()] = kmax;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) {
      kmin = min(kmin, (uint64_t)s_lo[w]);
      kmax = max(kmax, (uint64_t)s_hi[w]);
    }
    atomicMin(mm, (unsigned long long)kmin);
    atomicMax(mm + 1, (unsigned long long)kmax);
  }
}

const uint64_t *part_row_ids(Ctx *ctx, PartAggOutput &po) {
  if (!po.row_ids) {
    const int64_t g1 = std::max<int64_t>(po.groups, 1);
    po.row_ids = ctx->alloc(8 * (size_t)g1);
    if (po.groups) {
      first_to_rowid_kernel<<<dim3((unsigned)ceil_div(g1, 256)), dim3(256), 0, ctx->stream>>>(
          po.gfirst->as<uint32_t>(), po.groups, po.row_offset, po.row_ids->as<uint64_t>());
      SQ_HIP(hipGetLastError());
    }
  }
  return po.row_ids->as<uint64_t>();
}

bool partitioned_preaggregate(Ctx *ctx, const PartAggSpec &spec, const PartAggInput &in,
                              uint64_t row_offset, PartAggOutput *out) {
  int64_t n = in.n; // rows of the batch; after the partition: rows that passed in.filter
  if (n > 0xffffffffll || spec.n_acc > PART_MAX_ACC || spec.nv > 2) return false;
  // 1. how many groups?  -> bucket count.  Fused join: every build key needs a slot.
  const bool join_mode = in.join_keys != nullptr;
  uint64_t omin = ~0ull, omax = 0; // signed-order image of the smallest / largest key of interest
  bool sampled = false; // omin / omax are a sample's (widened below): rows are packed optimistically
  const char *kse = hook("SQLRS_KEY_STATS_EXACT"); // measurement hook, read per call: 1 = key statistics from a full pass
  const bool stats_exact_env = kse && kse[0] == '1';
  double est = join_mode ? (double)in.join_n
                         : estimate_distinct(ctx, in.keys, in.key_validity, n, &omin, &omax, (in.exact_stats || stats_exact_env) ? nullptr : &sampled);
  BufP ctr = ctx->alloc_zero(48); // {groups, overflow rows, join failure, split-table full, key outside the sampled range}
  if (sampled && omin <= omax) {
    // widen the sampled range: an eighth of its width (and at least 64 Ki values) on either side
    const uint64_t w = omax - omin, pad = std::max<uint64_t>(w / 8, 65536);
    omin = omin >= pad ? omin - pad : 0;
    omax = omax <= ~0ull - pad ? omax + pad : ~0ull;
  }
  static const double est_scale = [] { // test hook: mis-scale the estimate to force the overflow path
    const char *e = hook("SQLRS_EST_SCALE");
    return e ? std::atof(e) : 1.0;
  }();
  if (!join_mode) est = std::max(1.0, est * est_scale);
  // (key, row) packing: the keys of interest are the batch's own keys, or the build keys of the
  // fused join (a probe key outside their range cannot have a partner)
  KeyPack kp;
  const bool nullable = in.key_validity || in.val_validity[0] || in.val_validity[1];
  if (!nullable && spec.nv <= 1 && !(join_mode && in.join_validity)) {
    if (join_mode && in.join_range_known) {
      omin = in.join_omin;
      omax = in.join_omax;
    } else if (join_mode) {
      BufP mm = ctx->alloc(16); // {min = ~0, max = 0}
      SQ_HIP(hipMemsetAsync(mm->p, 0xff, 8, ctx->stream));
      SQ_HIP(hipMemsetAsync(mm->as<uint8_t>() + 8, 0, 8, ctx->stream));
      unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(in.join_n, 1024), 1024);
      key_range_kernel<<<dim3(std::max(blocks, 1u)), dim3(256), 0, ctx->stream>>>(
          in.join_keys, in.join_n, mm->as<unsigned long long>());
      SQ_HIP(hipGetLastError());
      const uint64_t *h = (const uint64_t *)ctx->fetch(mm->p, 16);
      omin = h[0];
      omax = h[1];
    }
    if (omin <= omax) {
      uint64_t range = omax - omin; // offsets 0..range; sentinel offset = kmask > range
      uint32_t kbits = 1;
      while (kbits < 64 && ((1ull << kbits) - 1) <= range) kbits++;
      int rowbits = 1;
      while (rowbits < 32 && (1ll << rowbits) < n) rowbits++;
      if (kbits + rowbits <= 64) {
        kp.kbits = kbits;
        kp.kmask = (1ull << kbits) - 1;
        kp.kmin = omin ^ (1ull << 63);
        if (sampled) { // keys outside the sampled range: outlier list (= the overflow rows of the bucket pass), a full list = rerun
          kp.oob = (unsigned int *)(ctr->as<uint64_t>() + 4);
          out->ov_rows = ctx->alloc(4 * (size_t)std::max<int64_t>(n, 1));
          kp.ov_rows = out->ov_rows->as<uint32_t>();
          kp.ov_count = ctr->as<unsigned long long>() + 1;
          kp.ov_cap = (uint32_t)std::min<int64_t>(n, std::max<int64_t>(65536, n / 1024));
        }
      }
    }
  }
  // LDS budget per workgroup and fill target.  72 KiB (two 512-thread workgroups per CU) filled
  // to 27 %: the probe loops of a wave run as long as its unluckiest lane, and at 55 % fill
  // (36 KiB tables, four workgroups per CU) that was ~5 trips per row instead of ~2:
  // 3.7 ms vs 3.0 ms for the C5 bucket pass.  Fewer groups per table would need more buckets,
  // which costs more in the partition passes than it saves here.
  const char *lds_e = hook("SQLRS_LDS_AGG_KB"); // tuning hook, read per call
  const size_t lds_budget = (size_t)(lds_e ? std::atoi(lds_e) : 72) * 1024;
  const size_t slot_bytes = 8 + 8 * (size_t)spec.n_acc + 4; // key, accumulators, first row
  uint32_t cap = 1;
  while ((size_t)(cap * 2 + 2) * slot_bytes <= lds_budget) cap *= 2;
  static const double load_factor = [] { // tuning hook
    const char *e = hook("SQLRS_LDS_LOAD");
    return e ? std::atof(e) : 0.275;
  }();
  const double groups_per_table = cap * load_factor;
  double want = est * 1.15 / groups_per_table;
  static const double max_frac = [] { // tuning hook: largest groups / rows ratio taken by this route
    const char *e = hook("SQLRS_PART_MAX_FRAC");
    return e ? std::atof(e) : 1.0;
  }();
  // (mostly distinct keys used to be sent to the row route, "est > n / 2": 17 ms instead of 1.7 ms
  // for 1e7 rows with 8e6 groups — the bucket count limit below is the only size limit now)
  if (!join_mode) est = std::min(est, (double)n);
  if (!join_mode && est > max_frac * (double)n) return false;
  // Dense keys: when the keys of interest fill most of their range (surrogate keys, dimension
  // primary keys) the partition is by key range and the bucket tables are addressed directly:
  // R = 2^rbits slots of 8 * n_acc + 4 bytes at 100 % fill instead of cap slots of 8 more bytes
  // at 27 %, i.e. ~5x fewer buckets (often one partition level instead of two), no probing and,
  // for the fused join, no build-side partition and no insert phase.
  const char *dense_e = hook("SQLRS_DENSE_AGG"); // test / tuning hook, read per call
  const bool dense_on = !(dense_e && dense_e[0] == '0');
  const uint64_t *partner_bits = nullptr; // dense fused join over build keys with gaps (PartAggInput::join_bits)
  const double want_hashed = want; // probing tables needed if the bucket pass ran on hashed buckets
  bool dense = false;
  if (dense_on && kp.kbits) {
    const uint64_t range = omax - omin;
    // Direct-addressed tables have no fill target to keep and nothing to probe: one table per CU, as large as LDS
    // allows, halves the bucket count (C5: 4883 -> 2442 buckets of 4096 slots, 39 x 128 -> 39 x 64 digits: level 2
    // 3.35 -> 3.07 ms, the bucket pass itself 1.63 -> 1.60 ms with one workgroup per CU instead of three; C4: 489 ->
    // 245 buckets, scatter 2.12 -> 1.52 ms).  8192 slots (first row kept inside the COUNT cell, 16-byte slots) were
    // measured too: level 1 -0.1 ms, bucket pass +0.13 ms, nothing gained.
    const char *dk_e = hook("SQLRS_LDS_DENSE_KB"); // tuning hook, read per call
    const size_t dense_budget = (size_t)(dk_e ? std::atoi(dk_e) : 150) * 1024;
    const size_t dslot = 8 * (size_t)spec.n_acc + 4;
    uint32_t rbits = 8;
    while (rbits < 14 && ((size_t)2 << rbits) * dslot <= dense_budget) rbits++;
    while (rbits > 8 && (range >> (rbits - 1)) == 0) rbits--; // a narrow key range: no larger than it needs (occupancy)
    const uint64_t pd = (range >> rbits) + 1;
    // join: unique build keys (the caller's pre-condition) that span exactly join_n values are
    // every value of the range; otherwise: at most ~4 slots per group
    // (with the existence bitmap of the range — the join's direct-address table — gaps are fine: <= 16 slots per key)
    const bool fills = join_mode ? (in.join_unique_known && (range + 1 == (uint64_t)in.join_n ||
                                                             ((in.join_bits || in.join_mult) && in.join_range_known && range / 16 <= (uint64_t)in.join_n)))
                                 : ((double)range + 1.0 <= 4.0 * est);
    partner_bits = (join_mode && fills && !in.join_mult && range + 1 != (uint64_t)in.join_n) ? in.join_bits : nullptr;
    if (fills && pd <= 65536 && (double)pd <= 1.5 * std::max(1.0, std::ceil(want))) {
      dense = true;
      kp.dense = 1;
      kp.rbits = rbits;
      kp.range = range;
      cap = 1u << rbits;
      want = (double)pd;
    }
  }
  if (join_mode && in.join_mult && !dense) return false; // duplicate build keys: the direct-addressed route or nothing
  if (want > 65536.0) return false; // too many groups: resolve path
  uint32_t P = (uint32_t)std::max(1.0, std::ceil(want));
  out->est_groups = est;
  // 2./3. rows in bucket order (LDS-staged multi-split, radix_part.hip)
  PartitionInput pin;
  pin.pack = kp;
  pin.filter = in.filter;
  pin.keys = in.keys;
  pin.key_validity = in.key_validity;
  pin.n = n;
  pin.nv = spec.nv;
  for (int k = 0; k < 2; k++) {
    pin.vals[k] = in.vals[k];
    pin.val_validity[k] = in.val_validity[k];
  }
  PartitionedRows pr;
  // One bucket (few groups) and nothing nullable: there is nothing to partition — the bucket pass
  // reads the caller's columns in place (row id = row index), cut into chunks by the few-buckets rule
  // below.  Saves the histogram and scatter passes: 50 int64 groups over 5e7 rows 1.12 -> 0.6 ms.
  const bool in_place = P == 1 && !join_mode && !nullable && want_hashed <= 1.0 && !in.filter.col; // (a row filter needs a partition pass)
  if (in_place) {
    dense = false;
    kp = KeyPack();
    cap = 1; // (dense mode had set its own)
    while ((size_t)(cap * 2 + 2) * slot_bytes <= lds_budget) cap *= 2;
    pr.n = n;
    pr.P = 1;
    auto borrowed = [&](const void *p) {
      BufP b = std::make_shared<Buf>(ctx, const_cast<void *>(p), 0);
      b->owned = false;
      return b;
    };
    pr.key = borrowed(in.keys);
    pr.v0 = spec.nv >= 1 ? borrowed(in.vals[0]) : nullptr;
    pr.v1 = spec.nv >= 2 ? borrowed(in.vals[1]) : nullptr;
    pr.bstart_host = {0u, (uint32_t)n};
    pr.bstart = nullptr;
  } else if (!partition_rows(ctx, pin, P, &pr)) {
    return false;
  }
  P = pr.P;
  n = pr.n; // (a fused row filter dropped the rest)
  if (dense != (pr.pack.dense != 0)) return false;
  if (n == 0) { // every row failed the fused filter: no groups
    out->groups = out->n_overflow = 0;
    out->gcap = 1;
    out->buckets = (int)P;
    out->gkey = ctx->alloc(8);
    out->gfirst = ctx->alloc(4);
    out->gacc = ctx->alloc(8 * (size_t)std::max(spec.n_acc, 1));
    out->ov_rows = ctx->alloc(4);
    out->row_ids = nullptr;
    out->row_offset = row_offset;
    return true;
  }
  out->buckets = (int)P;
  BufP pk = pr.key, pi = pr.idx, pv0 = pr.v0, pv1 = pr.v1, pf = pr.flags;
  PartitionedRows local_build;
  const PartitionedRows *bp = nullptr;
  if (join_mode && !dense) { // the build keys through the same bucket function (cached across probe batches)
    if (in.join_cache && in.join_cache->P == P && in.join_cache->n == in.join_n) {
      bp = in.join_cache;
    } else {
      PartitionInput bin;
      bin.keys = in.join_keys;
      bin.key_validity = in.join_validity;
      bin.n = in.join_n;
      bin.nv = 0;
      bin.build_side = true;
      PartitionedRows *dst = in.join_cache ? in.join_cache : &local_build;
      if (!partition_rows(ctx, bin, P, dst) || dst->P != P) return false;
      bp = dst;
    }
  }
  // 4. LDS aggregation, one workgroup per bucket
  LdsAggParams prm;
  prm.n_acc = spec.n_acc;
  for (int a = 0; a < PART_MAX_ACC; a++) {
    int kind = AK_COUNT;
    if (a < spec.n_acc) {
      switch (spec.op[a]) {
      case PART_COUNT: kind = AK_COUNT; break;
      case PART_SUM_I64: kind = AK_SUM_I64; break;
      case PART_SUM_F64: kind = AK_SUM_F64; break;
      case PART_MIN: kind = spec.kind[a] ? AK_MIN_F64 : AK_MIN_I64; break;
      default: kind = spec.kind[a] ? AK_MAX_F64 : AK_MAX_I64;
      }
    }
    prm.code[a] = a < spec.n_acc ? (kind | (spec.src[a] << 3)) : 0;
  }
  prm.cap = cap;
  prm.partner_bits = dense ? (const unsigned long long *)partner_bits : nullptr;
  prm.partner_mult = (dense && join_mode) ? in.join_mult : nullptr;
  {
    const char *seg_e = hook("SQLRS_AGG_SEG");
    prm.seg_off = (seg_e && seg_e[0] == '0') ? 1 : 0;
  }
  int64_t gcap = (int64_t)std::min<double>((double)n, est * 1.5 + 65536.0 + 2.0 * P);
  if (dense) gcap = (int64_t)std::min<uint64_t>((uint64_t)n, join_mode ? (uint64_t)in.join_n : kp.range + 1); // one group per key of the range (build key) at most
  size_t lds = dense ? round_up((size_t)cap * (slot_bytes - 8), 16) : round_up((size_t)(cap + 2) * slot_bytes, 16);
  // work list: buckets larger than `chunk` rows (key skew) are split so that no workgroup streams
  // more than `chunk` rows.  The chunks of a split bucket merge their tables into one small global
  // table per bucket (SplitTables), so a key is still emitted exactly once.
  const std::vector<uint32_t> &hb = pr.bstart_host;
  if (hb.size() != (size_t)P + 1) return false;
  uint32_t chunk = (uint32_t)std::max<int64_t>(32768, 2 * (n / std::max<uint32_t>(P, 1)));
  uint32_t split_above = chunk;
  if (dense) {
    // A range partition has few, large buckets (about one per workgroup slot of the chip for the
    // 2e8-row C4 batch), so one oversized bucket is a long tail: buckets more than a quarter above
    // the average are cut into half-average chunks and the work list is sorted by size, largest
    // first.  Merging a chunk into its bucket's direct-addressed global table is cheap (no probing).
    // (average over the NON-EMPTY buckets: an optimistically widened key range has empty buckets at both ends, and an
    //  average diluted by them made every real bucket look oversized — C4 bucket pass 0.65 -> 0.87 ms)
    int64_t nonempty_d = 0;
    for (uint32_t bkt = 0; bkt < P; bkt++) nonempty_d += pr.bucket_end(bkt) > hb[bkt];
    const int64_t avg = n / std::max<int64_t>(nonempty_d, 1);
    const char *cd_e = hook("SQLRS_DENSE_CHUNK_DIV"), *sa_e = hook("SQLRS_DENSE_SPLIT_PCT"); // tuning hooks, read per call
    const int cdiv = cd_e ? std::max(1, std::atoi(cd_e)) : 2;        // chunk = average / this
    const int sa_pct = sa_e ? std::max(1, std::atoi(sa_e)) : 125;    // buckets above this % of the average are split
    chunk = (uint32_t)std::max<int64_t>(32768, avg / cdiv);
    split_above = (uint32_t)std::max<int64_t>(65536, avg * sa_pct / 100);
  }
  {
    // Few buckets (few groups: GROUP BY state, a flag, a date part): without this the whole batch is
    // a handful of work items — one workgroup streamed 5e7 rows of a 50-group batch alone (26 ms).
    // With fewer non-empty buckets than half the workgroup slots of the chip, every bucket is cut
    // into chunks of 1/(2 x slots) of the batch; their tables merge through the split tables.
    const uint32_t slots = (lds > 80 * 1024 - 64 ? 1u : 2u) * (uint32_t)ctx->num_cus; // (tables of >= 80 KiB: one workgroup per CU)
    uint32_t nonempty = 0;
    for (uint32_t bkt = 0; bkt < P; bkt++) nonempty += pr.bucket_end(bkt) > hb[bkt];
    if (nonempty < slots / 2) {
      chunk = (uint32_t)std::max<int64_t>(32768, n / (2 * (int64_t)slots));
      split_above = chunk;
    }
  }
  if (pr.slim.on && !pr.slim.blk_bt) { // slim rows: a work item's run bits live in LDS next to its table (12 bytes per 64 rows)
    const size_t table = round_up((size_t)cap * (slot_bytes - 8), 16);
    const size_t room = 159 * 1024 > table + 4 * SLIM_BT_CAP ? 159 * 1024 - table - 4 * SLIM_BT_CAP : 0; // (160 KiB per workgroup, static LDS included)
    const uint32_t item_rows = (uint32_t)std::min<size_t>(1u << 18, (room / 12 > 2 ? room / 12 - 2 : 0) * 64);
    if (item_rows < 16384) return false; // (cannot happen: direct-addressed tables take <= 150 KiB)
    chunk = std::min(chunk, item_rows);
    split_above = std::min(split_above, item_rows);
  }
  std::vector<uint32_t> work, split_bucket, chunk_lo; // dense: chunk_lo[t] = first chunk table of split bucket t
  work.reserve(4 * ((size_t)P + 64));
  out->may_dup = false;
  uint32_t nsplit = 0, nchunk_tables = 0;
  for (uint32_t bkt = 0; bkt < P; bkt++) {
    uint32_t lo = hb[bkt], hi = pr.bucket_end(bkt); // (a claimed partition: slots of the bucket's region, sentinel rows included)
    if (dense && lo == hi) continue; // nothing to set up for an empty bucket (no build keys to insert)
    if (hi - lo <= split_above) {
      work.insert(work.end(), {bkt, lo, hi, 0xffffffffu});
    } else if (dense) { // every chunk stores its own table; the emit kernel reduces them
      chunk_lo.push_back(nchunk_tables);
      // (equal chunks: a remainder chunk of a few rows would still cost a whole table)
      const uint32_t pieces = (uint32_t)ceil_div((int64_t)(hi - lo), (int64_t)chunk);
      for (uint32_t q = 0; q < pieces; q++) {
        const uint32_t c0 = lo + (uint32_t)((uint64_t)(hi - lo) * q / pieces), c1 = lo + (uint32_t)((uint64_t)(hi - lo) * (q + 1) / pieces);
        work.insert(work.end(), {bkt, c0, c1, nchunk_tables++});
      }
      split_bucket.push_back(bkt);
      nsplit++;
    } else {
      for (uint32_t c0 = lo; c0 < hi; c0 += chunk) work.insert(work.end(), {bkt, c0, std::min(hi, c0 + chunk), nsplit});
      split_bucket.push_back(bkt);
      nsplit++;
    }
  }
  chunk_lo.push_back(nchunk_tables);
  if (nsplit) { // largest work items first
    std::vector<uint32_t> ord(work.size() / 4), sorted(work.size());
    for (uint32_t i = 0; i < ord.size(); i++) ord[i] = i;
    std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) {
      return work[4 * a + 2] - work[4 * a + 1] > work[4 * b + 2] - work[4 * b + 1];
    });
    for (size_t i = 0; i < ord.size(); i++) std::memcpy(&sorted[4 * i], &work[4 * ord[i]], 16);
    work.swap(sorted);
  }
  const uint32_t nwork = (uint32_t)(work.size() / 4);
  BufP dwork = ctx->alloc(4 * work.size() + 16);
  SQ_HIP(hipMemcpyAsync(dwork->p, work.data(), 4 * work.size(), hipMemcpyHostToDevice, ctx->stream));
  const uint32_t nslots_h = dense ? cap : cap + 2;
  SplitTables stb;
  BufP stb_key, stb_first, stb_acc, dsplit;
  BufP dchunk_lo;
  if (nsplit && dense) { // one table per CHUNK (stored whole by its workgroup: nothing to initialise)
    const int64_t total = (int64_t)nchunk_tables * nslots_h;
    stb_first = ctx->alloc(4 * (size_t)total);
    stb_acc = ctx->alloc(8 * (size_t)total * (size_t)std::max(spec.n_acc, 1));
    stb.first = stb_first->as<unsigned int>();
    stb.acc = stb_acc->as<unsigned long long>();
    stb.nsplit = nchunk_tables;
    dsplit = ctx->alloc(4 * (size_t)nsplit);
    dchunk_lo = ctx->alloc(4 * ((size_t)nsplit + 1));
    SQ_HIP(hipMemcpyAsync(dsplit->p, split_bucket.data(), 4 * (size_t)nsplit, hipMemcpyHostToDevice, ctx->stream));
    SQ_HIP(hipMemcpyAsync(dchunk_lo->p, chunk_lo.data(), 4 * ((size_t)nsplit + 1), hipMemcpyHostToDevice, ctx->stream));
  } else if (nsplit) {
    const int64_t total = (int64_t)nsplit * nslots_h;
    stb_key = ctx->alloc(8 * (size_t)total);
    stb_first = ctx->alloc(4 * (size_t)total);
    stb_acc = ctx->alloc(8 * (size_t)total * (size_t)std::max(spec.n_acc, 1));
    stb.key = stb_key->as<unsigned long long>();
    stb.first = stb_first->as<unsigned int>();
    stb.acc = stb_acc->as<unsigned long long>();
    stb.nsplit = nsplit;
    split_init_kernel<<<dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, ctx->stream>>>(stb, total, spec.n_acc, prm);
    SQ_HIP(hipGetLastError());
  }
  out->gkey = ctx->alloc(8 * (size_t)gcap);
  out->gfirst = ctx->alloc(4 * (size_t)gcap);
  out->gvalid = in.key_validity ? ctx->alloc((size_t)gcap) : nullptr;
  out->gacc = ctx->alloc(8 * (size_t)gcap * (size_t)std::max(spec.n_acc, 1));
  out->gcap = gcap;
  if (!out->ov_rows) out->ov_rows = ctx->alloc(4 * (size_t)std::max<int64_t>(n, 1)); // (n: rows that passed the fused filter)
  {
    ProfScope ps(ctx, "lds_agg");
#define SQ_LA(NV, FL, JN, NA, C0, C1, PK)                                                                        \
  do {                                                                                                         \
    auto kfn = lds_agg_kernel<NV, FL, JN, NA, C0, C1, PK>;                                                      \
    constexpr bool can_rec = NV == 1 && PK && !FL;                                                             \
    if (can_rec && pr.rec) kfn = lds_agg_kernel<NV, FL, JN, NA, C0, C1, PK, can_rec>;                          \
    else if (pr.rec) fail(SQLRS_ERR_INTERNAL, "record-form partition in a column-form bucket pass");           \
    allow_big_lds(ctx, kfn);                                                                                   \
    kfn<<<dim3(nwork), dim3(PART_WG), lds, ctx->stream>>>(                                                     \
        prm, pr.rec ? pr.rec->as<uint64_t>() : pk->as<uint64_t>(), pi ? pi->as<uint32_t>() : nullptr, pv0 ? pv0->as<uint64_t>() : nullptr,       \
        pv1 ? pv1->as<uint64_t>() : nullptr, pf ? pf->as<uint8_t>() : nullptr, dwork->as<uint32_t>(), P, n,    \
        ctr->as<unsigned long long>(), out->gkey->as<uint64_t>(), out->gfirst->as<uint32_t>(),                 \
        out->gvalid ? out->gvalid->as<uint8_t>() : nullptr, out->gacc->as<uint64_t>(), gcap,                   \
        ctr->as<unsigned long long>() + 1, out->ov_rows->as<uint32_t>(),                                       \
        bp ? bp->key->as<uint64_t>() : nullptr, (bp && bp->flags) ? bp->flags->as<uint8_t>() : nullptr,        \
        bp ? bp->bstart->as<uint32_t>() : nullptr, pr.pack, stb);                                              \
    launched = true;                                                                                           \
  } while (0)
#define SQ_LA_J(NV, FL, NA, C0, C1)                                                                            \
  do {                                                                                                         \
    if (pr.pack.kbits) {                                                                                       \
      if (join_mode) SQ_LA(NV, FL, true, NA, C0, C1, true);                                                    \
      else SQ_LA(NV, FL, false, NA, C0, C1, true);                                                             \
    } else {                                                                                                   \
      if (join_mode) SQ_LA(NV, FL, true, NA, C0, C1, false);                                                   \
      else SQ_LA(NV, FL, false, NA, C0, C1, false);                                                            \
    }                                                                                                          \
  } while (0)
    const int nvu = pv1 ? 2 : ((pv0 || pr.rec) ? 1 : 0);
    bool launched = false;
    if (dense && pr.slim.on) { // slim rows (radix_part.hpp): value + 32-bit word, row ids rebuilt from the runs
      uint32_t max_item = 64;
      for (uint32_t i = 0; i < nwork; i++) max_item = std::max(max_item, work[4 * i + 2] - work[4 * i + 1]);
      const bool blk = pr.slim.blk_bt != nullptr; // rows of a claimed single level: base tiles per block, no run lists
      SlimBucketIn sb;
      sb.rows = pr.slim.rows;
      sb.nzstart = blk ? nullptr : pr.slim.nzstart->as<uint32_t>();
      sb.nzbt = blk ? nullptr : pr.slim.nzbt->as<uint32_t>();
      sb.nzcount = blk ? nullptr : pr.slim.nzcount->as<uint32_t>();
      sb.bcol = blk ? nullptr : pr.slim.bcol->as<uint32_t>();
      sb.tile = pr.slim.tile;
      sb.groups = blk ? 0u : (uint32_t)ceil_div((int64_t)max_item, 64) + 1;
      sb.blk_bt = blk ? pr.slim.blk_bt->as<uint32_t>() : nullptr;
      sb.log_b = pr.slim.log_b;
      const size_t slds = round_up((size_t)cap * (slot_bytes - 8), 16) + (blk ? 16 : 12 * (size_t)sb.groups + 4 * SLIM_BT_CAP);
#define SQ_LS(JN, NA, C0, C1)                                                                                  \
  do {                                                                                                         \
    auto kfn = lds_agg_dense_slim_kernel<JN, NA, C0, C1>;                                                       \
    if (blk) kfn = lds_agg_dense_slim_kernel<JN, NA, C0, C1, true>;                                             \
    allow_big_lds(ctx, kfn, 159 * 1024);                                                                                 \
    kfn<<<dim3(nwork), dim3(DENSE_WG), slds, ctx->stream>>>(prm, sb, dwork->as<uint32_t>(), ctr->as<unsigned long long>(), \
                                                           out->gkey->as<uint64_t>(), out->gfirst->as<uint32_t>(), \
                                                           out->gacc->as<uint64_t>(), gcap, pr.pack, stb);        \
    launched = true;                                                                                           \
  } while (0)
#define SQ_LS_J(NA, C0, C1) do { if (join_mode) SQ_LS(true, NA, C0, C1); else SQ_LS(false, NA, C0, C1); } while (0)
      const int c0 = prm.code[0], c1 = spec.n_acc == 2 ? prm.code[1] : -1;
      if (spec.n_acc == 2 && c0 == AK_COUNT && c1 == AK_SUM_F64) SQ_LS_J(2, AK_COUNT, AK_SUM_F64);
      else if (spec.n_acc == 2 && c0 == AK_SUM_F64 && c1 == AK_COUNT) SQ_LS_J(2, AK_SUM_F64, AK_COUNT);
      else if (spec.n_acc == 1 && c0 == AK_SUM_F64) SQ_LS_J(1, AK_SUM_F64, 0);
      else if (spec.n_acc == 1 && c0 == AK_COUNT) SQ_LS_J(1, AK_COUNT, 0);
      else SQ_LS_J(-1, 0, 0);
#undef SQ_LS_J
#undef SQ_LS
      if (nsplit) {
        split_emit_dense_kernel<<<dim3((unsigned)(nsplit * ceil_div((int64_t)nslots_h, SED_SLOTS))), dim3(SED_SLOTS * SED_GROUPS), 0, ctx->stream>>>(
            stb, dsplit->as<uint32_t>(), dchunk_lo->as<uint32_t>(), nsplit, cap, pr.pack, prm, spec.n_acc,
            ctr->as<unsigned long long>(), out->gkey->as<uint64_t>(), out->gfirst->as<uint32_t>(), out->gacc->as<uint64_t>(), gcap);
      }
    } else if (dense) { // direct-addressed tables (packed rows, nothing nullable, at most one value column)
#define SQ_LD(NV, JN, NA, C0, C1)                                                                              \
  do {                                                                                                         \
    auto kfn = lds_agg_dense_kernel<NV, JN, NA, C0, C1>;                                                        \
    if (NV == 1 && pr.rec) kfn = lds_agg_dense_kernel<NV, JN, NA, C0, C1, NV == 1>;                            \
    allow_big_lds(ctx, kfn);                                                                                   \
    kfn<<<dim3(nwork), dim3(DENSE_WG), lds, ctx->stream>>>(                                                    \
        prm, pr.rec ? pr.rec->as<uint64_t>() : pk->as<uint64_t>(), pv0 ? pv0->as<uint64_t>() : nullptr, dwork->as<uint32_t>(), \
        ctr->as<unsigned long long>(), out->gkey->as<uint64_t>(), out->gfirst->as<uint32_t>(),                 \
        out->gacc->as<uint64_t>(), gcap, pr.pack, stb);                                                        \
    launched = true;                                                                                           \
  } while (0)
#define SQ_LD_J(NV, NA, C0, C1) do { if (join_mode) SQ_LD(NV, true, NA, C0, C1); else SQ_LD(NV, false, NA, C0, C1); } while (0)
      if (nvu == 1 && spec.n_acc <= 2) {
        const int c0 = prm.code[0], c1 = spec.n_acc == 2 ? prm.code[1] : -1;
#define SQ_SIG1(K) if (!launched && spec.n_acc == 1 && c0 == K) SQ_LD_J(1, 1, K, 0)
#define SQ_SIG2(K0, K1) if (!launched && spec.n_acc == 2 && c0 == K0 && c1 == K1) SQ_LD_J(1, 2, K0, K1)
        SQ_SIG1(AK_COUNT); SQ_SIG1(AK_SUM_I64); SQ_SIG1(AK_SUM_F64); SQ_SIG1(AK_MIN_I64); SQ_SIG1(AK_MIN_F64);
        SQ_SIG1(AK_MAX_I64); SQ_SIG1(AK_MAX_F64);
        SQ_SIG2(AK_COUNT, AK_SUM_F64); SQ_SIG2(AK_SUM_F64, AK_COUNT);
        SQ_SIG2(AK_COUNT, AK_SUM_I64); SQ_SIG2(AK_SUM_I64, AK_COUNT);
        SQ_SIG2(AK_MIN_F64, AK_MAX_F64); SQ_SIG2(AK_MIN_I64, AK_MAX_I64);
#undef SQ_SIG1
#undef SQ_SIG2
      }
      if (!launched) { if (nvu == 0) SQ_LD_J(0, -1, 0, 0); else SQ_LD_J(1, -1, 0, 0); }
#undef SQ_LD_J
#undef SQ_LD
      if (nsplit) {
        split_emit_dense_kernel<<<dim3((unsigned)(nsplit * ceil_div((int64_t)nslots_h, SED_SLOTS))), dim3(SED_SLOTS * SED_GROUPS), 0, ctx->stream>>>(
            stb, dsplit->as<uint32_t>(), dchunk_lo->as<uint32_t>(), nsplit, cap, pr.pack, prm, spec.n_acc,
            ctr->as<unsigned long long>(), out->gkey->as<uint64_t>(), out->gfirst->as<uint32_t>(), out->gacc->as<uint64_t>(), gcap);
      }
    }
    // specialised kernels: no nullable column, one value column, the usual accumulator lists
    if (!launched && !pf && nvu == 1 && spec.n_acc <= 2) {
      const int c0 = prm.code[0], c1 = spec.n_acc == 2 ? prm.code[1] : -1;
#define SQ_SIG1(K) if (!launched && spec.n_acc == 1 && c0 == K) SQ_LA_J(1, false, 1, K, 0)
#define SQ_SIG2(K0, K1) if (!launched && spec.n_acc == 2 && c0 == K0 && c1 == K1) SQ_LA_J(1, false, 2, K0, K1)
      SQ_SIG1(AK_COUNT); SQ_SIG1(AK_SUM_I64); SQ_SIG1(AK_SUM_F64); SQ_SIG1(AK_MIN_I64); SQ_SIG1(AK_MIN_F64);
      SQ_SIG1(AK_MAX_I64); SQ_SIG1(AK_MAX_F64);
      SQ_SIG2(AK_COUNT, AK_SUM_F64); SQ_SIG2(AK_SUM_F64, AK_COUNT);
      SQ_SIG2(AK_COUNT, AK_SUM_I64); SQ_SIG2(AK_SUM_I64, AK_COUNT);
      SQ_SIG2(AK_MIN_F64, AK_MAX_F64); SQ_SIG2(AK_MIN_I64, AK_MAX_I64);
#undef SQ_SIG1
#undef SQ_SIG2
    }
    if (!launched) { // generic: the accumulator list is interpreted per row
      if (pf) { if (nvu == 0) SQ_LA_J(0, true, -1, 0, 0); else if (nvu == 1) SQ_LA_J(1, true, -1, 0, 0); else SQ_LA_J(2, true, -1, 0, 0); }
      else { if (nvu == 0) SQ_LA_J(0, false, -1, 0, 0); else if (nvu == 1) SQ_LA_J(1, false, -1, 0, 0); else SQ_LA_J(2, false, -1, 0, 0); }
    }
#undef SQ_LA_J
#undef SQ_LA
    if (nsplit && !dense) {
      const int64_t total = (int64_t)nsplit * nslots_h;
      split_emit_kernel<<<dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, ctx->stream>>>(
          stb, nslots_h, cap, spec.n_acc, ctr->as<unsigned long long>(), out->gkey->as<uint64_t>(),
          out->gfirst->as<uint32_t>(), out->gvalid ? out->gvalid->as<uint8_t>() : nullptr,
          out->gacc->as<uint64_t>(), gcap);
    }
    SQ_HIP(hipGetLastError());
  }
  // (`work` on the host was the source of an async upload: the fetch below synchronises)
  const uint64_t *h = (const uint64_t *)ctx->fetch(ctr->p, 40);
  if (h[4]) { // a key outside the sampled range was packed as the sentinel: nothing of this attempt is valid
    out->retry_exact = true;
    return false;
  }
  out->groups = (int64_t)h[0];
  out->n_overflow = (int64_t)h[1];
  out->may_dup = h[3] != 0; // a split bucket's global table was full: some keys were emitted twice
  if (join_mode && h[2]) return false; // a bucket table could not hold its build keys
  if (out->groups > gcap) return false; // estimate far too low: caller falls back to the resolve path
  // NULL-key bitmap for the merge (the first rows as global row numbers are made when a merge asks: part_row_ids)
  int64_t g1 = std::max<int64_t>(out->groups, 1);
  out->row_offset = row_offset;
  out->row_ids = nullptr;
  if (out->groups) {
    if (out->gvalid) {
      out->gvalid_bits = ctx->alloc(bitmap_bytes(g1));
      int64_t g64 = (int64_t)round_up((size_t)g1, 64);
      bytes_pack_kernel<<<dim3((unsigned)ceil_div(g64, 256)), dim3(256), 0, ctx->stream>>>(
          out->gvalid->as<uint8_t>(), out->groups, out->gvalid_bits->as<uint64_t>());
    }
    SQ_HIP(hipGetLastError());
  }
  return true;
}

} // namespace sq
