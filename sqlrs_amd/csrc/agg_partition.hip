// agg_partition.hip — LDS-partitioned pre-aggregation for large HashAgg batches.
//
// Why: MI355X executes ~24 G random global atomics/s but ~1500 G LDS atomics/s
// (profiles/r01_ubench_mi355x.txt), so a 2e8-row group-by done with global atomics is
// 60x off the HBM roofline.  Rows are therefore radix-partitioned by key hash into P buckets
// small enough that each bucket's groups fit one workgroup's LDS hash table:
//
//   key_stats   : HyperLogLog over the keys -> estimated group count -> P        ( 8 B/row read)
//   part_hist   : per tile (65536 rows) bucket histogram -> matrix [P][tiles]    ( 8 B/row read)
//   scan        : exclusive scan of the matrix = every (bucket, tile) run start
//   part_scatter: rows -> (key, row id, values) in bucket order                  (8+8v read, 12+8v written)
//   lds_agg     : one workgroup per bucket: open-addressing table in LDS (64-bit ds CAS to
//                 claim a slot, ds_add/ds_min/ds_max on the accumulator cells), then the
//                 occupied slots are written out as (key, first row, accumulators)  (12+8v read)
//
// The per-batch groups are merged into the operator state by the ordinary resolve path
// (agg.hip) with explicit first-row ids and pre-aggregated weights: O(groups) atomics.
// Rows whose bucket table overflows (estimate too low) are returned to the caller and take
// the resolve path directly, so the result never depends on the estimate.
#include <cstdlib>

#include "agg_partition.hpp"
#include "device_utils.hpp"
#include "radix_part.hpp"

namespace sq {

constexpr int PART_WG = 512;
constexpr uint64_t LDS_EMPTY = ~0ull;

__device__ __forceinline__ uint32_t bucket_of(uint64_t h, uint32_t P) {
  return (uint32_t)__umul64hi(h, (uint64_t)P);
}

// ---------------------------------------------------------------- key statistics --
__global__ __launch_bounds__(PART_WG) void key_stats_kernel(const uint64_t *__restrict__ keys,
                                                            const uint64_t *__restrict__ validity,
                                                            int64_t n, unsigned int *__restrict__ hll) {
  __shared__ unsigned int reg[4096];
  for (int i = threadIdx.x; i < 4096; i += PART_WG) reg[i] = 0;
  __syncthreads();
  for (int64_t r = blockIdx.x * (int64_t)PART_WG + threadIdx.x; r < n; r += (int64_t)gridDim.x * PART_WG) {
    if (validity && !((validity[r >> 6] >> (r & 63)) & 1)) continue;
    uint64_t h = mix64(keys[r] ^ 0x2545f4914f6cdd1dULL);
    unsigned idx = (unsigned)(h >> 52);
    unsigned rank = (unsigned)__builtin_clzll((h << 12) | (1ull << 11)) + 1;
    if (reg[idx] < rank) atomicMax(&reg[idx], rank);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 4096; i += PART_WG)
    if (reg[i]) atomicMax(&hll[i], reg[i]);
}

double estimate_distinct(Ctx *ctx, const uint64_t *keys, const uint64_t *validity, int64_t n) {
  BufP hll = ctx->alloc_zero(4096 * 4);
  {
    ProfScope ps(ctx, "key_stats");
    unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(n, PART_WG * 16), 2048);
    key_stats_kernel<<<dim3(std::max(blocks, 1u)), dim3(PART_WG), 0, ctx->stream>>>(keys, validity, n,
                                                                                  hll->as<unsigned int>());
    SQ_HIP(hipGetLastError());
  }
  std::vector<unsigned int> reg(4096);
  SQ_HIP(hipMemcpyAsync(reg.data(), hll->p, 4096 * 4, hipMemcpyDeviceToHost, ctx->stream));
  ctx->sync();
  const double m = 4096.0;
  double sum = 0;
  int zeros = 0;
  for (unsigned r : reg) {
    sum += std::ldexp(1.0, -(int)r);
    zeros += (r == 0);
  }
  double e = (0.7213 / (1.0 + 1.079 / m)) * m * m / sum;
  if (e <= 2.5 * m && zeros) e = m * std::log(m / zeros);
  return e;
}

// ----------------------------------------------------------------- LDS aggregate --
struct LdsAggParams {
  int n_acc;
  int op[PART_MAX_ACC];   // PartOp
  int src[PART_MAX_ACC];  // value column
  int kind[PART_MAX_ACC]; // MIN/MAX: 0 i64, 1 f64
  int cells;              // 8-byte cells per slot = 2 + n_acc
  uint32_t cap;           // slots (power of two); +2 reserved slots follow
};

__device__ __forceinline__ uint64_t acc_identity_cell(int op) { return op == PART_MIN ? ~0ull : 0ull; }

__global__ __launch_bounds__(PART_WG) void lds_agg_kernel(
    LdsAggParams prm, const uint64_t *__restrict__ pk, const uint32_t *__restrict__ pi,
    const uint64_t *__restrict__ pv0, const uint64_t *__restrict__ pv1,
    const uint8_t *__restrict__ pf, const uint32_t *__restrict__ work /* {bucket, lo, hi} triples */, uint32_t P,
    int64_t n, unsigned long long *out_count, uint64_t *__restrict__ gkey,
    uint32_t *__restrict__ gfirst, uint8_t *__restrict__ gvalid, uint64_t *__restrict__ gacc,
    int64_t gcap, unsigned long long *ov_count, uint32_t *__restrict__ ov_rows,
    const uint64_t *__restrict__ bk, const uint8_t *__restrict__ bf, const uint32_t *__restrict__ bbstart,
    int join_mode) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long tab[];
  __shared__ unsigned int s_cnt;
  __shared__ unsigned long long s_base;
  // one work item = one bucket, or one chunk of an oversized (skewed) bucket
  const uint32_t b = work[3 * blockIdx.x];
  const int cells = prm.cells;
  const uint32_t cap = prm.cap, mask = cap - 1, nslots = cap + 2;
  for (uint32_t s = threadIdx.x; s < nslots; s += PART_WG) {
    unsigned long long *c = tab + (size_t)s * cells;
    c[0] = LDS_EMPTY;
    c[1] = ~0ull; // first row (low 32 bits used)
    for (int a = 0; a < prm.n_acc; a++) c[2 + a] = acc_identity_cell(prm.op[a]);
  }
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  if (join_mode) {
    // fused inner join: the bucket's BUILD keys are inserted first; probe rows then only
    // accumulate into slots that exist (a probe key without a build partner is dropped)
    const int64_t blo = bbstart[b], bhi = bbstart[b + 1];
    for (int64_t i = blo + threadIdx.x; i < bhi; i += PART_WG) {
      uint64_t key = bk[i];
      bool valid = bf ? (bf[i] & 1) : true;
      if (!valid) {
        tab[(size_t)cap * cells] = 1; // NULL build key present (NULL = NULL matches)
      } else if (key == LDS_EMPTY) {
        tab[(size_t)(cap + 1) * cells] = 1;
      } else {
        uint32_t s = (uint32_t)(mix64(key) >> 7) & mask;
        uint32_t probes = 0;
        while (true) {
          unsigned long long prev = atomicCAS(&tab[(size_t)s * cells], LDS_EMPTY, (unsigned long long)key);
          if (prev == LDS_EMPTY || prev == key) break;
          s = (s + 1) & mask;
          if (++probes >= cap) {
            atomicExch(ov_count + 1, 1ull); // table too small for the build side: caller falls back
            break;
          }
        }
      }
    }
    __syncthreads();
  }
  const int64_t lo = work[3 * blockIdx.x + 1];
  const int64_t hi = work[3 * blockIdx.x + 2];
  // 4 rows per thread per trip, all loads issued before the first dependent LDS op
  for (int64_t i0 = lo + threadIdx.x; i0 < hi; i0 += 4 * PART_WG) {
    uint64_t keys4[4], v04[4], v14[4];
    uint32_t idx4[4];
    uint8_t f4[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      int64_t i = i0 + (int64_t)u * PART_WG;
      bool in = i < hi;
      keys4[u] = in ? pk[i] : 0;
      idx4[u] = in ? pi[i] : 0;
      f4[u] = in ? (pf ? pf[i] : 7) : 0xff;
      v04[u] = (in && pv0) ? pv0[i] : 0;
      v14[u] = (in && pv1) ? pv1[i] : 0;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
    if (f4[u] == 0xff) continue;
    uint64_t key = keys4[u];
    uint32_t idx = idx4[u];
    uint8_t f = f4[u];
    uint32_t s;
    bool ok = true;
    if (!(f & 1)) {
      s = cap; // NULL keys: one group
      if (join_mode && tab[(size_t)s * cells] == LDS_EMPTY) continue;
    } else if (key == LDS_EMPTY) {
      s = cap + 1;
      if (join_mode && tab[(size_t)s * cells] == LDS_EMPTY) continue;
    } else if (join_mode) {
      s = (uint32_t)(mix64(key) >> 7) & mask;
      uint32_t probes = 0;
      bool found = false;
      while (probes++ < cap) {
        unsigned long long cur = tab[(size_t)s * cells];
        if (cur == key) {
          found = true;
          break;
        }
        if (cur == LDS_EMPTY) break;
        s = (s + 1) & mask;
      }
      if (!found) continue;
    } else {
      s = (uint32_t)(mix64(key) >> 7) & mask;
      uint32_t probes = 0;
      while (true) {
        unsigned long long cur = tab[(size_t)s * cells];
        if (cur == key) break;
        if (cur == LDS_EMPTY) {
          unsigned long long prev = atomicCAS(&tab[(size_t)s * cells], LDS_EMPTY, (unsigned long long)key);
          if (prev == LDS_EMPTY || prev == key) break;
        }
        s = (s + 1) & mask;
        if (++probes >= cap) { // table full: hand the row back to the caller
          ok = false;
          break;
        }
      }
    }
    if (!ok) {
      unsigned long long o = atomicAdd(ov_count, 1ull);
      ov_rows[o] = idx;
      continue;
    }
    unsigned long long *c = tab + (size_t)s * cells;
    atomicMin((unsigned int *)&c[1], idx);
    for (int a = 0; a < prm.n_acc; a++) {
      int src = prm.src[a];
      if (!(f & (2 << src))) continue; // NULL input is skipped by every accumulator
      uint64_t v = src ? v14[u] : v04[u];
      switch (prm.op[a]) {
      case PART_COUNT: atomicAdd(&c[2 + a], 1ull); break;
      case PART_SUM_I64: atomicAdd(&c[2 + a], (unsigned long long)v); break;
      case PART_SUM_F64: unsafeAtomicAdd((double *)&c[2 + a], __longlong_as_double((long long)v)); break;
      case PART_MIN:
        atomicMin(&c[2 + a], (unsigned long long)(prm.kind[a] ? f64_to_ordered(__longlong_as_double((long long)v))
                                                               : i64_to_ordered((int64_t)v)));
        break;
      default:
        atomicMax(&c[2 + a], (unsigned long long)(prm.kind[a] ? f64_to_ordered(__longlong_as_double((long long)v))
                                                               : i64_to_ordered((int64_t)v)));
      }
    }
    } // u
  }
  __syncthreads();
  // compact the occupied slots of this bucket into the global group list
  unsigned int mine = 0;
  for (uint32_t s = threadIdx.x; s < nslots; s += PART_WG) {
    const unsigned long long *c = tab + (size_t)s * cells;
    bool occ = (unsigned int)c[1] != 0xffffffffu;
    mine += occ;
  }
  unsigned int my_off = atomicAdd(&s_cnt, mine);
  __syncthreads();
  if (threadIdx.x == 0) s_base = atomicAdd(out_count, (unsigned long long)s_cnt);
  __syncthreads();
  unsigned long long base = s_base + my_off;
  for (uint32_t s = threadIdx.x; s < nslots; s += PART_WG) {
    const unsigned long long *c = tab + (size_t)s * cells;
    bool occ = (unsigned int)c[1] != 0xffffffffu;
    if (!occ) continue;
    if ((int64_t)base < gcap) {
      gkey[base] = s == cap + 1 ? LDS_EMPTY : c[0];
      gfirst[base] = (unsigned int)c[1];
      if (gvalid) gvalid[base] = s == cap ? 0 : 1;
      for (int a = 0; a < prm.n_acc; a++) gacc[(size_t)a * gcap + base] = c[2 + a];
    }
    base++;
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

bool partitioned_preaggregate(Ctx *ctx, const PartAggSpec &spec, const PartAggInput &in,
                              uint64_t row_offset, PartAggOutput *out) {
  const int64_t n = in.n;
  if (n > 0xffffffffll || spec.n_acc > PART_MAX_ACC || spec.nv > 2) return false;
  // 1. how many groups?  -> bucket count.  Fused join: every build key needs a slot.
  const bool join_mode = in.join_keys != nullptr;
  double est = join_mode ? (double)in.join_n : estimate_distinct(ctx, in.keys, in.key_validity, n);
  static const double est_scale = [] { // test hook: mis-scale the estimate to force the overflow path
    const char *e = std::getenv("SQLRS_EST_SCALE");
    return e ? std::atof(e) : 1.0;
  }();
  if (!join_mode) est = std::max(1.0, est * est_scale);
  const int cells = 2 + spec.n_acc;
  // LDS budget per workgroup: 36 KiB tables let four 512-thread workgroups share a CU
  static const size_t lds_budget = [] {
    const char *e = std::getenv("SQLRS_LDS_AGG_KB");
    return (size_t)(e ? std::atoi(e) : 36) * 1024;
  }();
  uint32_t cap = 1;
  while ((size_t)(cap * 2 + 2) * cells * 8 <= lds_budget) cap *= 2;
  const double groups_per_table = cap * 0.55;
  double want = est * 1.15 / groups_per_table;
  if (want > 65536.0 || (!join_mode && est > 0.5 * (double)n)) return false; // too many groups: resolve path
  uint32_t P = (uint32_t)std::max(1.0, std::ceil(want));
  out->est_groups = est;
  // 2./3. rows in bucket order (LDS-staged multi-split, radix_part.hip)
  PartitionInput pin;
  pin.keys = in.keys;
  pin.key_validity = in.key_validity;
  pin.n = n;
  pin.nv = spec.nv;
  for (int k = 0; k < 2; k++) {
    pin.vals[k] = in.vals[k];
    pin.val_validity[k] = in.val_validity[k];
  }
  PartitionedRows pr;
  if (!partition_rows(ctx, pin, P, &pr)) return false;
  P = pr.P;
  out->buckets = (int)P;
  BufP pk = pr.key, pi = pr.idx, pv0 = pr.v0, pv1 = pr.v1, pf = pr.flags;
  PartitionedRows local_build;
  const PartitionedRows *bp = nullptr;
  if (join_mode) { // the build keys through the same bucket function (cached across probe batches)
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
    prm.op[a] = a < spec.n_acc ? spec.op[a] : 0;
    prm.src[a] = a < spec.n_acc ? spec.src[a] : 0;
    prm.kind[a] = a < spec.n_acc ? spec.kind[a] : 0;
  }
  prm.cells = cells;
  prm.cap = cap;
  int64_t gcap = (int64_t)std::min<double>((double)n, est * 1.5 + 65536.0 + 2.0 * P);
  BufP ctr = ctx->alloc_zero(24);
  size_t lds = (size_t)(cap + 2) * cells * 8;
  static bool attr_set = false;
  if (!attr_set) {
    SQ_HIP(hipFuncSetAttribute((const void *)lds_agg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                               150 * 1024));
    attr_set = true;
  }
  // work list: buckets larger than `chunk` rows (key skew) are split so that no workgroup streams
  // more than `chunk` rows; the same key may then appear in several chunks (out->may_dup) and the
  // caller merges the groups instead of adopting them as they are
  std::vector<uint32_t> hb((size_t)P + 1);
  SQ_HIP(hipMemcpyAsync(hb.data(), pr.bstart->p, 4 * hb.size(), hipMemcpyDeviceToHost, ctx->stream));
  ctx->sync();
  const uint32_t chunk = (uint32_t)std::max<int64_t>(65536, 8 * (n / std::max<uint32_t>(P, 1)));
  std::vector<uint32_t> work;
  work.reserve(3 * ((size_t)P + 64));
  out->may_dup = false;
  for (uint32_t bkt = 0; bkt < P; bkt++) {
    uint32_t lo = hb[bkt], hi = hb[bkt + 1];
    if (hi - lo <= chunk) {
      work.insert(work.end(), {bkt, lo, hi});
    } else {
      out->may_dup = true;
      for (uint32_t c0 = lo; c0 < hi; c0 += chunk) work.insert(work.end(), {bkt, c0, std::min(hi, c0 + chunk)});
    }
  }
  const uint32_t nwork = (uint32_t)(work.size() / 3);
  BufP dwork = ctx->alloc(4 * work.size() + 16);
  SQ_HIP(hipMemcpyAsync(dwork->p, work.data(), 4 * work.size(), hipMemcpyHostToDevice, ctx->stream));
  gcap += (int64_t)(nwork - P) * (int64_t)(cap + 2); // every extra chunk can add a table's worth of partials
  out->gkey = ctx->alloc(8 * (size_t)gcap);
  out->gfirst = ctx->alloc(4 * (size_t)gcap);
  out->gvalid = in.key_validity ? ctx->alloc((size_t)gcap) : nullptr;
  out->gacc = ctx->alloc(8 * (size_t)gcap * (size_t)std::max(spec.n_acc, 1));
  out->gcap = gcap;
  out->ov_rows = ctx->alloc(4 * (size_t)n);
  {
    ProfScope ps(ctx, "lds_agg");
    lds_agg_kernel<<<dim3(nwork), dim3(PART_WG), lds, ctx->stream>>>(
        prm, pk->as<uint64_t>(), pi->as<uint32_t>(), pv0 ? pv0->as<uint64_t>() : nullptr,
        pv1 ? pv1->as<uint64_t>() : nullptr, pf ? pf->as<uint8_t>() : nullptr, dwork->as<uint32_t>(), P,
        n, ctr->as<unsigned long long>(), out->gkey->as<uint64_t>(), out->gfirst->as<uint32_t>(),
        out->gvalid ? out->gvalid->as<uint8_t>() : nullptr, out->gacc->as<uint64_t>(), gcap,
        ctr->as<unsigned long long>() + 1, out->ov_rows->as<uint32_t>(),
        bp ? bp->key->as<uint64_t>() : nullptr, (bp && bp->flags) ? bp->flags->as<uint8_t>() : nullptr,
        bp ? bp->bstart->as<uint32_t>() : nullptr, join_mode ? 1 : 0);
    SQ_HIP(hipGetLastError());
  }
  ctx->sync(); // `work` (host) was the source of an async upload
  const uint64_t *h = (const uint64_t *)ctx->fetch(ctr->p, 24);
  out->groups = (int64_t)h[0];
  out->n_overflow = (int64_t)h[1];
  if (join_mode && h[2]) return false; // a bucket table could not hold its build keys
  if (out->groups > gcap) return false; // estimate far too low: caller falls back to the resolve path
  // first-row ids as global row numbers + NULL-key bitmap for the merge
  int64_t g1 = std::max<int64_t>(out->groups, 1);
  out->row_ids = ctx->alloc(8 * (size_t)g1);
  if (out->groups) {
    first_to_rowid_kernel<<<dim3((unsigned)ceil_div(g1, 256)), dim3(256), 0, ctx->stream>>>(
        out->gfirst->as<uint32_t>(), out->groups, row_offset, out->row_ids->as<uint64_t>());
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
