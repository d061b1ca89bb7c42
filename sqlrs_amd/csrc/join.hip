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

#include "join_probe_kernels.hpp" // every kernel of this operator (namespace sq)

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
