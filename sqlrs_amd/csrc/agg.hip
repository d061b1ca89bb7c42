// agg.hip — HashAggExecutor on device (src/executor/aggregate/hash_agg.rs:32-150 and the
// accumulators of aggregate/{count,sum,min_max}.rs).
//
// State: an open-addressing table {key, first_row} in HBM plus DENSE per-group accumulator
// arrays indexed by a group id that is handed out in first-seen order, so finalize emits the
// accumulator arrays as they are (hash_agg.rs:98,132: groups in first-seen order).
//
// Per batch (resolve path):
//   1. agg_resolve   : row -> slot (find-or-insert, one CAS per new group) and
//                      first_row[slot] = min(first_row[slot], global row) (read-then-atomicMin)
//   2. agg_mark_first: rows that ARE their group's first row -> new groups, in row order
//   3. agg_assign_gid: new groups get the next dense ids; their key values are gathered once
//   4. agg_update_*  : acc[gid] (+)= value (global atomics; MI355X measures ~24 G atomics/s,
//                      see profiles/r01_ubench_mi355x.txt — large batches go through the
//                      LDS-partitioned pre-aggregation of agg_partition.hip first)
// Algorithmic HBM bytes (C4): 16 B per input row + 24 B per group.
#include "common.hpp"
#include "device_utils.hpp"
#include "prims.hpp"
#include "agg_state.hpp"

namespace sq {

__global__ void agg_table_init_kernel(AggSlot *t, int64_t n) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) {
    t[i].key = AGG_EMPTY_KEY;
    t[i].first_row = ~0ull;
  }
}

// 1. row -> slot, first_row min.  row id = row_ids ? row_ids[r] : offset + r
__global__ __launch_bounds__(BLOCK) void agg_resolve_kernel(
    const uint64_t *__restrict__ keys, const uint64_t *__restrict__ validity, int64_t n,
    const uint64_t *__restrict__ row_ids, uint64_t offset, AggSlot *table, uint64_t mask,
    uint32_t *__restrict__ row_slot, unsigned long long *new_count, uint64_t max_new, int *overflow) {
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= n) return;
  if (__hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  const uint64_t cap = mask + 1;
  uint64_t key = keys[r];
  uint64_t s;
  bool special = false, created = false;
  if (validity && !((validity[r >> 6] >> (r & 63)) & 1)) {
    s = cap; // NULL keys form one group (hash_utils.rs:91-104, aggregation.slt:21-26)
    special = true;
  } else if (key == AGG_EMPTY_KEY) {
    s = cap + 1;
    special = true;
  } else {
    s = mix64(key) & mask;
    uint64_t probes = 0;
    while (true) {
      unsigned long long cur = __hip_atomic_load(&table[s].key, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
      if (cur == key) break;
      if (cur == AGG_EMPTY_KEY) {
        unsigned long long prev = atomicCAS(&table[s].key, AGG_EMPTY_KEY, (unsigned long long)key);
        if (prev == AGG_EMPTY_KEY) { // this thread created the group
          created = true;
          break;
        }
        if (prev == key) break;
      }
      s = (s + 1) & mask;
      if (++probes > cap) {
        __hip_atomic_store(overflow, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
    }
  }
  (void)special;
  { // new groups are counted once per wave: one atomic per new group on this single address ran at
    // 0.65 G/s (1e7 mostly distinct keys: 15 ms in this kernel)
    const uint64_t cm = __ballot(created);
    if (cm && lane_id() == __builtin_ctzll(cm)) {
      const unsigned long long cnt = (unsigned long long)__popcll(cm);
      unsigned long long c = atomicAdd(new_count, cnt);
      if (c + cnt > max_new) __hip_atomic_store(overflow, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  uint64_t rid = row_ids ? row_ids[r] : offset + (uint64_t)r;
  // first_row only ever decreases, so a stale larger value just costs one extra atomic
  unsigned long long cur = __hip_atomic_load(&table[s].first_row, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
  if (rid < cur) atomicMin(&table[s].first_row, (unsigned long long)rid);
  row_slot[r] = (uint32_t)s;
}

// 2. bit r set  <=>  row r is the first row of its group (so the group is new in this batch)
__global__ __launch_bounds__(BLOCK) void agg_mark_first_kernel(
    const uint32_t *__restrict__ row_slot, int64_t n, const uint64_t *__restrict__ row_ids,
    uint64_t offset, const AggSlot *__restrict__ table, const uint32_t *__restrict__ slot_gid,
    uint64_t *__restrict__ bits) {
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  bool first = false;
  if (r < n) {
    uint32_t s = row_slot[r];
    uint64_t rid = row_ids ? row_ids[r] : offset + (uint64_t)r;
    first = table[s].first_row == rid && slot_gid[s] == 0xffffffffu;
  }
  uint64_t m = __ballot(first);
  if (lane_id() == 0 && r < n) bits[r >> 6] = m;
}

// 3. dense ids for the new groups (new_rows ascending => ids in first-seen order)
__global__ void agg_assign_gid_kernel(const uint32_t *__restrict__ new_rows, int64_t nnew,
                                      const uint32_t *__restrict__ row_slot,
                                      const uint64_t *__restrict__ row_ids, uint64_t offset,
                                      uint32_t base, uint32_t *__restrict__ slot_gid,
                                      uint64_t *__restrict__ gfirst) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= nnew) return;
  uint32_t r = new_rows[i];
  slot_gid[row_slot[r]] = base + (uint32_t)i;
  gfirst[base + i] = row_ids ? row_ids[r] : offset + r;
}

__global__ void agg_row_gid_kernel(const uint32_t *__restrict__ row_slot,
                                   const uint32_t *__restrict__ slot_gid, int64_t n,
                                   uint32_t *__restrict__ row_gid) {
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r < n) row_gid[r] = slot_gid[row_slot[r]];
}

// 4. accumulate.  `weights` (optional) = pre-aggregated non-null counts per input row.
__global__ __launch_bounds__(BLOCK) void agg_count_kernel(const uint32_t *__restrict__ row_gid,
                                                          const uint64_t *__restrict__ validity,
                                                          const int64_t *__restrict__ weights,
                                                          int64_t n, unsigned long long *nn) {
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= n) return;
  if (validity && !((validity[r >> 6] >> (r & 63)) & 1)) return;
  unsigned long long w = weights ? (unsigned long long)weights[r] : 1ull;
  if (w) atomicAdd(&nn[row_gid[r]], w);
}
__global__ __launch_bounds__(BLOCK) void agg_sum_i64_kernel(const uint32_t *__restrict__ row_gid,
                                                            const int64_t *__restrict__ vals,
                                                            const uint64_t *__restrict__ validity,
                                                            int64_t n, unsigned long long *sum) {
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= n) return;
  if (validity && !((validity[r >> 6] >> (r & 63)) & 1)) return;
  atomicAdd(&sum[row_gid[r]], (unsigned long long)vals[r]); // wrapping, like arrow's sum
}
__global__ __launch_bounds__(BLOCK) void agg_sum_f64_kernel(const uint32_t *__restrict__ row_gid,
                                                            const double *__restrict__ vals,
                                                            const uint64_t *__restrict__ validity,
                                                            int64_t n, double *sum) {
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= n) return;
  if (validity && !((validity[r >> 6] >> (r & 63)) & 1)) return;
  unsafeAtomicAdd(&sum[row_gid[r]], vals[r]);
}
// MIN/MAX on order-preserving u64 images; kind: 0 = i64, 1 = f64, 2 = i32
template <int KIND, bool IS_MIN>
__global__ __launch_bounds__(BLOCK) void agg_minmax_kernel(const uint32_t *__restrict__ row_gid,
                                                           const void *__restrict__ vals,
                                                           const uint64_t *__restrict__ validity,
                                                           int64_t n, unsigned long long *acc) {
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= n) return;
  if (validity && !((validity[r >> 6] >> (r & 63)) & 1)) return;
  uint64_t u;
  if (KIND == 0) u = i64_to_ordered(((const int64_t *)vals)[r]);
  else if (KIND == 1) u = f64_to_ordered(((const double *)vals)[r]);
  else if (KIND == 2) u = i64_to_ordered((int64_t)((const int32_t *)vals)[r]);
  else u = ((const uint64_t *)vals)[r]; // already an order-preserving image (partial MIN/MAX)
  unsigned long long *p = &acc[row_gid[r]];
  unsigned long long cur = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (IS_MIN) {
    if (u < cur) atomicMin(p, (unsigned long long)u);
  } else {
    if (u > cur) atomicMax(p, (unsigned long long)u);
  }
}

// finalize helpers
template <int KIND>
__global__ void agg_unorder_kernel(const uint64_t *__restrict__ acc, int64_t n, void *__restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (KIND == 0) ((int64_t *)out)[i] = ordered_to_i64(acc[i]);
  else if (KIND == 1) ((double *)out)[i] = ordered_to_f64(acc[i]);
  else ((int32_t *)out)[i] = (int32_t)ordered_to_i64(acc[i]);
}
__global__ void agg_nonzero_bits_kernel(const uint64_t *__restrict__ nn, int64_t n,
                                        uint64_t *__restrict__ bits) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  bool b = (i < n) && nn[i] != 0;
  uint64_t m = __ballot(b);
  if (lane_id() == 0 && i < n) bits[i >> 6] = m;
}
__global__ void fill_u64_kernel(uint64_t *p, int64_t n, uint64_t v) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void fill_u32_kernel(uint32_t *p, int64_t n, uint32_t v) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// re-insert every occupied slot of the old table into the (empty) new one
__global__ void agg_rehash_kernel(const AggSlot *__restrict__ old_t, const uint32_t *__restrict__ old_gid,
                                  uint64_t old_cap, AggSlot *new_t, uint32_t *new_gid,
                                  uint64_t new_mask) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= (int64_t)old_cap + 2) return;
  AggSlot s = old_t[i];
  uint64_t d;
  if ((uint64_t)i >= old_cap) {
    d = new_mask + 1 + ((uint64_t)i - old_cap); // the two reserved slots keep their roles
    if (s.first_row == ~0ull) return;
  } else {
    if (s.key == AGG_EMPTY_KEY) return;
    d = mix64(s.key) & new_mask;
    while (true) {
      unsigned long long prev = atomicCAS(&new_t[d].key, AGG_EMPTY_KEY, s.key);
      if (prev == AGG_EMPTY_KEY) break;
      d = (d + 1) & new_mask;
    }
  }
  new_t[d].first_row = s.first_row;
  new_gid[d] = old_gid[i];
}

// gfirst[gid] = first_row of the group's slot (the slot keeps the running minimum; gfirst is
// written when the group is created and goes stale if an earlier row arrives in a later call)
__global__ void agg_refresh_gfirst_kernel(const AggSlot *__restrict__ t, const uint32_t *__restrict__ slot_gid,
                                          int64_t nslots, uint64_t *__restrict__ gfirst) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= nslots) return;
  uint32_t g = slot_gid[i];
  if (g != 0xffffffffu) gfirst[g] = t[i].first_row;
}
void agg_refresh_gfirst(Ctx *ctx, AggState &st) {
  if (!st.table || st.ngroups == 0) return;
  int64_t nslots = (int64_t)st.mask + 3;
  agg_refresh_gfirst_kernel<<<dim3((unsigned)ceil_div(nslots, 256)), dim3(256), 0, ctx->stream>>>(
      st.table->as<AggSlot>(), st.slot_gid->as<uint32_t>(), nslots, st.gfirst.buf->as<uint64_t>());
  SQ_HIP(hipGetLastError());
}

void fill_u64(Ctx *ctx, uint64_t *p, int64_t n, uint64_t v) {
  if (n <= 0) return;
  fill_u64_kernel<<<dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream>>>(p, n, v);
  SQ_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ GrowBuf --
void GrowBuf::ensure(Ctx *ctx, int64_t n, uint64_t init) {
  if (n <= capacity) return;
  int64_t ncap = std::max<int64_t>(capacity * 2, std::max<int64_t>(n, 1024));
  BufP nb = ctx->alloc(8 * (size_t)ncap);
  if (capacity)
    SQ_HIP(hipMemcpyAsync(nb->p, buf->p, 8 * (size_t)capacity, hipMemcpyDeviceToDevice, ctx->stream));
  fill_u64_kernel<<<dim3((unsigned)ceil_div(ncap - capacity, 256)), dim3(256), 0, ctx->stream>>>(
      nb->as<uint64_t>() + capacity, ncap - capacity, init);
  SQ_HIP(hipGetLastError());
  buf = nb;
  capacity = ncap;
}

// ------------------------------------------------------------------- AggState --
void agg_table_alloc(Ctx *ctx, AggState &st, uint64_t cap) {
  int64_t nslots = (int64_t)cap + 2;
  BufP t = ctx->alloc(sizeof(AggSlot) * (size_t)nslots);
  BufP g = ctx->alloc(4 * (size_t)nslots);
  agg_table_init_kernel<<<dim3((unsigned)ceil_div(nslots, 256)), dim3(256), 0, ctx->stream>>>(
      t->as<AggSlot>(), nslots);
  fill_u32_kernel<<<dim3((unsigned)ceil_div(nslots, 256)), dim3(256), 0, ctx->stream>>>(
      g->as<uint32_t>(), nslots, 0xffffffffu);
  SQ_HIP(hipGetLastError());
  if (st.table) { // grow: rehash
    ProfScope ps(ctx, "agg_rehash");
    int64_t old_slots = (int64_t)st.mask + 3;
    agg_rehash_kernel<<<dim3((unsigned)ceil_div(old_slots, 256)), dim3(256), 0, ctx->stream>>>(
        st.table->as<AggSlot>(), st.slot_gid->as<uint32_t>(), st.mask + 1, t->as<AggSlot>(),
        g->as<uint32_t>(), cap - 1);
    SQ_HIP(hipGetLastError());
  }
  st.table = t;
  st.slot_gid = g;
  st.mask = cap - 1;
}

// Resolves every row of a key stream to a dense group id (creating groups as needed) and
// returns row_gid[u32 n].  new_rows_out / nnew_out describe the groups created by this call.
BufP agg_resolve_rows(Ctx *ctx, AggState &st, const NKeys &k, const uint64_t *row_ids,
                      uint64_t offset, BufP *new_rows_out, int64_t *nnew_out) {
  int64_t n = k.rows;
  if (!st.table) {
    st.exact = k.exact;
    st.key_dtype = k.dtype;
    uint64_t cap = 1024;
    agg_table_alloc(ctx, st, cap);
  }
  if (k.exact != st.exact || (k.exact && k.dtype != st.key_dtype))
    fail(SQLRS_ERR_ARROW, "group key type changed between batches");
  BufP row_slot = ctx->alloc(4 * (size_t)std::max<int64_t>(n, 1));
  BufP ctr = ctx->alloc(16);
  dim3 g((unsigned)ceil_div(std::max<int64_t>(n, 1), BLOCK)), b(BLOCK);
  // Capacity policy: keep load <= 1/2.  Small batches reserve for "every row a new group";
  // big batches start optimistic and grow x4 on overflow (the resolve pass is idempotent:
  // keys already inserted are found again, first_row minima are unchanged).
  const uint64_t SMALL = 4ull << 20;
  while (true) {
    uint64_t cap = st.mask + 1;
    uint64_t want = (uint64_t)st.ngroups + std::min<uint64_t>((uint64_t)n, SMALL);
    if (cap < 2 * want) {
      uint64_t nc = cap;
      while (nc < 2 * want) nc <<= 1;
      agg_table_alloc(ctx, st, nc);
      cap = nc;
    }
    uint64_t max_new = cap / 2 > (uint64_t)st.occupied ? cap / 2 - (uint64_t)st.occupied : 0;
    SQ_HIP(hipMemsetAsync(ctr->p, 0, 16, ctx->stream));
    {
      ProfScope ps(ctx, "agg_resolve");
      if (n)
        agg_resolve_kernel<<<g, b, 0, ctx->stream>>>(
            k.keys->as<uint64_t>(), k.validity, n, row_ids, offset, st.table->as<AggSlot>(), st.mask,
            row_slot->as<uint32_t>(), ctr->as<unsigned long long>(), max_new,
            (int *)(ctr->as<uint64_t>() + 1));
      SQ_HIP(hipGetLastError());
    }
    const uint64_t *h = (const uint64_t *)ctx->fetch(ctr->p, 16);
    uint64_t created = h[0];
    int overflow = (int)h[1];
    st.occupied += (int64_t)std::min<uint64_t>(created, cap); // slots claimed (even if we retry)
    if (!overflow) break;
    agg_table_alloc(ctx, st, cap * 4);
    // occupancy of the rebuilt table = what the rehash copied
    // (claimed slots stay claimed; they are re-found by the retry)
  }
  // new groups of this call, in row order
  Selection sel;
  sel.rows = n;
  int64_t nwords = ceil_div(std::max<int64_t>(n, 1), 64);
  sel.own_bits = ctx->alloc(8 * (size_t)nwords);
  sel.bits = sel.own_bits->as<uint64_t>();
  if (n) {
    ProfScope ps(ctx, "agg_mark_first");
    int64_t n64 = (int64_t)round_up((size_t)n, 64);
    agg_mark_first_kernel<<<dim3((unsigned)ceil_div(n64, BLOCK)), b, 0, ctx->stream>>>(
        row_slot->as<uint32_t>(), n, row_ids, offset, st.table->as<AggSlot>(),
        st.slot_gid->as<uint32_t>(), sel.own_bits->as<uint64_t>());
    SQ_HIP(hipGetLastError());
  }
  selection_finish(ctx, sel);
  int64_t nnew = sel.count;
  BufP new_rows = selection_indices_u32(ctx, sel);
  if ((int64_t)st.ngroups + nnew > 0xfffffff0ll) fail(SQLRS_ERR_INTERNAL, "more than 2^32 groups");
  st.gfirst.ensure(ctx, st.ngroups + nnew, ~0ull);
  if (nnew) {
    agg_assign_gid_kernel<<<dim3((unsigned)ceil_div(nnew, 256)), dim3(256), 0, ctx->stream>>>(
        new_rows->as<uint32_t>(), nnew, row_slot->as<uint32_t>(), row_ids, offset,
        (uint32_t)st.ngroups, st.slot_gid->as<uint32_t>(), st.gfirst.buf->as<uint64_t>());
    SQ_HIP(hipGetLastError());
  }
  BufP row_gid = ctx->alloc(4 * (size_t)std::max<int64_t>(n, 1));
  if (n) {
    agg_row_gid_kernel<<<g, b, 0, ctx->stream>>>(row_slot->as<uint32_t>(),
                                                 st.slot_gid->as<uint32_t>(), n,
                                                 row_gid->as<uint32_t>());
    SQ_HIP(hipGetLastError());
  }
  st.ngroups += nnew;
  if (new_rows_out) *new_rows_out = new_rows;
  if (nnew_out) *nnew_out = nnew;
  return row_gid;
}

void agg_update_count(Ctx *ctx, GrowBuf &nn, const uint32_t *row_gid, const uint64_t *validity,
                      const int64_t *weights, int64_t n) {
  if (!n) return;
  ProfScope ps(ctx, "agg_update");
  agg_count_kernel<<<dim3((unsigned)ceil_div(n, BLOCK)), dim3(BLOCK), 0, ctx->stream>>>(
      row_gid, validity, weights, n, nn.buf->as<unsigned long long>());
  SQ_HIP(hipGetLastError());
}
void agg_update_sum(Ctx *ctx, GrowBuf &acc, int32_t dtype, const uint32_t *row_gid, const void *vals,
                    const uint64_t *validity, int64_t n) {
  if (!n) return;
  ProfScope ps(ctx, "agg_update");
  dim3 g((unsigned)ceil_div(n, BLOCK)), b(BLOCK);
  if (dtype == SQLRS_FLOAT64)
    agg_sum_f64_kernel<<<g, b, 0, ctx->stream>>>(row_gid, (const double *)vals, validity, n,
                                                 acc.buf->as<double>());
  else
    agg_sum_i64_kernel<<<g, b, 0, ctx->stream>>>(row_gid, (const int64_t *)vals, validity, n,
                                                 acc.buf->as<unsigned long long>());
  SQ_HIP(hipGetLastError());
}
void agg_update_minmax(Ctx *ctx, GrowBuf &acc, int32_t dtype, bool is_min, const uint32_t *row_gid,
                       const void *vals, const uint64_t *validity, int64_t n) {
  if (!n) return;
  ProfScope ps(ctx, "agg_update");
  dim3 g((unsigned)ceil_div(n, BLOCK)), b(BLOCK);
  unsigned long long *a = acc.buf->as<unsigned long long>();
#define SQ_MM(K)                                                                                   \
  do {                                                                                             \
    if (is_min) agg_minmax_kernel<K, true><<<g, b, 0, ctx->stream>>>(row_gid, vals, validity, n, a); \
    else agg_minmax_kernel<K, false><<<g, b, 0, ctx->stream>>>(row_gid, vals, validity, n, a);     \
  } while (0)
  if (dtype == SQLRS_INT64) SQ_MM(0);
  else if (dtype == SQLRS_FLOAT64) SQ_MM(1);
  else if (dtype == SQLRS_INT32) SQ_MM(2);
  else if (dtype == SQLRS_UINT64) SQ_MM(3);
  else fail(SQLRS_ERR_INTERNAL, "unsupported min/max type"); // min_max.rs:41
#undef SQ_MM
  SQ_HIP(hipGetLastError());
}

DCol agg_finalize_values(Ctx *ctx, int func, int32_t dtype, GrowBuf &acc, GrowBuf *nn, int64_t G) {
  return agg_finalize_raw(ctx, func, dtype, acc.buf->as<uint64_t>(), nn ? nn->buf->as<uint64_t>() : nullptr, G);
}

DCol agg_finalize_raw(Ctx *ctx, int func, int32_t dtype, const uint64_t *acc, const uint64_t *nn, int64_t G,
                      const BufP &acc_owner) {
  DCol o;
  o.length = G;
  int64_t g1 = std::max<int64_t>(G, 1);
  dim3 g((unsigned)ceil_div(g1, 256)), b(256);
  const bool view = acc_owner && G > 0;
  if (func == SQLRS_AGG_COUNT) {
    o.dtype = SQLRS_INT64;
    if (view) {
      o.own_values = buf_view(acc_owner, acc, 8 * (size_t)G);
    } else {
      o.own_values = ctx->alloc(8 * (size_t)g1);
      if (G) SQ_HIP(hipMemcpyAsync(o.own_values->p, acc, 8 * (size_t)G, hipMemcpyDeviceToDevice, ctx->stream));
    }
    o.values = o.own_values->p;
    return o;
  }
  o.dtype = dtype;
  if (func == SQLRS_AGG_SUM) {
    if (view) {
      o.own_values = buf_view(acc_owner, acc, 8 * (size_t)G);
    } else {
      o.own_values = ctx->alloc(8 * (size_t)g1);
      if (G) SQ_HIP(hipMemcpyAsync(o.own_values->p, acc, 8 * (size_t)G, hipMemcpyDeviceToDevice, ctx->stream));
    }
  } else {
    o.own_values = ctx->alloc(8 * (size_t)g1);
    if (G) {
      if (dtype == SQLRS_INT64) agg_unorder_kernel<0><<<g, b, 0, ctx->stream>>>(acc, G, o.own_values->p);
      else if (dtype == SQLRS_FLOAT64) agg_unorder_kernel<1><<<g, b, 0, ctx->stream>>>(acc, G, o.own_values->p);
      else agg_unorder_kernel<2><<<g, b, 0, ctx->stream>>>(acc, G, o.own_values->p);
      SQ_HIP(hipGetLastError());
    }
  }
  o.values = o.own_values->p;
  if (nn) { // a group without any non-NULL input evaluates to NULL (sum.rs:25-34, min_max.rs:75-84)
    o.own_validity = ctx->alloc(bitmap_bytes(g1));
    int64_t g64 = (int64_t)round_up((size_t)g1, 64);
    agg_nonzero_bits_kernel<<<dim3((unsigned)ceil_div(g64, 256)), b, 0, ctx->stream>>>(
        nn, G, o.own_validity->as<uint64_t>());
    SQ_HIP(hipGetLastError());
    o.validity = o.own_validity->as<uint64_t>();
    o.null_count = -1;
  }
  return o;
}

} // namespace sq
