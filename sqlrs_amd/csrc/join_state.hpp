// join_state.hpp — state of one HashJoinExecutor (join.hip); shared with the fused
// join + aggregate operator in hashagg_op.hip.
#pragma once

#include "common.hpp"
#include "host_stage.hpp"
#include "prims.hpp"
#include "radix_part.hpp"

struct sqlrs_hash_join {
  sq::Ctx *ctx = nullptr;
  sq::HostStage hstage; // small HOST build batches until one upload (host_stage.hpp)
  std::unique_ptr<sq::HostStage> probe_stage; // sqlrs_hash_join_probe_push_many: the call's small HOST probe batches
  void *pin_out = nullptr;                    // ... and where their joined rows land on the host (pinned)
  size_t pin_cap = 0;
  ~sqlrs_hash_join() {
    if (pin_out) (void)hipHostFree(pin_out);
  }
  int join_type = 0;
  std::vector<sq::Expr> lkeys, rkeys;
  bool has_filter = false;
  sq::Expr filter;
  std::vector<int32_t> right_dtypes;
  // build
  std::vector<sq::DBatch> left_batches;
  std::vector<sq::NKeys> left_key_parts;
  // Several key columns (ON a.x = b.x AND a.y = b.y), OPT-IN (SQLRS_JOIN_COMPOSITE=1: exact equality instead of the
  // reference's match-by-hash, whose collisions are part of its results): the evaluated key columns of every build batch are
  // kept until the build is finished; when they are all integers without NULLs whose ranges multiply to < 2^62, both sides
  // are joined on ONE exact int64 key, sum_c (v_c - min_c) * stride_c (join.hip, composite_build_keys) — dense ranges then
  // take the direct-address table, the rest the LDS route
  std::vector<std::vector<sq::DCol>> left_keycol_parts;
  struct Composite {
    bool on = false;
    int nk = 0;
    int32_t dtype[4] = {0, 0, 0, 0};
    int64_t min[4] = {0, 0, 0, 0};
    uint64_t range[4] = {0, 0, 0, 0}, stride[4] = {0, 0, 0, 0};
  } comp;
  bool finished = false, empty_build = true;
  sq::DBatch left;
  int64_t nB = 0;
  sq::BufP table;
  uint64_t mask = 0;
  bool unique = true, exact = true;
  // A join owned by a HashJoin+HashAgg (sqlrs_join_agg) defers its general hash table: the fused route never
  // probes it (its bucket pass inserts the build keys into LDS tables and reports duplicates itself), so the table
  // (1.1 ms for 1e7 keys) is built on first need only — `unique` is a fact once `unique_known` is set.
  bool lazy_table = false, table_built = false, unique_known = false;
  int32_t key_dtype = SQLRS_INT64;
  sq::BufP rows_by_slot;
  sq::BufP visited; // bit per build row
  sq::BufP dense;              // direct-address table (u32 build row per key - dense_min) or null
  uint64_t dense_min = 0, dense_range = 0;
  uint32_t dense_null_head = 0xffffffffu;
  sq::BufP dense_packed; // bit-packed copy of `dense` for the probe kernels (join.hip, DenseTable) or null
  uint32_t dense_pbits = 0;
  bool probe_miss_seen = false; // a probe batch had a row without partner: no more optimistic all-hit attempts (join.hip)
  // The direct-address build WITHOUT its host round trip (join.hip, dense_resolve): the kernels of the attempt are queued,
  // their verdict (key range, occupied slots, NULL keys: `pend_st`) is still on the device.  The first probe either runs
  // its optimistic all-hit kernel against the device-side verdict and fetches both answers at once, or resolves first.
  bool dense_pending = false;
  sq::BufP pend_st, pend_dense, pend_packed;
  uint32_t pend_bits = 0;
  uint64_t pend_max_range = 0;
  bool pend_validity = false;
  sq::BufP dense_bits; // one bit per possible key of the direct-address table (key-only build side, join.hip)
  // duplicate build keys over a dense range (no NULL key): the range the direct-address build found, and — on first
  // need of the fused join+aggregate — how many build rows carry each key of it (u32 per key, 0 = none)
  uint64_t dup_min = 0, dup_range = 0;
  sq::BufP dup_mult;
  // round 6: duplicate build keys over a dense range WITHOUT the general table: dd_table[key - dup_min] = first entry of the
  // key's run in rows_by_slot, dd_table[.. + 1] = the end of the run (build_dense_dup, join.hip) — the count pass of a probe is
  // one L2-resident 8-byte lookup per row
  sq::BufP dd_table;
  sq::BufP bkeys, bkeys_validity; // normalised build keys (u64[nB]) and their validity bitmap
  // general keys on LDS tables (join.hip, lds_join_match): the build keys in bucket order, built at the first probe
  // that takes the route; lds_slots = 0: the route does not apply to this build side
  std::shared_ptr<sq::PartitionedRows> lds_build;
  uint32_t lds_slots = 0;
  // ... and of a build side with DUPLICATE keys, its DISTINCT keys (the non-empty slots of the general table) in bucket order:
  // "row" d of that partition indexes lds_dmatch[d] = {first entry of the key's run in rows_by_slot, rows of the run}
  std::shared_ptr<sq::PartitionedRows> lds_distinct;
  sq::BufP lds_dkeys, lds_dmatch;
  uint32_t lds_dslots = 0;
  // round 6: `unique` established on the LDS bucket tables (lds_build_first); the global table is not built until a
  // probe batch that cannot take the LDS route asks for it (`table_built` stays false until then)
  bool lds_first = false;
  bool async_ordered = false; // the async path's side streams have been ordered behind this join's build (small_async.hpp)
};

// builds the deferred hash table of a `lazy_table` join (join.hip); no-op otherwise
namespace sq {
void hash_join_ensure_table(sqlrs_hash_join *j);
// existence bitmap of a direct-address (`dense`) join: bit (key - dense_min) is set when the key has a build row
// (join.hip, dense_bits_kernel); built once, on first need
const uint64_t *hash_join_dense_bits(sqlrs_hash_join *j);
// build rows per key of [dup_min, dup_min + dup_range) of a join with duplicate build keys over a dense range
// (join.hip); null when the build side has no such range
const uint32_t *hash_join_dup_mult(sqlrs_hash_join *j);
// ---- general keys on LDS bucket tables (join_lds.hip) ----
// the probe rows of one batch in (range, bucket) order with the match of every row's key beside them — the build row, or an
// index into lds_dmatch when `distinct` (duplicate build keys), or 0xffffffff; `ok` = false: route not taken
struct LdsJoinMatch {
  bool ok = false;
  BufP idx, mpart; // u16[n]: the row's number inside its 2^15-row range; u32[n]: build row | DENSE_EMPTY
};
// a build side that will be probed on LDS tables establishes `unique` there and leaves the general table unbuilt; false: not such
// a build side, or its keys are not unique (the caller builds the general table)
bool lds_build_first(sqlrs_hash_join *j);
LdsJoinMatch lds_join_match(sqlrs_hash_join *j, const NKeys &pk, bool distinct = false);
int64_t lds_join_tiles(int64_t n); // look-back descriptors lds_join_restore needs for a batch of n rows
void lds_join_restore(Ctx *ctx, const LdsJoinMatch &lm, int64_t n, uint64_t *left_idx, uint32_t *right_idx, uint64_t *desc, unsigned *ticket,
                      uint64_t *total, int use_ticket);
void lds_join_unpermute(sqlrs_hash_join *j, const LdsJoinMatch &lm, int64_t n, int outer_right, uint2 *match, uint32_t *counts, int grouped);
}
