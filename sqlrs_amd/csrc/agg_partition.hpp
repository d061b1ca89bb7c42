// agg_partition.hpp — interface of the LDS-partitioned pre-aggregation (agg_partition.hip).
#pragma once

#include <cmath>

#include "common.hpp"
#include "prims.hpp"
#include "radix_part.hpp"

namespace sq {

constexpr int PART_MAX_ACC = 6;
enum PartOp { PART_COUNT = 0, PART_SUM_I64 = 1, PART_SUM_F64 = 2, PART_MIN = 3, PART_MAX = 4 };

struct PartAggSpec {
  int n_acc = 0;
  int op[PART_MAX_ACC] = {0};
  int src[PART_MAX_ACC] = {0};  // index into PartAggInput::vals
  int kind[PART_MAX_ACC] = {0}; // MIN/MAX element type: 0 = i64, 1 = f64
  int nv = 0;                   // value columns used (0..2), all 8 bytes wide
};

struct PartAggInput {
  const uint64_t *keys = nullptr; // normalised keys (NKeys)
  const uint64_t *key_validity = nullptr;
  int64_t n = 0;
  const void *vals[2] = {nullptr, nullptr};
  const uint64_t *val_validity[2] = {nullptr, nullptr};
  // fused inner join (optional): normalised build keys; rows whose key has no build partner
  // are dropped, groups exist only for keys that have one
  const uint64_t *join_keys = nullptr;
  const uint64_t *join_validity = nullptr;
  int64_t join_n = 0;
  struct PartitionedRows *join_cache = nullptr; // build side in bucket order, reused across batches
  // smallest / largest valid build key (signed-order images) when the caller already knows them
  // (the join's direct-address table): saves a pass over the build keys and a host round trip
  bool join_range_known = false;
  uint64_t join_omin = 0, join_omax = 0;
  // the build keys are KNOWN to be unique (the caller's pre-condition).  false: not established yet — only the route
  // that inserts the build keys into its bucket tables (and notices a key twice) may run, not the direct-addressed
  // one, which takes "every key of the range has a partner" from uniqueness
  bool join_unique_known = true;
  // existence bitmap of the build keys over [join_omin, join_omax] (bit = key offset; the join's direct-address
  // table, join_state.hpp): lets the direct-addressed route run when the unique build keys do NOT cover their whole
  // range — a slot is a group only where the bit is set.  Requires join_range_known and join_unique_known.
  const uint64_t *join_bits = nullptr;
  // DUPLICATE build keys over [join_omin, join_omax] (no NULL key): build rows per key offset (u32, 0 = none).  Every
  // probe row then stands for that many joined rows: the direct-addressed route aggregates the probe rows as usual and
  // multiplies COUNT / SUM cells by the key's multiplicity when a slot is emitted (MIN / MAX are unaffected; the
  // first-seen order is the probe rows').  Only that route: the call returns false when it does not apply.
  const uint32_t *join_mult = nullptr;
  // FilterExecutor directly below (fused join only): rows failing `filter` do not exist for the
  // operator.  Evaluated by the chunked first partition level; when that level does not apply the
  // call returns false and the caller runs the Filter operator first.
  RowFilter filter;
  // true: key statistics from a full pass (the rerun after an optimistic attempt saw a key outside its sampled range)
  bool exact_stats = false;
};

// Groups of ONE batch: key, first row (local index), one 8-byte cell per accumulator
// (COUNT: count, SUM: partial sum, MIN/MAX: order-preserving u64 image).
struct PartAggOutput {
  int64_t groups = 0, gcap = 0;
  BufP gkey, gfirst, gvalid, gvalid_bits, gacc;
  BufP row_ids;            // gfirst as global row numbers (u64): made on demand, part_row_ids()
  uint64_t row_offset = 0; // global row number of the batch's row 0
  int64_t n_overflow = 0; // rows that did not fit their bucket table
  BufP ov_rows;           // their local row ids (u32)
  bool may_dup = false;   // a skewed bucket was split: the same key can appear more than once
  double est_groups = 0;
  int buckets = 0;
  // partitioned_preaggregate returned false because a key lay outside the SAMPLED key range the rows were packed
  // with: call again with PartAggInput::exact_stats (nothing of this attempt is kept)
  bool retry_exact = false;
};

// first rows of the groups as global row numbers (row_offset + gfirst), made on first use: only a merge into an existing
// aggregation state needs them; a single batch's groups are ordered by the local u32 first rows as they are
const uint64_t *part_row_ids(Ctx *ctx, PartAggOutput &po);
// also returns min / max of the valid keys' signed-order image (key ^ 1 << 63) when asked
double estimate_distinct(Ctx *ctx, const uint64_t *keys, const uint64_t *validity, int64_t n,
                         uint64_t *omin = nullptr, uint64_t *omax = nullptr, bool *sampled = nullptr);
// false = not applicable (too many groups for one partition level, or estimate blown):
// the caller uses the resolve path for the whole batch.
bool partitioned_preaggregate(Ctx *ctx, const PartAggSpec &spec, const PartAggInput &in,
                              uint64_t row_offset, PartAggOutput *out);

} // namespace sq
