// agg_state.hpp — shared state of the hash aggregation (agg.hip kernels, ops.hip operator,
// agg_partition.hip pre-aggregation).
#pragma once

#include "common.hpp"
#include "prims.hpp"

namespace sq {

constexpr uint64_t AGG_EMPTY_KEY = ~0ull;

struct AggSlot {
  unsigned long long key;
  unsigned long long first_row; // global index of the first row that carried this key
};

// device array of 8-byte elements that grows by doubling, new tail initialised to `init`
struct GrowBuf {
  BufP buf;
  int64_t capacity = 0;
  void ensure(Ctx *ctx, int64_t n, uint64_t init);
};

struct AggState {
  BufP table;        // AggSlot[cap + 2]; slot cap = NULL key, slot cap+1 = key == AGG_EMPTY_KEY
  BufP slot_gid;     // u32[cap + 2]: dense group id of a slot (0xffffffff = none yet)
  uint64_t mask = 0; // cap - 1
  int64_t ngroups = 0;
  int64_t occupied = 0;
  bool exact = true;
  int32_t key_dtype = SQLRS_INT64;
  GrowBuf gfirst; // u64[ngroups]: first row of each group
};

void fill_u64(Ctx *ctx, uint64_t *p, int64_t n, uint64_t v);
void agg_refresh_gfirst(Ctx *ctx, AggState &st);
void agg_table_alloc(Ctx *ctx, AggState &st, uint64_t cap);
BufP agg_resolve_rows(Ctx *ctx, AggState &st, const NKeys &k, const uint64_t *row_ids,
                      uint64_t offset, BufP *new_rows_out, int64_t *nnew_out);
void agg_update_count(Ctx *ctx, GrowBuf &nn, const uint32_t *row_gid, const uint64_t *validity,
                      const int64_t *weights, int64_t n);
void agg_update_sum(Ctx *ctx, GrowBuf &acc, int32_t dtype, const uint32_t *row_gid, const void *vals,
                    const uint64_t *validity, int64_t n);
void agg_update_minmax(Ctx *ctx, GrowBuf &acc, int32_t dtype, bool is_min, const uint32_t *row_gid,
                       const void *vals, const uint64_t *validity, int64_t n);
// `acc_owner` (optional): the buffer `acc` points into; COUNT / SUM columns are then views of it
// instead of copies (the cells already hold the final values)
DCol agg_finalize_raw(Ctx *ctx, int func, int32_t dtype, const uint64_t *acc, const uint64_t *nn, int64_t G,
                      const BufP &acc_owner = nullptr);
DCol agg_finalize_values(Ctx *ctx, int func, int32_t dtype, GrowBuf &acc, GrowBuf *nn, int64_t G);

} // namespace sq
