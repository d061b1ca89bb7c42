// prims.hpp — host-callable device primitives shared by the operators.
#pragma once

#include <functional>

#include "common.hpp"

namespace sq {

// ---- scans (scan.hip) --------------------------------------------------------------
// Exclusive prefix sum of n u32 values (single pass, decoupled look-back).  Writes u64
// and/or u32 outputs (either may be null) and the grand total to *total (device u64).
void exclusive_scan_u32(Ctx *ctx, const uint32_t *in, int64_t n, uint64_t *out64, uint32_t *out32,
                        uint64_t *total);

// ---- selections / compaction (select.hip) ------------------------------------------
// A row selection over `rows` rows: bit i of `bits` set = keep row i.  tile_off[t] is the
// number of kept rows before tile t (TILE_ROWS rows per tile); `count` kept rows total.
struct Selection {
  int64_t rows = 0;
  int64_t count = 0;
  const uint64_t *bits = nullptr;
  BufP own_bits;
  BufP tile_off; // u64[num_tiles]
};
// keep rows whose BOOLEAN mask is valid AND true (arrow filter semantics, filter.rs:23)
Selection selection_from_mask(Ctx *ctx, const DCol &mask);
// keep rows whose bit is CLEAR in `bits` (unvisited rows, hash_join.rs:298-301)
Selection selection_from_clear_bits(Ctx *ctx, const uint64_t *bits, int64_t rows);
// keep rows whose bit is SET in `bits` (a copy with the padding bits behind row `rows` cleared: Arrow leaves them unspecified)
Selection selection_from_set_bits(Ctx *ctx, const uint64_t *bits, int64_t rows);
// finishes a Selection whose bits are given (computes tile offsets + count; syncs)
void selection_finish(Ctx *ctx, Selection &s);
DCol compact_column(Ctx *ctx, const DCol &c, const Selection &s);
// u32 row ids of the kept rows
BufP selection_indices_u32(Ctx *ctx, const Selection &s);
BufP selection_indices_u64(Ctx *ctx, const Selection &s);

// Fused C2 fast path: predicate `col OP constant` + order-preserving compaction of that
// column in one pass.  Returns the Selection (bits + tile offsets) so that further columns
// can be compacted, and the compacted predicate column itself.
bool filter_fast_path(Ctx *ctx, const Expr &e, const std::function<const DCol &(int)> &col,
                      int64_t rows, int *col_index, Selection *sel, DCol *out_col);

// ---- gather (gather.hip) ------------------------------------------------------------
// out[i] = src[idx[i]]; NULL index (idx_validity bit clear) or NULL source row => NULL.
DCol gather_column(Ctx *ctx, const DCol &src, const void *idx, bool idx_is_u64,
                   const uint64_t *idx_validity, int64_t n);
// 2..4 plain 8-byte columns by one u32 permutation, through packed rows (gather.hip); false = shape not taken
bool gather_columns_packed(Ctx *ctx, std::vector<DCol> &cols, int64_t src_rows, const uint32_t *idx, int64_t n);
DCol concat_columns(Ctx *ctx, const std::vector<const DCol *> &parts); // one part: returned as is (no copy)
DCol copy_column(Ctx *ctx, const DCol &c); // deep copy into buffers owned by the result
// ops.hip: stable sort of the rows in `perm` by a Utf8 column (NULL rows tie); keys = n-element scratch
void sort_perm_by_utf8(Ctx *ctx, const DCol &c, const uint64_t *valid, int desc, uint32_t *perm, uint64_t *keys,
                       int64_t n);
DCol materialize_scalar(Ctx *ctx, const DCol &c, int64_t rows);
// number of clear validity bits among the first `rows` rows
int64_t count_clear_bits(Ctx *ctx, const uint64_t *bits, int64_t rows);

// ---- expressions (expr.hip) ---------------------------------------------------------
// BoundExpr::eval_column (evaluator.rs:13-28).  The result may be a broadcast scalar
// (stride 0) when the expression is a bare constant; pass materialize=true to expand it.
DCol eval_expr(Ctx *ctx, const Expr &e, const std::function<const DCol &(int)> &col, int64_t rows,
               bool materialize);

// ---- key normalisation (keys.hip) ---------------------------------------------------
// Join / group keys as one u64 per row.  Single fixed-width key column: the value itself
// (sign-extended / bit pattern), compared exactly, NULL tracked in `validity`.  Utf8 or
// multi-column keys: a 64-bit hash folded exactly like create_hashes/combine_hashes
// (hash_utils.rs:13-16,161-220: NULL leaves the running hash unchanged) and matched by
// hash only, which is the reference's own matching rule (hash_join.rs:222-232).
struct NKeys {
  int64_t rows = 0;
  BufP keys;                          // u64[rows]
  const uint64_t *validity = nullptr; // exact mode only: NULL key rows
  BufP own_validity;
  bool exact = true;
  int32_t dtype = SQLRS_INT64;        // exact mode: dtype of the single key column
};
NKeys normalize_keys(Ctx *ctx, const std::vector<DCol> &cols, int64_t rows);
// library-internal de-duplication keys: asymmetric strong hash, NULL is a value (keys.hip)
NKeys normalize_keys_strong(Ctx *ctx, const std::vector<DCol> &cols, int64_t rows);

// ---- radix sort (sort.hip) ----------------------------------------------------------
// Stable LSD radix sort of (u64 key, u32 value) pairs on bits [begin_bit, end_bit).
// Results are left in keys/vals (temporaries come from the pool).  `keys_below_end_bit`: the caller guarantees that no
// key has a bit at or above end_bit set (row numbers below a known count) — the question "on which bytes do the keys
// differ" (a pass over the keys + a host round trip) is then not asked: every byte of [begin_bit, end_bit) is sorted.
void radix_sort_pairs(Ctx *ctx, uint64_t *keys, uint32_t *vals, int64_t n, int begin_bit,
                      int end_bit, bool keys_below_end_bit = false);
// perm[0 .. n) = the indices 0 .. n-1 in the stable order of their u32 keys (all below 2^bits)
void radix_sort_index_u32(Ctx *ctx, const uint32_t *keys, int64_t n, int bits, uint32_t *perm);
void iota_u32(Ctx *ctx, uint32_t *out, int64_t n);
// order_fast.hip: ORDER BY one fixed-width key without NULLs, rows (key, one carried 8-byte column, row id)
// travel through <= 2 HBM passes + an in-LDS finish; false = shape / data do not fit (general path)
bool order_fast(Ctx *ctx, const DCol &key, int desc, const DCol *carry, int64_t n, DCol *key_out, DCol *carry_out,
                BufP *perm, bool want_perm, bool *in_order = nullptr);
// 2..4 plain int64 / int32 key columns, or 1..4 when one has NULLs (NULLs first: a valid bit in front of the value), whose
// fields together fit 64 bits: ordered by one composite key through order_fast; keys_out = the key columns in output order
// (values and validity decoded from the sorted composite).  carry_out stays empty when the
// row ids travelled instead (want_perm with wide composites): the caller gathers that column like the others
bool order_composite(Ctx *ctx, const std::vector<const DCol *> &keys, const std::vector<int> &desc, const DCol *carry, int64_t n,
                     std::vector<DCol> *keys_out, DCol *carry_out, BufP *perm, bool want_perm, bool *in_order);

} // namespace sq
