// radix_part.hpp — interface of the LDS-staged hash partitioner (radix_part.hip).
#pragma once

#include "common.hpp"
#include "prims.hpp"

namespace sq {

struct PartitionInput {
  const uint64_t *keys = nullptr; // normalised keys (NKeys)
  const uint64_t *key_validity = nullptr;
  int64_t n = 0;
  int nv = 0; // 8-byte value columns carried along (0..2)
  const void *vals[2] = {nullptr, nullptr};
  const uint64_t *val_validity[2] = {nullptr, nullptr};
  bool build_side = false; // profiling label only
};

// Rows in bucket order: bucket b = rows [bstart[b], bstart[b+1]).  `idx` = original row,
// `flags` (null when nothing is nullable) bit0 key valid, bit1 v0 valid, bit2 v1 valid.
// bucket(key) = mulhi(mix64(key), P); NULL keys -> bucket 0.
struct PartitionedRows {
  int64_t n = 0;
  uint32_t P = 0;
  BufP key, v0, v1, idx, flags;
  BufP bstart; // u32[P + 1]
};

// P_wanted <= 65536; the actual bucket count (>= P_wanted) is returned in out->P.
bool partition_rows(Ctx *ctx, const PartitionInput &in, uint32_t P_wanted, PartitionedRows *out);

} // namespace sq
