// join_kernels.hpp — what the translation units of the HashJoinExecutor share on the device side (join.hip: build, direct-address
// and general-table probes, the operator; join_lds.hip: general keys on LDS bucket tables): the general table's slot and the
// "no build row" word.
#pragma once

#include "common.hpp"
#include "device_utils.hpp"

namespace sq {

constexpr uint64_t EMPTY_KEY = ~0ull;

struct Slot {
  unsigned long long key;
  uint32_t head;  // unique: build row; otherwise start into rows_by_slot
  uint32_t count; // build rows with this key
};

#if defined(__HIPCC__)
__device__ __forceinline__ Slot load_slot(const Slot *p) {
  ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(p);
  Slot s;
  s.key = v.x;
  s.head = (uint32_t)v.y;
  s.count = (uint32_t)(v.y >> 32);
  return s;
}
#endif

constexpr uint32_t DENSE_EMPTY = 0xffffffffu; // "no build row" in the direct-address tables, the match arrays and the pair kernels

} // namespace sq
