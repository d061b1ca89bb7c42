// host_stage.hpp — small HOST batches pushed into a blocking operator (HashAgg, the fused join's probe
// side, Order, the join's build side) are appended to a host-side staging area and uploaded in ONE
// transfer per column when enough rows have arrived or the operator finishes.
//
// The reference feeds its operators 1024-row batches (storage/csv.rs:105).  Uploading every such batch on
// its own costs a copy + a stream synchronisation per column and call (~50-100 us: 10-20 Mrows/s), while
// the operators are blocking anyway and aggregate what they receive as one batch.  Staged pushes cost a
// memcpy; the upload happens per HOST_STAGE_FLUSH_ROWS rows.  Only fixed-width columns are staged (Utf8 /
// Boolean batches take the ordinary path); row order across pushes is preserved (a push that cannot be
// staged flushes the stage first).
#pragma once

#include "common.hpp"

namespace sq {

constexpr int64_t HOST_STAGE_FLUSH_ROWS = 1ll << 22;
constexpr int64_t HOST_STAGE_MAX_BATCH = 1ll << 20; // larger host batches are uploaded directly
constexpr int64_t HOST_STAGE_PIN_ROWS = 1ll << 16;  // staged rows from which the staging area is pinned memory

struct HostStage {
  Ctx *ctx = nullptr;
  // A column's staged values live in a std::vector while there are few of them (an operator that sees four rows must
  // not pay for a pinned allocation) and in a PINNED buffer from HOST_STAGE_PIN_ROWS rows on: the upload of a flush then
  // runs at the PCIe rate instead of the pageable-copy rate, and the buffer — kept across flushes — is touched once
  // (19 532 batches of 1024 rows into HashAgg from a native caller: 196 -> 41 ms per 2e7 rows; most of the 196 was the pageable
  // H2D copy and the first-touch page faults of a fresh 32 MB vector per column and flush).
  struct Col {
    int32_t dtype = 0;
    std::vector<uint8_t> vals;
    uint8_t *pin = nullptr; // hipHostMalloc'd; owns the values once non-null
    size_t pin_cap = 0, pin_size = 0;
    std::vector<uint64_t> valid; // bit per row, only maintained once a NULL has been seen
    int64_t nulls = 0;
  };
  std::vector<Col> cols;
  bool has_schema = false;
  int64_t rows = 0;
  HostStage() = default;
  HostStage(const HostStage &) = delete;
  HostStage &operator=(const HostStage &) = delete;
  ~HostStage();

  bool accepts(const sqlrs_batch_t *b) const;
  void append(const sqlrs_batch_t *b);
  // the staged rows as one library-owned DEVICE batch (nullptr when no batch is staged); empties the stage
  sqlrs_batch_t *take();
};

} // namespace sq
