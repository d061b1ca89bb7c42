// bench_host_batches.cpp — HashAgg fed the reference's batch shape by a NATIVE caller: 19 532 pageable host batches
// of 1024 rows (storage/csv.rs:105) through sqlrs_hash_agg_push, result on the host.  What a Rust drop-in pays per
// batch — a C call, no interpreter — next to bench.py's `C4_host_batches_1024`, whose wall clock includes one Python
// ctypes call per batch.  Prints one JSON object; `bench.py` adds it to the line as `C4_host_batches_1024_native`.
//   ./bench_host_batches [rows = 2e7] [groups = 1e6] [batch = 1024]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sqlrs_hip.h"

static uint64_t splitmix64(uint64_t seed, uint64_t i) { // the generator of sqlrs_amd/datagen.py
  uint64_t z = seed * 0x9E3779B97F4A7C15ull + (i + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
#define CHECK(x)                                                                                  \
  do {                                                                                            \
    int st_ = (x);                                                                                \
    if (st_ != SQLRS_OK) {                                                                        \
      std::fprintf(stderr, "%s failed (%d): %s\n", #x, st_, ctx ? sqlrs_last_error(ctx) : "");    \
      return 1;                                                                                   \
    }                                                                                             \
  } while (0)

int main(int argc, char **argv) {
  const int64_t n = argc > 1 ? (int64_t)std::atof(argv[1]) : 20000000, G = argc > 2 ? (int64_t)std::atof(argv[2]) : 1000000;
  const int64_t B = argc > 3 ? std::atoll(argv[3]) : 1024;
  sqlrs_ctx_t *ctx = nullptr;
  if (sqlrs_ctx_create(0, &ctx) != SQLRS_OK) {
    std::printf("{\"error\": \"no device\"}\n");
    return 2;
  }
  std::vector<int64_t> key((size_t)n);
  std::vector<double> val((size_t)n);
  for (int64_t i = 0; i < n; i++) {
    key[(size_t)i] = (int64_t)(splitmix64(0xA1, (uint64_t)i) % (uint64_t)G);
    val[(size_t)i] = (double)(splitmix64(0xF2, (uint64_t)i) >> 11) * (1.0 / 9007199254740992.0);
  }
  sqlrs_expr_node_t k0{}, v1{};
  k0.op = SQLRS_EXPR_INPUT_REF;
  k0.index = 0;
  v1.op = SQLRS_EXPR_INPUT_REF;
  v1.index = 1;
  sqlrs_expr_t gb{&k0, 1, 0};
  sqlrs_agg_func_t aggs[2] = {};
  aggs[0].func = SQLRS_AGG_COUNT;
  aggs[0].return_dtype = SQLRS_INT64;
  aggs[0].arg = sqlrs_expr_t{&v1, 1, 0};
  aggs[1].func = SQLRS_AGG_SUM;
  aggs[1].return_dtype = SQLRS_FLOAT64;
  aggs[1].arg = sqlrs_expr_t{&v1, 1, 0};
  double best = 1e30, sum_check = 0;
  int64_t groups = 0, count_check = 0, batches = 0;
  for (int rep = 0; rep < 3; rep++) { // (the first repetition warms the pool and the staging area)
    auto t0 = std::chrono::steady_clock::now();
    sqlrs_hash_agg_t *a = nullptr;
    CHECK(sqlrs_hash_agg_create(ctx, 1, &gb, 2, aggs, &a));
    batches = 0;
    for (int64_t lo = 0; lo < n; lo += B) {
      const int64_t m = std::min<int64_t>(B, n - lo);
      sqlrs_column_t cols[2] = {};
      cols[0].dtype = SQLRS_INT64;
      cols[0].mem = SQLRS_MEM_HOST;
      cols[0].length = m;
      cols[0].values = key.data() + lo;
      cols[1].dtype = SQLRS_FLOAT64;
      cols[1].mem = SQLRS_MEM_HOST;
      cols[1].length = m;
      cols[1].values = val.data() + lo;
      sqlrs_batch_t b{};
      b.num_rows = m;
      b.num_columns = 2;
      b.columns = cols;
      CHECK(sqlrs_hash_agg_push(a, &b));
      batches++;
    }
    sqlrs_batch_t *out = nullptr;
    CHECK(sqlrs_hash_agg_finish(a, SQLRS_MEM_HOST, &out));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    groups = out->num_rows;
    count_check = 0;
    sum_check = 0;
    const int64_t *cnt = (const int64_t *)out->columns[1].values;
    const double *sm = (const double *)out->columns[2].values;
    for (int64_t g = 0; g < groups; g++) {
      count_check += cnt[g];
      sum_check += sm[g];
    }
    sqlrs_batch_release(out);
    sqlrs_hash_agg_destroy(a);
    if (rep > 0 && ms < best) best = ms;
  }
  double exp_sum = 0;
  for (int64_t i = 0; i < n; i++) exp_sum += val[(size_t)i];
  const bool ok = count_check == n && std::abs(sum_check - exp_sum) <= 1e-9 * exp_sum;
  std::printf("{\"rows\": %lld, \"batches\": %lld, \"batch_rows\": %lld, \"groups\": %lld, \"ms\": %.1f, \"Mrows_s\": %.1f, "
              "\"pcie_GBps\": %.2f, \"check\": \"%s\", \"note\": \"native caller (C ABI, no interpreter): pageable %lld-row host "
              "batches through sqlrs_hash_agg_push's host staging, result on the host, best of 2 after a warm-up\"}\n",
              (long long)n, (long long)batches, (long long)B, (long long)groups, best, (double)n / best / 1e3,
              16.0 * (double)n / best / 1e6, ok ? "OK" : "mismatch", (long long)B);
  sqlrs_ctx_destroy(ctx);
  return ok ? 0 : 1;
}
