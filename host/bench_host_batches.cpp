// bench_host_batches.cpp — HashAgg fed the reference's batch shape by a NATIVE caller: 19 532 pageable host batches
// of 1024 rows (storage/csv.rs:105) through sqlrs_hash_agg_push, result on the host.  What a Rust drop-in pays per
// batch — a C call, no interpreter — next to bench.py's `C4_host_batches_1024`, whose wall clock includes one Python
// ctypes call per batch.  Prints one JSON object; `bench.py` adds it to the line as `C4_host_batches_1024_native`.
//   ./bench_host_batches [rows = 2e7] [groups = 1e6] [batch = 1024]
//   ./bench_host_batches filter ... | probe ...   the streaming operators at the same batch shape (see bench_filter / bench_probe)
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

static void host_col(sqlrs_column_t &c, int32_t dtype, const void *values, int64_t m) {
  std::memset(&c, 0, sizeof(c));
  c.dtype = dtype;
  c.mem = SQLRS_MEM_HOST;
  c.length = m;
  c.values = values;
}

// ./bench_host_batches filter [rows = 2e7] [batch = 1024] [group = 1024]
// FilterExecutor (C2's query: SELECT v1 FROM t WHERE v1 > k, selectivity 0.5) fed pageable 1024-row host batches by a native
// caller, result batches on the host: (a) one sqlrs_filter_push per batch, (b) sqlrs_filter_push_many over groups of batches.
static int bench_filter(int argc, char **argv) {
  const int64_t n = argc > 2 ? (int64_t)std::atof(argv[2]) : 20000000, B = argc > 3 ? std::atoll(argv[3]) : 1024;
  const int group = argc > 4 ? std::atoi(argv[4]) : 1024;
  sqlrs_ctx_t *ctx = nullptr;
  if (sqlrs_ctx_create(0, &ctx) != SQLRS_OK) {
    std::printf("{\"error\": \"no device\"}\n");
    return 2;
  }
  std::vector<int64_t> v((size_t)n);
  int64_t expect = 0;
  const int64_t k = (1ll << 30);
  for (int64_t i = 0; i < n; i++) {
    v[(size_t)i] = (int64_t)(splitmix64(0xC2, (uint64_t)i) % (1ull << 31));
    expect += v[(size_t)i] > k;
  }
  sqlrs_expr_node_t nodes[3] = {};
  nodes[0].op = SQLRS_EXPR_INPUT_REF;
  nodes[0].index = 0;
  nodes[1].op = SQLRS_EXPR_CONSTANT;
  nodes[1].dtype = SQLRS_INT64;
  nodes[1].i = k;
  nodes[2].op = SQLRS_EXPR_GT;
  sqlrs_expr_t pred{nodes, 3, 0};
  const int64_t nb = (n + B - 1) / B;
  std::vector<sqlrs_column_t> cols((size_t)nb);
  std::vector<sqlrs_batch_t> batches((size_t)nb);
  std::vector<const sqlrs_batch_t *> ptrs((size_t)nb);
  for (int64_t b = 0; b < nb; b++) {
    const int64_t m = std::min<int64_t>(B, n - b * B);
    host_col(cols[(size_t)b], SQLRS_INT64, v.data() + b * B, m);
    std::memset(&batches[(size_t)b], 0, sizeof(sqlrs_batch_t));
    batches[(size_t)b].num_rows = m;
    batches[(size_t)b].num_columns = 1;
    batches[(size_t)b].columns = &cols[(size_t)b];
    ptrs[(size_t)b] = &batches[(size_t)b];
  }
  double best[3] = {1e30, 1e30, 1e30};
  int64_t kept[3] = {0, 0, 0};
  bool order_ok = true;
  const int DEPTH = 8; // tickets in flight of the async mode (the caller waits that many batches behind)
  for (int mode = 0; mode < 3; mode++) {
    for (int rep = 0; rep < 3; rep++) {
      auto t0 = std::chrono::steady_clock::now();
      sqlrs_filter_t *f = nullptr;
      CHECK(sqlrs_filter_create(ctx, &pred, &f));
      int64_t got = 0;
      std::vector<sqlrs_batch_t *> outs((size_t)group);
      if (mode == 0) {
        for (int64_t b = 0; b < nb; b++) {
          sqlrs_batch_t *o = nullptr;
          CHECK(sqlrs_filter_push(f, ptrs[(size_t)b], SQLRS_MEM_HOST, &o));
          got += o->num_rows;
          sqlrs_batch_release(o);
        }
      } else if (mode == 2) { // one batch per call, no synchronisation per call: sqlrs_filter_push_async + sqlrs_batch_wait
        std::vector<sqlrs_ticket_t *> q((size_t)DEPTH, nullptr);
        auto take = [&](int64_t b) { // the batch of input batch b
          sqlrs_batch_t *o = nullptr;
          CHECK(sqlrs_batch_wait(q[(size_t)(b % DEPTH)], &o));
          const int64_t m = o->num_rows;
          if (rep == 0 && m) {
            const int64_t *ov = (const int64_t *)o->columns[0].values;
            const int64_t *iv = v.data() + b * B;
            int64_t j = 0;
            for (int64_t r = 0; r < batches[(size_t)b].num_rows && j < m; r++)
              if (iv[r] > k) order_ok = order_ok && ov[j++] == iv[r];
            order_ok = order_ok && j == m;
          }
          got += m;
          sqlrs_batch_release(o);
          return 0;
        };
        for (int64_t b = 0; b < nb; b++) {
          if (b >= DEPTH && take(b - DEPTH)) return 1;
          CHECK(sqlrs_filter_push_async(f, ptrs[(size_t)b], &q[(size_t)(b % DEPTH)]));
        }
        for (int64_t b = std::max<int64_t>(0, nb - DEPTH); b < nb; b++)
          if (take(b)) return 1;
      } else {
        for (int64_t b0 = 0; b0 < nb; b0 += group) {
          const int g = (int)std::min<int64_t>(group, nb - b0);
          CHECK(sqlrs_filter_push_many(f, g, ptrs.data() + b0, SQLRS_MEM_HOST, outs.data()));
          for (int i = 0; i < g; i++) {
            const int64_t m = outs[(size_t)i]->num_rows;
            if (rep == 0 && m) { // one output batch per input batch, rows in input order
              const int64_t *ov = (const int64_t *)outs[(size_t)i]->columns[0].values;
              const int64_t *iv = v.data() + (b0 + i) * B;
              int64_t j = 0;
              for (int64_t r = 0; r < batches[(size_t)(b0 + i)].num_rows && j < m; r++)
                if (iv[r] > k) order_ok = order_ok && ov[j++] == iv[r];
              order_ok = order_ok && j == m;
            }
            got += m;
            sqlrs_batch_release(outs[(size_t)i]);
          }
        }
      }
      sqlrs_filter_destroy(f);
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      kept[mode] = got;
      if (rep > 0 && ms < best[mode]) best[mode] = ms;
    }
  }
  const bool ok = kept[0] == expect && kept[1] == expect && kept[2] == expect && order_ok;
  std::printf("{\"rows\": %lld, \"batches\": %lld, \"batch_rows\": %lld, \"kept\": %lld, \"ms_push\": %.1f, \"Mrows_s_push\": %.1f, "
              "\"group\": %d, \"ms_push_many\": %.1f, \"Mrows_s_push_many\": %.1f, \"depth\": %d, \"ms_push_async\": %.1f, "
              "\"Mrows_s_push_async\": %.1f, \"check\": \"%s\", \"note\": \"native caller (C ABI): "
              "pageable %lld-row host batches, one result batch per input batch on the host; push = sqlrs_filter_push per batch, "
              "push_many = sqlrs_filter_push_many over groups of batches, push_async = sqlrs_filter_push_async per batch with `depth` "
              "tickets in flight + sqlrs_batch_wait; best of 2 after a warm-up\"}\n",
              (long long)n, (long long)nb, (long long)B, (long long)kept[1], best[0], (double)n / best[0] / 1e3, group, best[1],
              (double)n / best[1] / 1e3, DEPTH, best[2], (double)n / best[2] / 1e3, ok ? "OK" : "mismatch", (long long)B);
  sqlrs_ctx_destroy(ctx);
  return ok ? 0 : 1;
}

// ./bench_host_batches probe [rows = 2e7] [build = 1e6] [batch = 1024]
// HashJoinExecutor (C3's join: fact x dim on an int64 key, Inner, every probe row matches) with the PROBE side fed as pageable
// 1024-row host batches, joined batches (dim key, dim payload, fact key, fact val) on the host: sqlrs_hash_join_probe_push per batch.
static int bench_probe(int argc, char **argv) {
  const int64_t n = argc > 2 ? (int64_t)std::atof(argv[2]) : 20000000, nB = argc > 3 ? (int64_t)std::atof(argv[3]) : 1000000;
  const int64_t B = argc > 4 ? std::atoll(argv[4]) : 1024;
  sqlrs_ctx_t *ctx = nullptr;
  if (sqlrs_ctx_create(0, &ctx) != SQLRS_OK) {
    std::printf("{\"error\": \"no device\"}\n");
    return 2;
  }
  std::vector<int64_t> dk((size_t)nB), dp((size_t)nB), fk((size_t)n);
  std::vector<double> fv((size_t)n);
  for (int64_t i = 0; i < nB; i++) {
    dk[(size_t)i] = (i * 7919) % nB; // a permutation when gcd(7919, nB) = 1
    dp[(size_t)i] = dk[(size_t)i] * 3 + 1;
  }
  for (int64_t i = 0; i < n; i++) {
    fk[(size_t)i] = (int64_t)(splitmix64(0xF1, (uint64_t)i) % (uint64_t)nB);
    fv[(size_t)i] = (double)(splitmix64(0xF2, (uint64_t)i) >> 11) * (1.0 / 9007199254740992.0);
  }
  sqlrs_expr_node_t k0{};
  k0.op = SQLRS_EXPR_INPUT_REF;
  k0.index = 0;
  sqlrs_expr_t key{&k0, 1, 0};
  const int32_t right_dtypes[2] = {SQLRS_INT64, SQLRS_FLOAT64};
  double best = 1e30, best_many = 1e30;
  int64_t joined = 0, joined_many = 0;
  bool ok = true;
  { // the same probe batches through sqlrs_hash_join_probe_push_many, 1024 batches per call
    const int64_t nb = (n + B - 1) / B;
    const int group = 1024;
    std::vector<sqlrs_column_t> cols((size_t)nb * 2);
    std::vector<sqlrs_batch_t> batches((size_t)nb);
    std::vector<const sqlrs_batch_t *> ptrs((size_t)nb);
    for (int64_t b = 0; b < nb; b++) {
      const int64_t m = std::min<int64_t>(B, n - b * B);
      host_col(cols[(size_t)b * 2], SQLRS_INT64, fk.data() + b * B, m);
      host_col(cols[(size_t)b * 2 + 1], SQLRS_FLOAT64, fv.data() + b * B, m);
      std::memset(&batches[(size_t)b], 0, sizeof(sqlrs_batch_t));
      batches[(size_t)b].num_rows = m;
      batches[(size_t)b].num_columns = 2;
      batches[(size_t)b].columns = &cols[(size_t)b * 2];
      ptrs[(size_t)b] = &batches[(size_t)b];
    }
    for (int rep = 0; rep < 3; rep++) {
      auto t0 = std::chrono::steady_clock::now();
      sqlrs_hash_join_t *j = nullptr;
      CHECK(sqlrs_hash_join_create(ctx, SQLRS_JOIN_INNER, 1, &key, &key, nullptr, 2, right_dtypes, &j));
      sqlrs_column_t lc[2];
      host_col(lc[0], SQLRS_INT64, dk.data(), nB);
      host_col(lc[1], SQLRS_INT64, dp.data(), nB);
      sqlrs_batch_t lb{};
      lb.num_rows = nB;
      lb.num_columns = 2;
      lb.columns = lc;
      CHECK(sqlrs_hash_join_build_push(j, &lb));
      CHECK(sqlrs_hash_join_build_finish(j));
      joined_many = 0;
      std::vector<sqlrs_batch_t *> outs((size_t)group);
      for (int64_t b0 = 0; b0 < nb; b0 += group) {
        const int g = (int)std::min<int64_t>(group, nb - b0);
        CHECK(sqlrs_hash_join_probe_push_many(j, g, ptrs.data() + b0, SQLRS_MEM_HOST, outs.data()));
        for (int i = 0; i < g; i++) {
          sqlrs_batch_t *o = outs[(size_t)i];
          if (!o) continue;
          if (rep == 0 && o->num_rows == batches[(size_t)(b0 + i)].num_rows) {
            const int64_t *k = (const int64_t *)o->columns[0].values, *p = (const int64_t *)o->columns[1].values;
            const int64_t *rk = (const int64_t *)o->columns[2].values;
            for (int64_t r = 0; r < o->num_rows; r += 97)
              ok = ok && k[r] == fk[(size_t)((b0 + i) * B + r)] && p[r] == 3 * k[r] + 1 && rk[r] == k[r];
          } else if (rep == 0) {
            ok = false;
          }
          joined_many += o->num_rows;
          sqlrs_batch_release(o);
        }
      }
      sqlrs_hash_join_destroy(j);
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (rep > 0 && ms < best_many) best_many = ms;
    }
    ok = ok && joined_many == n;
  }
  for (int rep = 0; rep < 3; rep++) {
    auto t0 = std::chrono::steady_clock::now();
    sqlrs_hash_join_t *j = nullptr;
    CHECK(sqlrs_hash_join_create(ctx, SQLRS_JOIN_INNER, 1, &key, &key, nullptr, 2, right_dtypes, &j));
    sqlrs_column_t lc[2];
    host_col(lc[0], SQLRS_INT64, dk.data(), nB);
    host_col(lc[1], SQLRS_INT64, dp.data(), nB);
    sqlrs_batch_t lb{};
    lb.num_rows = nB;
    lb.num_columns = 2;
    lb.columns = lc;
    CHECK(sqlrs_hash_join_build_push(j, &lb));
    CHECK(sqlrs_hash_join_build_finish(j));
    joined = 0;
    for (int64_t lo = 0; lo < n; lo += B) {
      const int64_t m = std::min<int64_t>(B, n - lo);
      sqlrs_column_t rc[2];
      host_col(rc[0], SQLRS_INT64, fk.data() + lo, m);
      host_col(rc[1], SQLRS_FLOAT64, fv.data() + lo, m);
      sqlrs_batch_t rb{};
      rb.num_rows = m;
      rb.num_columns = 2;
      rb.columns = rc;
      sqlrs_batch_t *o = nullptr;
      CHECK(sqlrs_hash_join_probe_push(j, &rb, SQLRS_MEM_HOST, &o));
      if (o) {
        if (rep == 0 && o->num_rows == m) { // PK-FK: one joined row per probe row, probe order; payload = 3 * key + 1
          const int64_t *k = (const int64_t *)o->columns[0].values, *p = (const int64_t *)o->columns[1].values;
          for (int64_t r = 0; r < m; r += 97) ok = ok && k[r] == fk[(size_t)(lo + r)] && p[r] == 3 * k[r] + 1;
        }
        joined += o->num_rows;
        sqlrs_batch_release(o);
      }
    }
    sqlrs_hash_join_destroy(j);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rep > 0 && ms < best) best = ms;
  }
  ok = ok && joined == n;
  // one probe batch per call without a synchronisation per call: sqlrs_hash_join_probe_push_async + sqlrs_batch_wait
  const int DEPTH = 8;
  double best_async = 1e30;
  int64_t joined_async = 0;
  for (int rep = 0; rep < 3; rep++) {
    auto t0 = std::chrono::steady_clock::now();
    sqlrs_hash_join_t *j = nullptr;
    CHECK(sqlrs_hash_join_create(ctx, SQLRS_JOIN_INNER, 1, &key, &key, nullptr, 2, right_dtypes, &j));
    sqlrs_column_t lc[2];
    host_col(lc[0], SQLRS_INT64, dk.data(), nB);
    host_col(lc[1], SQLRS_INT64, dp.data(), nB);
    sqlrs_batch_t lb{};
    lb.num_rows = nB;
    lb.num_columns = 2;
    lb.columns = lc;
    CHECK(sqlrs_hash_join_build_push(j, &lb));
    CHECK(sqlrs_hash_join_build_finish(j));
    joined_async = 0;
    const int64_t nb = (n + B - 1) / B;
    std::vector<sqlrs_ticket_t *> q((size_t)DEPTH, nullptr);
    auto take = [&](int64_t b) {
      sqlrs_batch_t *o = nullptr;
      CHECK(sqlrs_batch_wait(q[(size_t)(b % DEPTH)], &o));
      if (o) {
        const int64_t m = std::min<int64_t>(B, n - b * B);
        if (rep == 0 && o->num_rows == m) {
          const int64_t *k = (const int64_t *)o->columns[0].values, *p = (const int64_t *)o->columns[1].values;
          const int64_t *rk = (const int64_t *)o->columns[2].values;
          const double *rv = (const double *)o->columns[3].values;
          for (int64_t r = 0; r < m; r += 97)
            ok = ok && k[r] == fk[(size_t)(b * B + r)] && p[r] == 3 * k[r] + 1 && rk[r] == k[r] && rv[r] == fv[(size_t)(b * B + r)];
        } else if (rep == 0)
          ok = false;
        joined_async += o->num_rows;
        sqlrs_batch_release(o);
      }
      return 0;
    };
    for (int64_t b = 0; b < nb; b++) {
      if (b >= DEPTH && take(b - DEPTH)) return 1;
      const int64_t lo = b * B, m = std::min<int64_t>(B, n - lo);
      sqlrs_column_t rc[2];
      host_col(rc[0], SQLRS_INT64, fk.data() + lo, m);
      host_col(rc[1], SQLRS_FLOAT64, fv.data() + lo, m);
      sqlrs_batch_t rb{};
      rb.num_rows = m;
      rb.num_columns = 2;
      rb.columns = rc;
      CHECK(sqlrs_hash_join_probe_push_async(j, &rb, &q[(size_t)(b % DEPTH)]));
    }
    for (int64_t b = std::max<int64_t>(0, nb - DEPTH); b < nb; b++)
      if (take(b)) return 1;
    sqlrs_hash_join_destroy(j);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rep > 0 && ms < best_async) best_async = ms;
  }
  ok = ok && joined_async == n;
  std::printf("{\"depth\": %d, \"ms_push_async\": %.1f, \"Mrows_s_push_async\": %.1f, ", DEPTH, best_async, (double)n / best_async / 1e3);
  std::printf("\"probe_rows\": %lld, \"build_rows\": %lld, \"batch_rows\": %lld, \"joined\": %lld, \"ms_push\": %.1f, \"Mrows_s_push\": %.1f, "
              "\"group\": 1024, \"ms_push_many\": %.1f, \"Mrows_s_push_many\": %.1f, "
              "\"check\": \"%s\", \"note\": \"native caller (C ABI): build side one host batch, probe side pageable %lld-row host batches, "
              "joined batches (4 columns) on the host, one per probe batch; push = sqlrs_hash_join_probe_push per batch, push_many = "
              "sqlrs_hash_join_probe_push_many over groups of batches; build included; best of 2 after a warm-up\"}\n",
              (long long)n, (long long)nB, (long long)B, (long long)joined, best, (double)n / best / 1e3, best_many, (double)n / best_many / 1e3,
              ok ? "OK" : "mismatch", (long long)B);
  sqlrs_ctx_destroy(ctx);
  return ok ? 0 : 1;
}

// ./bench_host_batches project [rows = 2e7] [batch = 1024]
// ProjectExecutor `SELECT v, v * 3 + 1, v > 2^30` over pageable 1024-row host batches (one bare column, two computed ones):
// sqlrs_project_push per batch against sqlrs_project_push_async with DEPTH tickets in flight.
static int bench_project(int argc, char **argv) {
  const int64_t n = argc > 2 ? (int64_t)std::atof(argv[2]) : 20000000, B = argc > 3 ? std::atoll(argv[3]) : 1024;
  sqlrs_ctx_t *ctx = nullptr;
  if (sqlrs_ctx_create(0, &ctx) != SQLRS_OK) {
    std::printf("{\"error\": \"no device\"}\n");
    return 2;
  }
  std::vector<int64_t> v((size_t)n);
  for (int64_t i = 0; i < n; i++) v[(size_t)i] = (int64_t)(splitmix64(0xC2, (uint64_t)i) % (1ull << 31));
  const int64_t k = (1ll << 30);
  sqlrs_expr_node_t e0[1] = {}, e1[5] = {}, e2[3] = {};
  e0[0].op = SQLRS_EXPR_INPUT_REF;
  e1[0].op = SQLRS_EXPR_INPUT_REF;
  e1[1].op = SQLRS_EXPR_CONSTANT; e1[1].dtype = SQLRS_INT64; e1[1].i = 3;
  e1[2].op = SQLRS_EXPR_MULTIPLY;
  e1[3].op = SQLRS_EXPR_CONSTANT; e1[3].dtype = SQLRS_INT64; e1[3].i = 1;
  e1[4].op = SQLRS_EXPR_PLUS;
  e2[0].op = SQLRS_EXPR_INPUT_REF;
  e2[1].op = SQLRS_EXPR_CONSTANT; e2[1].dtype = SQLRS_INT64; e2[1].i = k;
  e2[2].op = SQLRS_EXPR_GT;
  sqlrs_expr_t exprs[3] = {{e0, 1, 0}, {e1, 5, 0}, {e2, 3, 0}};
  const int64_t nb = (n + B - 1) / B;
  std::vector<sqlrs_column_t> cols((size_t)nb);
  std::vector<sqlrs_batch_t> batches((size_t)nb);
  for (int64_t b = 0; b < nb; b++) {
    const int64_t m = std::min<int64_t>(B, n - b * B);
    host_col(cols[(size_t)b], SQLRS_INT64, v.data() + b * B, m);
    std::memset(&batches[(size_t)b], 0, sizeof(sqlrs_batch_t));
    batches[(size_t)b].num_rows = m;
    batches[(size_t)b].num_columns = 1;
    batches[(size_t)b].columns = &cols[(size_t)b];
  }
  const int DEPTH = 8;
  double best[2] = {1e30, 1e30};
  bool ok = true;
  int64_t rows_out[2] = {0, 0};
  for (int mode = 0; mode < 2; mode++)
    for (int rep = 0; rep < 3; rep++) {
      auto t0 = std::chrono::steady_clock::now();
      sqlrs_project_t *pj = nullptr;
      CHECK(sqlrs_project_create(ctx, 3, exprs, &pj));
      int64_t got = 0;
      auto consume = [&](sqlrs_batch_t *o, int64_t b) {
        const int64_t m = o->num_rows;
        if (rep == 0) { // every row of every column
          const int64_t *iv = v.data() + b * B, *c0 = (const int64_t *)o->columns[0].values, *c1 = (const int64_t *)o->columns[1].values;
          const uint8_t *c2 = (const uint8_t *)o->columns[2].values;
          ok = ok && m == batches[(size_t)b].num_rows && o->num_columns == 3;
          for (int64_t r = 0; ok && r < m; r++)
            ok = c0[r] == iv[r] && c1[r] == iv[r] * 3 + 1 && (((c2[r >> 3] >> (r & 7)) & 1) != 0) == (iv[r] > k);
        }
        got += m;
        sqlrs_batch_release(o);
      };
      if (mode == 0) {
        for (int64_t b = 0; b < nb; b++) {
          sqlrs_batch_t *o = nullptr;
          CHECK(sqlrs_project_push(pj, &batches[(size_t)b], SQLRS_MEM_HOST, &o));
          consume(o, b);
        }
      } else {
        std::vector<sqlrs_ticket_t *> q((size_t)DEPTH, nullptr);
        for (int64_t b = 0; b < nb + DEPTH; b++) {
          if (b >= DEPTH) {
            sqlrs_batch_t *o = nullptr;
            CHECK(sqlrs_batch_wait(q[(size_t)(b % DEPTH)], &o));
            consume(o, b - DEPTH);
          }
          if (b < nb) CHECK(sqlrs_project_push_async(pj, &batches[(size_t)b], &q[(size_t)(b % DEPTH)]));
        }
      }
      sqlrs_project_destroy(pj);
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      rows_out[mode] = got;
      if (rep > 0 && ms < best[mode]) best[mode] = ms;
    }
  ok = ok && rows_out[0] == n && rows_out[1] == n;
  std::printf("{\"rows\": %lld, \"batches\": %lld, \"batch_rows\": %lld, \"ms_push\": %.1f, \"Mrows_s_push\": %.1f, \"depth\": %d, "
              "\"ms_push_async\": %.1f, \"Mrows_s_push_async\": %.1f, \"check\": \"%s\", \"note\": \"native caller (C ABI): SELECT v, v * 3 + 1, "
              "v > 2^30 over pageable %lld-row host batches, one result batch per input batch on the host; best of 2 after a warm-up\"}\n",
              (long long)n, (long long)nb, (long long)B, best[0], (double)n / best[0] / 1e3, DEPTH, best[1], (double)n / best[1] / 1e3,
              ok ? "OK" : "mismatch", (long long)B);
  sqlrs_ctx_destroy(ctx);
  return ok ? 0 : 1;
}

int main(int argc, char **argv) {
  if (argc > 1 && std::strcmp(argv[1], "filter") == 0) return bench_filter(argc, argv);
  if (argc > 1 && std::strcmp(argv[1], "project") == 0) return bench_project(argc, argv);
  if (argc > 1 && std::strcmp(argv[1], "probe") == 0) return bench_probe(argc, argv);
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
