/*
 * cpu_fair.cpp — the "fair" CPU baseline of SURVEY.md §8d(ii) for the headline workload (C5):
 *
 *     SELECT d.key, COUNT(f.val), SUM(f.val) FROM fact f JOIN dim d ON f.key = d.key
 *     WHERE f.val > threshold GROUP BY d.key
 *
 * TEST / BENCH INFRASTRUCTURE ONLY (like sqlrs_oracle.cpp): loaded by bench.py's `cpu_baseline`
 * leg and by tests/test_cpu_fair.py, never by anything under sqlrs_amd/.
 *
 * The faithful oracle (sqlrs_oracle.cpp) restates what the reference does today: one thread,
 * std::unordered_map keyed by hash, per-batch per-group take + accumulate
 * (hash_join.rs:146-323, hash_agg.rs:32-150).  This file is the honest comparison point a CPU
 * engineer would write for the same query on the same box: all host cores (OpenMP), radix
 * partitioning by key hash with software write-combining, one flat open-addressing table per
 * partition (sized to stay in a core's L2), filter fused into the partition pass.  Same results
 * as the oracle up to group order (rows come out partition by partition; callers sort by key)
 * and SUM(double) association order.
 *
 * Also holds the SplitMix64 column generators of sqlrs_amd/datagen.py so that a multi-GB sample
 * is produced at memory speed instead of through numpy temporaries.
 */
#include <omp.h>

#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

inline uint64_t splitmix64(uint64_t seed, uint64_t idx) {
  uint64_t z = (idx + 1) * 0x9E3779B97F4A7C15ull + seed;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
inline uint64_t mix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
}
struct Row {
  int64_t key;
  double val;
};
constexpr int WC = 8; // rows per write-combining line (8 x 16 B = 128 B)

double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

} // namespace

extern "C" {

int fair_max_threads(void) { return omp_get_max_threads(); }

// datagen.key_np / val_np / dim_key_np, rows [start, start + n)
void fair_gen_key(uint64_t seed, int64_t start, int64_t n, uint64_t modulus, int64_t *out) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) out[i] = (int64_t)((splitmix64(seed, (uint64_t)(start + i)) >> 11) % modulus);
}
void fair_gen_val(uint64_t seed, int64_t start, int64_t n, double *out) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++)
    out[i] = (double)(splitmix64(seed, (uint64_t)(start + i)) >> 11) * (1.0 / 9007199254740992.0);
}
void fair_gen_dim_key(int64_t start, int64_t n, uint64_t n_dim, uint64_t multiplier, int64_t *out) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) out[i] = (int64_t)((((uint64_t)(start + i)) * multiplier + 12345ull) % n_dim);
}

/* Filter -> HashJoin(Inner) -> HashAgg(COUNT, SUM) on `threads` threads (0 = all).
 * out_* must hold n_dim entries (one group per distinct dim key at most).  Duplicate dim keys
 * multiply the joined rows like the reference's Vec<usize> per hash (hash_join.rs:225-234).
 * Returns 0, or 1 on allocation failure.  *seconds = wall time of the query itself. */
int fair_c5(int64_t n_fact, int64_t n_dim, double threshold, int threads, const int64_t *fact_key,
            const double *fact_val, const int64_t *dim_key, int64_t *out_keys, int64_t *out_counts,
            double *out_sums, int64_t *n_groups, double *seconds) {
  const int T = threads > 0 ? threads : omp_get_max_threads();
  const double t0 = now();
  // partitions: ~8K dim keys each, so that a partition's table (2 x 8K x 32 B) stays in L2
  int pbits = 0;
  while (pbits < 14 && ((int64_t)8192 << pbits) < n_dim) pbits++;
  const int64_t P = (int64_t)1 << pbits;
  const uint64_t pmask = (uint64_t)P - 1;
  std::vector<int64_t> fh((size_t)T * P, 0), dh((size_t)T * P, 0), fo((size_t)T * P), doff((size_t)T * P);
  std::vector<int64_t> fstart((size_t)P + 1), dstart((size_t)P + 1);
  auto slice = [&](int64_t n, int t, int64_t *lo, int64_t *hi) {
    *lo = n * t / T;
    *hi = n * (t + 1) / T;
  };
  // ---- pass 1: histograms (filter fused: only kept fact rows are counted)
#pragma omp parallel num_threads(T)
  {
    const int t = omp_get_thread_num();
    int64_t lo, hi;
    slice(n_fact, t, &lo, &hi);
    int64_t *h = &fh[(size_t)t * P];
    for (int64_t i = lo; i < hi; i++)
      if (fact_val[i] > threshold) h[mix64((uint64_t)fact_key[i]) & pmask]++;
    slice(n_dim, t, &lo, &hi);
    h = &dh[(size_t)t * P];
    for (int64_t i = lo; i < hi; i++) h[mix64((uint64_t)dim_key[i]) & pmask]++;
  }
  int64_t ftot = 0, dtot = 0;
  for (int64_t p = 0; p < P; p++) {
    fstart[(size_t)p] = ftot;
    dstart[(size_t)p] = dtot;
    for (int t = 0; t < T; t++) {
      fo[(size_t)t * P + p] = ftot;
      ftot += fh[(size_t)t * P + p];
      doff[(size_t)t * P + p] = dtot;
      dtot += dh[(size_t)t * P + p];
    }
  }
  fstart[(size_t)P] = ftot;
  dstart[(size_t)P] = dtot;
  Row *frows = (Row *)std::malloc(sizeof(Row) * (size_t)(ftot > 0 ? ftot : 1));
  int64_t *dkeys = (int64_t *)std::malloc(8 * (size_t)(dtot > 0 ? dtot : 1));
  if (!frows || !dkeys) {
    std::free(frows);
    std::free(dkeys);
    return 1;
  }
  // ---- pass 2: scatter through per-thread write-combining lines
#pragma omp parallel num_threads(T)
  {
    const int t = omp_get_thread_num();
    std::vector<Row> buf((size_t)P * WC);
    std::vector<uint8_t> fill((size_t)P, 0);
    int64_t *o = &fo[(size_t)t * P];
    int64_t lo, hi;
    slice(n_fact, t, &lo, &hi);
    for (int64_t i = lo; i < hi; i++) {
      const double v = fact_val[i];
      if (!(v > threshold)) continue;
      const int64_t k = fact_key[i];
      const size_t p = (size_t)(mix64((uint64_t)k) & pmask);
      Row *b = &buf[p * WC];
      b[fill[p]] = Row{k, v};
      if (++fill[p] == WC) {
        std::memcpy(frows + o[p], b, sizeof(Row) * WC);
        o[p] += WC;
        fill[p] = 0;
      }
    }
    for (size_t p = 0; p < (size_t)P; p++)
      if (fill[p]) {
        std::memcpy(frows + o[p], &buf[p * WC], sizeof(Row) * fill[p]);
        o[p] += fill[p];
      }
    slice(n_dim, t, &lo, &hi);
    int64_t *od = &doff[(size_t)t * P];
    for (int64_t i = lo; i < hi; i++) {
      const int64_t k = dim_key[i];
      dkeys[od[mix64((uint64_t)k) & pmask]++] = k;
    }
  }
  // ---- pass 3: one flat table per partition: build from the dim keys, probe with the fact rows
  int64_t emitted = 0;
#pragma omp parallel num_threads(T)
  {
    std::vector<int64_t> tkey, tcnt;
    std::vector<int32_t> tmult;
    std::vector<double> tsum;
#pragma omp for schedule(dynamic, 1)
    for (int64_t p = 0; p < P; p++) {
      const int64_t nd = dstart[(size_t)p + 1] - dstart[(size_t)p];
      if (nd == 0) continue;
      size_t cap = 16;
      while (cap < (size_t)nd * 2) cap <<= 1;
      const uint64_t m = cap - 1;
      tkey.assign(cap, 0);
      tcnt.assign(cap, 0);
      tmult.assign(cap, 0);
      tsum.assign(cap, 0.0);
      for (int64_t i = dstart[(size_t)p]; i < dstart[(size_t)p + 1]; i++) {
        const int64_t k = dkeys[i];
        uint64_t s = (mix64((uint64_t)k) >> pbits) & m;
        while (tmult[s] && tkey[s] != k) s = (s + 1) & m;
        tkey[s] = k;
        tmult[s]++;
      }
      for (int64_t i = fstart[(size_t)p]; i < fstart[(size_t)p + 1]; i++) {
        const int64_t k = frows[i].key;
        uint64_t s = (mix64((uint64_t)k) >> pbits) & m;
        while (tmult[s] && tkey[s] != k) s = (s + 1) & m;
        if (!tmult[s]) continue; // no build partner: Inner join drops the row
        tcnt[s] += tmult[s];
        tsum[s] += frows[i].val * (double)tmult[s];
      }
      int64_t mine = 0;
      for (size_t s = 0; s < cap; s++) mine += tcnt[s] != 0;
      int64_t at;
#pragma omp atomic capture
      {
        at = emitted;
        emitted += mine;
      }
      for (size_t s = 0; s < cap; s++)
        if (tcnt[s]) {
          out_keys[at] = tkey[s];
          out_counts[at] = tcnt[s];
          out_sums[at] = tsum[s];
          at++;
        }
    }
  }
  std::free(frows);
  std::free(dkeys);
  *n_groups = emitted;
  *seconds = now() - t0;
  return 0;
}

} // extern "C"
