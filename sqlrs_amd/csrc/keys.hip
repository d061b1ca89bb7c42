// keys.hip — join / group keys as one u64 per row (see prims.hpp NKeys).
//
// Reference semantics being preserved (hash_utils.rs:161-220, SURVEY §8a quirks 2-4,11):
//  * keys are compared at their native width and f64 by bit pattern;
//  * a NULL key never changes the running hash, so NULL keys equal each other;
//  * multi-column / Utf8 keys are matched by 64-bit hash only — here with this library's own
//    mixer instead of ahash (hash values are never observable in operator output).
#include "common.hpp"
#include "device_utils.hpp"
#include "prims.hpp"

namespace sq {

template <class T> __global__ void widen_kernel(const T *__restrict__ in, int64_t n,
                                                uint64_t *__restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = (uint64_t)(int64_t)in[i];
}
__global__ void widen_bool_kernel(const uint64_t *__restrict__ in, int64_t n,
                                  uint64_t *__restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = (in[i >> 6] >> (i & 63)) & 1;
}

__device__ __forceinline__ uint64_t combine_hashes(uint64_t l, uint64_t r) { // hash_utils.rs:13-16
  uint64_t h = (uint64_t)(17 * 37) + l;
  return h * 37 + r;
}

// folds one column into the running per-row hash (hash stays put on NULL)
template <class T>
__global__ void fold_fixed_kernel(const T *__restrict__ in, const uint64_t *__restrict__ validity,
                                  int64_t n, uint64_t tag, int multi, uint64_t *__restrict__ h) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (validity && !((validity[i >> 6] >> (i & 63)) & 1)) return;
  uint64_t v = mix64((uint64_t)in[i] + tag);
  h[i] = multi ? combine_hashes(v, h[i]) : v;
}
__global__ void fold_bool_kernel(const uint64_t *__restrict__ in, const uint64_t *__restrict__ validity,
                                 int64_t n, int multi, uint64_t *__restrict__ h) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (validity && !((validity[i >> 6] >> (i & 63)) & 1)) return;
  uint64_t v = mix64(((in[i >> 6] >> (i & 63)) & 1) ^ 0x0808080808080808ULL);
  h[i] = multi ? combine_hashes(v, h[i]) : v;
}
__global__ void fold_utf8_kernel(const uint8_t *__restrict__ data, const int32_t *__restrict__ off,
                                 const uint64_t *__restrict__ validity, int64_t n, int multi,
                                 uint64_t *__restrict__ h) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (validity && !((validity[i >> 6] >> (i & 63)) & 1)) return;
  uint64_t x = 0xcbf29ce484222325ULL; // FNV-1a over the bytes, then the mixer
  for (int32_t k = off[i]; k < off[i + 1]; k++) {
    x ^= data[k];
    x *= 0x100000001b3ULL;
  }
  uint64_t v = mix64(x ^ 0x7575757575757575ULL);
  h[i] = multi ? combine_hashes(v, h[i]) : v;
}

// Several fixed-width key columns folded in ONE pass (round 6): the column-by-column form above reads and rewrites the running
// hash once per column behind a memset — 5.6 GB for two int64 columns of 1e8 rows (0.99 ms of the 3.7 ms two-column-key join);
// this reads the columns once and writes the hash once (2.4 GB).  Same arithmetic, same order: h = 0, then per column
// h = combine_hashes(mix64(value + tag), h) unless the value is NULL.
struct FoldCols {
  const void *v[4];
  const uint64_t *valid[4];
  uint64_t tag[4];
  int is32[4];
  int n;
};
__global__ __launch_bounds__(256) void fold_fixed_multi_kernel(FoldCols fc, int64_t rows, uint64_t *__restrict__ h) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < rows; i += (int64_t)gridDim.x * 256) {
    uint64_t acc = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      if (c >= fc.n) break;
      if (fc.valid[c] && !((fc.valid[c][i >> 6] >> (i & 63)) & 1)) continue;
      const uint64_t x = fc.is32[c] ? (uint64_t)((const uint32_t *)fc.v[c])[i] : ((const uint64_t *)fc.v[c])[i];
      acc = combine_hashes(mix64(x + fc.tag[c]), acc);
    }
    h[i] = acc;
  }
}

// Strong (asymmetric, position-dependent) 64-bit fold for library-internal de-duplication keys
// such as (group key..., DISTINCT argument): unlike combine_hashes, (a, b) and (b, a) differ and
// a NULL is a value of its own.  mode: 0 = fixed 8 B, 1 = fixed 4 B, 2 = bool bits, 3 = utf8
template <int MODE>
__global__ void fold_strong_kernel(const void *__restrict__ data, const int32_t *__restrict__ off,
                                   const uint64_t *__restrict__ validity, int64_t n, uint64_t salt,
                                   uint64_t *__restrict__ h) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t v;
  if (validity && !((validity[i >> 6] >> (i & 63)) & 1))
    v = 0x6e756c6c6e756c6cULL; // "nullnull"
  else if (MODE == 0)
    v = mix64(((const uint64_t *)data)[i] + 0x9e3779b97f4a7c15ULL);
  else if (MODE == 1)
    v = mix64((uint64_t)((const uint32_t *)data)[i] ^ 0x3232323200000000ULL);
  else if (MODE == 2)
    v = mix64(((((const uint64_t *)data)[i >> 6] >> (i & 63)) & 1) ^ 0x0808080808080808ULL);
  else {
    uint64_t x = 0xcbf29ce484222325ULL;
    const uint8_t *b = (const uint8_t *)data;
    for (int32_t k = off[i]; k < off[i + 1]; k++) {
      x ^= b[k];
      x *= 0x100000001b3ULL;
    }
    v = mix64(x ^ 0x7575757575757575ULL);
  }
  h[i] = mix64((h[i] ^ salt) * 0x9e3779b97f4a7c15ULL + v);
}

NKeys normalize_keys_strong(Ctx *ctx, const std::vector<DCol> &cols_in, int64_t rows) {
  if (cols_in.empty()) fail(SQLRS_ERR_INTERNAL, "no key columns");
  NKeys k;
  k.rows = rows;
  k.exact = false;
  int64_t n1 = std::max<int64_t>(rows, 1);
  k.keys = ctx->alloc(8 * (size_t)n1);
  SQ_HIP(hipMemsetAsync(k.keys->p, 0, 8 * (size_t)n1, ctx->stream));
  if (rows == 0) return k;
  dim3 g((unsigned)ceil_div(n1, 256)), b(256);
  ProfScope ps(ctx, "normalize_keys");
  uint64_t salt = 0x1234567;
  for (const DCol &cin : cols_in) {
    DCol c = cin.stride == 0 ? materialize_scalar(ctx, cin, rows) : cin;
    const uint64_t *v = (c.validity && c.null_count != 0) ? c.validity : nullptr;
    uint64_t *h = k.keys->as<uint64_t>();
    salt = salt * 6364136223846793005ULL + 1442695040888963407ULL;
    switch (c.dtype) {
    case SQLRS_INT64:
    case SQLRS_FLOAT64:
      fold_strong_kernel<0><<<g, b, 0, ctx->stream>>>(c.values, nullptr, v, rows, salt, h);
      break;
    case SQLRS_INT32:
      fold_strong_kernel<1><<<g, b, 0, ctx->stream>>>(c.values, nullptr, v, rows, salt, h);
      break;
    case SQLRS_BOOLEAN:
      fold_strong_kernel<2><<<g, b, 0, ctx->stream>>>(c.values, nullptr, v, rows, salt, h);
      break;
    case SQLRS_UTF8:
      fold_strong_kernel<3><<<g, b, 0, ctx->stream>>>(c.values, c.offsets, v, rows, salt, h);
      break;
    default:
      fail(SQLRS_ERR_INTERNAL, "Unsupported data type in hasher");
    }
    SQ_HIP(hipGetLastError());
  }
  return k;
}

NKeys normalize_keys(Ctx *ctx, const std::vector<DCol> &cols_in, int64_t rows) {
  if (cols_in.empty()) fail(SQLRS_ERR_INTERNAL, "no key columns");
  std::vector<DCol> cols;
  for (const DCol &c : cols_in) cols.push_back(c.stride == 0 ? materialize_scalar(ctx, c, rows) : c);
  NKeys k;
  k.rows = rows;
  int64_t n1 = std::max<int64_t>(rows, 1);
  dim3 g((unsigned)ceil_div(n1, 256)), b(256);
  ProfScope ps(ctx, "normalize_keys");
  const DCol &c0 = cols[0];
  bool fixed = c0.dtype == SQLRS_INT32 || c0.dtype == SQLRS_INT64 || c0.dtype == SQLRS_FLOAT64 ||
               c0.dtype == SQLRS_BOOLEAN;
  bool alias = cols.size() == 1 && rows > 0 && (c0.dtype == SQLRS_INT64 || c0.dtype == SQLRS_FLOAT64);
  if (!alias) k.keys = ctx->alloc(8 * (size_t)n1);
  if (cols.size() == 1 && fixed) {
    k.exact = true;
    k.dtype = c0.dtype;
    if (c0.validity && c0.null_count != 0) {
      k.validity = c0.validity;
      k.own_validity = c0.own_validity;
    }
    if (rows == 0) return k;
    switch (c0.dtype) {
    case SQLRS_INT64:
    case SQLRS_FLOAT64: // bit pattern: -0.0 != +0.0, NaN payloads distinct (hash_utils.rs:124-131)
      // the column buffer IS the key array: no copy.  A borrowed caller buffer stays valid
      // for the duration of the call that normalises it (operators that retain keys copy).
      if (c0.own_values) {
        k.keys = c0.own_values;
      } else {
        k.keys = std::make_shared<Buf>(ctx, const_cast<void *>(c0.values), 0);
        k.keys->owned = false; // non-owning view of the caller's buffer
      }
      return k;
    case SQLRS_INT32:
      widen_kernel<int32_t><<<g, b, 0, ctx->stream>>>(c0.v<int32_t>(), rows, k.keys->as<uint64_t>());
      break;
    default:
      widen_bool_kernel<<<g, b, 0, ctx->stream>>>(c0.v<uint64_t>(), rows, k.keys->as<uint64_t>());
    }
    SQ_HIP(hipGetLastError());
    return k;
  }
  // hash mode: every_rows_hashes = vec![0; n]; create_hashes(...)   (hash_join.rs:169-170)
  k.exact = false;
  if (rows > 0 && cols.size() >= 2 && cols.size() <= 4) { // all fixed-width: one pass, no memset
    FoldCols fc;
    fc.n = (int)cols.size();
    bool ok = true;
    for (int c = 0; c < 4; c++) {
      fc.v[c] = nullptr;
      fc.valid[c] = nullptr;
      fc.tag[c] = 0;
      fc.is32[c] = 0;
      if (c >= fc.n) continue;
      const DCol &col = cols[(size_t)c];
      ok = ok && (col.dtype == SQLRS_INT32 || col.dtype == SQLRS_INT64 || col.dtype == SQLRS_FLOAT64);
      fc.v[c] = col.values;
      fc.valid[c] = (col.validity && col.null_count != 0) ? col.validity : nullptr;
      fc.is32[c] = col.dtype == SQLRS_INT32;
      fc.tag[c] = col.dtype == SQLRS_INT32 ? 0x3232323200000000ULL : 0x9e3779b97f4a7c15ULL;
    }
    if (ok) {
      const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(rows, 256 * 4), 16 * (int64_t)ctx->num_cus));
      fold_fixed_multi_kernel<<<dim3(blocks), dim3(256), 0, ctx->stream>>>(fc, rows, k.keys->as<uint64_t>());
      SQ_HIP(hipGetLastError());
      return k;
    }
  }
  SQ_HIP(hipMemsetAsync(k.keys->p, 0, 8 * (size_t)n1, ctx->stream));
  if (rows == 0) return k;
  int multi = cols.size() > 1;
  for (const DCol &c : cols) {
    const uint64_t *v = (c.validity && c.null_count != 0) ? c.validity : nullptr;
    uint64_t *h = k.keys->as<uint64_t>();
    switch (c.dtype) {
    case SQLRS_INT32:
      fold_fixed_kernel<uint32_t><<<g, b, 0, ctx->stream>>>(c.v<uint32_t>(), v, rows,
                                                            0x3232323200000000ULL, multi, h);
      break;
    case SQLRS_INT64:
    case SQLRS_FLOAT64:
      fold_fixed_kernel<uint64_t><<<g, b, 0, ctx->stream>>>(c.v<uint64_t>(), v, rows,
                                                            0x9e3779b97f4a7c15ULL, multi, h);
      break;
    case SQLRS_BOOLEAN:
      fold_bool_kernel<<<g, b, 0, ctx->stream>>>(c.v<uint64_t>(), v, rows, multi, h);
      break;
    case SQLRS_UTF8:
      fold_utf8_kernel<<<g, b, 0, ctx->stream>>>(c.v<uint8_t>(), c.offsets, v, rows, multi, h);
      break;
    default:
      fail(SQLRS_ERR_INTERNAL, "Unsupported data type in hasher"); // hash_utils.rs:210-216
    }
    SQ_HIP(hipGetLastError());
  }
  return k;
}

} // namespace sq
