// expr.hip — BoundExpr::eval_column on device (evaluator.rs:13-28, array_compute.rs:70-90).
//
// One elementwise kernel per expression node over (pointer, stride) operands; a constant is
// a stride-0 operand (never materialised per batch, unlike evaluator.rs:21).  Comparison
// results are Arrow bitmaps built with one ballot per 64 rows; AND/OR are Kleene on whole
// words.  All kernels are streaming and HBM-bound.
#include "common.hpp"
#include "device_utils.hpp"
#include "prims.hpp"

namespace sq {

enum { AR_ADD = 0, AR_SUB, AR_MUL, AR_DIV };
enum { CMP_GT = 0, CMP_LT, CMP_GE, CMP_LE, CMP_EQ, CMP_NE };

template <class T> struct OrdKey {
  static __device__ __forceinline__ T key(T v) { return v; }
};
template <> struct OrdKey<double> { // total order, like arrow-array 28's float compare
  static __device__ __forceinline__ uint64_t key(double v) { return f64_to_ordered(v); }
};

template <class T, int OP>
__global__ __launch_bounds__(BLOCK) void arith_kernel(const T *__restrict__ a, int sa,
                                                      const T *__restrict__ b, int sb,
                                                      const uint64_t *__restrict__ valid, int64_t n,
                                                      T *__restrict__ out, int *err) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  T x = a[sa ? i : 0], y = b[sb ? i : 0], r;
  if (OP == AR_DIV) {
    bool ok = !valid || ((valid[i >> 6] >> (i & 63)) & 1);
    if (ok && y == T(0)) {
      *err = 1; // arrow: DivideByZero
      out[i] = T(0);
      return;
    }
    if (!ok) {
      out[i] = T(0);
      return;
    }
  }
  if constexpr (std::is_floating_point<T>::value) {
    r = OP == AR_ADD ? x + y : OP == AR_SUB ? x - y : OP == AR_MUL ? x * y : x / y;
  } else {
    using U = typename std::make_unsigned<T>::type;
    U ux = (U)x, uy = (U)y;
    if (OP == AR_ADD) r = (T)(ux + uy);
    else if (OP == AR_SUB) r = (T)(ux - uy);
    else if (OP == AR_MUL) r = (T)(ux * uy);
    else r = (y == T(-1)) ? (T)(U(0) - ux) : (T)(x / y); // MIN / -1 wraps
  }
  out[i] = r;
}

template <class T, int OP>
__global__ __launch_bounds__(BLOCK) void cmp_kernel(const T *__restrict__ a, int sa,
                                                    const T *__restrict__ b, int sb, int64_t n,
                                                    uint64_t *__restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  bool r = false;
  if (i < n) {
    auto x = OrdKey<T>::key(a[sa ? i : 0]);
    auto y = OrdKey<T>::key(b[sb ? i : 0]);
    r = OP == CMP_GT ? x > y : OP == CMP_LT ? x < y : OP == CMP_GE ? x >= y
        : OP == CMP_LE ? x <= y : OP == CMP_EQ ? x == y : x != y;
  }
  uint64_t m = __ballot(r);
  if (lane_id() == 0 && i < n) out[i >> 6] = m;
}

// Utf8 comparison: byte-wise lexicographic (arrow string order), one row per lane; a stride-0
// operand is a constant (row 0 of its buffers).
template <int OP>
__global__ __launch_bounds__(BLOCK) void cmp_utf8_kernel(const uint8_t *__restrict__ ad,
                                                         const int32_t *__restrict__ ao, int sa,
                                                         const uint8_t *__restrict__ bd,
                                                         const int32_t *__restrict__ bo, int sb, int64_t n,
                                                         uint64_t *__restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  bool r = false;
  if (i < n) {
    int64_t ia = sa ? i : 0, ib = sb ? i : 0;
    int32_t a0 = ao[ia], la = ao[ia + 1] - a0, b0 = bo[ib], lb = bo[ib + 1] - b0;
    int32_t m = la < lb ? la : lb;
    int c = 0;
    for (int32_t k = 0; k < m && c == 0; k++) c = (int)ad[a0 + k] - (int)bd[b0 + k];
    if (c == 0) c = (la > lb) - (la < lb);
    r = OP == CMP_GT ? c > 0 : OP == CMP_LT ? c < 0 : OP == CMP_GE ? c >= 0
        : OP == CMP_LE ? c <= 0 : OP == CMP_EQ ? c == 0 : c != 0;
  }
  uint64_t mm = __ballot(r);
  if (lane_id() == 0 && i < n) out[i >> 6] = mm;
}

// boolean compare / Kleene logic on whole words.  mode: 0..5 = CMP_*, 6 = AND, 7 = OR
__global__ void bool_words_kernel(const uint64_t *__restrict__ a, int sa,
                                  const uint64_t *__restrict__ av, int sav,
                                  const uint64_t *__restrict__ b, int sb,
                                  const uint64_t *__restrict__ bv, int sbv, int mode, int64_t nwords,
                                  uint64_t *__restrict__ out, uint64_t *__restrict__ out_valid) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= nwords) return;
  uint64_t l = a[sa ? i : 0], r = b[sb ? i : 0];
  uint64_t lv = av ? av[sav ? i : 0] : ~0ull, rv = bv ? bv[sbv ? i : 0] : ~0ull;
  uint64_t val, valid;
  if (mode == 6) { // and_kleene
    uint64_t kf = (lv & ~l) | (rv & ~r), kt = lv & l & rv & r;
    val = kt;
    valid = kf | kt;
  } else if (mode == 7) { // or_kleene
    uint64_t kt = (lv & l) | (rv & r), kf = lv & ~l & rv & ~r;
    val = kt;
    valid = kf | kt;
  } else {
    valid = lv & rv;
    switch (mode) {
    case CMP_GT: val = l & ~r; break;
    case CMP_LT: val = ~l & r; break;
    case CMP_GE: val = l | ~r; break;
    case CMP_LE: val = ~l | r; break;
    case CMP_EQ: val = ~(l ^ r); break;
    default: val = l ^ r; break;
    }
  }
  out[i] = val;
  if (out_valid) out_valid[i] = valid;
}

__global__ void and_words_kernel(const uint64_t *__restrict__ a, const uint64_t *__restrict__ b,
                                 int64_t nwords, uint64_t *__restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < nwords) out[i] = a[i] & b[i];
}

// numeric casts; out-of-range -> NULL (arrow safe cast).  `drop` collects newly-null rows.
template <class S, class D>
__global__ __launch_bounds__(BLOCK) void cast_kernel(const S *__restrict__ in, int64_t n,
                                                     D *__restrict__ out,
                                                     uint64_t *__restrict__ ok_bits) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  bool ok = true;
  if (i < n) {
    S v = in[i];
    if constexpr (std::is_floating_point<S>::value && !std::is_floating_point<D>::value) {
      double lim = sizeof(D) == 4 ? 2147483648.0 : 9223372036854775808.0;
      ok = (v > -lim - 1) && (v < lim);
      out[i] = ok ? (D)v : D(0);
    } else if constexpr (sizeof(D) < sizeof(S) && !std::is_floating_point<D>::value) {
      ok = (v <= (S)INT32_MAX) && (v >= (S)INT32_MIN);
      out[i] = ok ? (D)v : D(0);
    } else
      out[i] = (D)v;
  }
  if (ok_bits) {
    uint64_t m = __ballot(ok);
    if (lane_id() == 0 && i < n) ok_bits[i >> 6] = m;
  }
}
template <class D>
__global__ void cast_bool_kernel(const uint64_t *__restrict__ in, int64_t n, D *__restrict__ out) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = (D)((in[i >> 6] >> (i & 63)) & 1);
}

// ------------------------------------------------------------------ host side --
static DCol make_scalar(Ctx *ctx, const sqlrs_expr_node_t &n, const std::string &str, int64_t rows) {
  DCol c;
  c.dtype = n.dtype;
  c.length = rows;
  c.stride = 0;
  c.scalar_null = n.is_null != 0;
  if (n.dtype == SQLRS_UTF8) { // one string: bytes + offsets {0, len}; validity word after them
    size_t len = c.scalar_null ? 0 : str.size();
    size_t off_at = round_up(len + 1, 8);
    std::vector<uint8_t> host(off_at + 16, 0);
    std::memcpy(host.data(), str.data(), len);
    int32_t offs[2] = {0, (int32_t)len};
    std::memcpy(host.data() + off_at, offs, 8);
    uint64_t vword = c.scalar_null ? 0ull : ~0ull;
    std::memcpy(host.data() + off_at + 8, &vword, 8);
    c.own_values = ctx->alloc(host.size());
    SQ_HIP(hipMemcpyAsync(c.own_values->p, host.data(), host.size(), hipMemcpyHostToDevice, ctx->stream));
    ctx->sync();
    c.values = c.own_values->p;
    c.offsets = (const int32_t *)(c.own_values->as<uint8_t>() + off_at);
    c.data_bytes = (int64_t)len;
    c.null_count = c.scalar_null ? rows : 0;
    if (c.scalar_null) c.validity = (const uint64_t *)(c.own_values->as<uint8_t>() + off_at + 8);
    return c;
  }
  uint64_t bits = 0;
  switch (n.dtype) {
  case SQLRS_INT64: bits = (uint64_t)n.i; break;
  case SQLRS_INT32: bits = (uint64_t)(uint32_t)(int32_t)n.i; break;
  case SQLRS_FLOAT64: std::memcpy(&bits, &n.f, 8); break;
  case SQLRS_BOOLEAN: bits = n.i ? ~0ull : 0ull; break;
  default:
    fail(SQLRS_ERR_INTERNAL, "constant of unsupported dtype");
  }
  c.scalar_bits = bits;
  c.null_count = c.scalar_null ? rows : 0;
  // device copy: word 0 = value, word 1 = validity word (all zero when NULL)
  uint64_t host[2] = {bits, c.scalar_null ? 0ull : ~0ull};
  c.own_values = ctx->alloc(16);
  SQ_HIP(hipMemcpyAsync(c.own_values->p, host, 16, hipMemcpyHostToDevice, ctx->stream));
  ctx->sync(); // `host` is a stack temporary
  c.values = c.own_values->p;
  if (c.scalar_null) c.validity = c.own_values->as<uint64_t>() + 1;
  return c;
}

// validity of a binary result: AND of the operands' validity (scalar NULL -> all NULL)
static void combine_validity(Ctx *ctx, const DCol &l, const DCol &r, int64_t rows, DCol &o) {
  if ((l.stride == 0 && l.scalar_null) || (r.stride == 0 && r.scalar_null)) {
    o.own_validity = ctx->alloc_zero(bitmap_bytes(std::max<int64_t>(rows, 1)));
    o.validity = o.own_validity->as<uint64_t>();
    o.null_count = rows;
    return;
  }
  const uint64_t *lv = (l.stride && l.validity && l.null_count != 0) ? l.validity : nullptr;
  const uint64_t *rv = (r.stride && r.validity && r.null_count != 0) ? r.validity : nullptr;
  if (!lv && !rv) {
    o.null_count = 0;
    return;
  }
  if (lv && rv) {
    int64_t nw = ceil_div(std::max<int64_t>(rows, 1), 64);
    o.own_validity = ctx->alloc(8 * (size_t)nw);
    and_words_kernel<<<dim3((unsigned)ceil_div(nw, 256)), dim3(256), 0, ctx->stream>>>(
        lv, rv, nw, o.own_validity->as<uint64_t>());
    SQ_HIP(hipGetLastError());
    o.validity = o.own_validity->as<uint64_t>();
    o.null_count = -1;
    return;
  }
  const DCol &src = lv ? l : r;
  o.validity = src.validity;
  o.own_validity = src.own_validity;
  o.null_count = src.null_count;
}

template <class T>
static void launch_arith(Ctx *ctx, int op, const DCol &l, const DCol &r, const uint64_t *valid,
                         int64_t n, T *out, int *err) {
  dim3 g((unsigned)ceil_div(n, BLOCK)), b(BLOCK);
  const T *a = l.v<T>(), *c = r.v<T>();
  switch (op) {
  case AR_ADD: arith_kernel<T, AR_ADD><<<g, b, 0, ctx->stream>>>(a, l.stride, c, r.stride, valid, n, out, err); break;
  case AR_SUB: arith_kernel<T, AR_SUB><<<g, b, 0, ctx->stream>>>(a, l.stride, c, r.stride, valid, n, out, err); break;
  case AR_MUL: arith_kernel<T, AR_MUL><<<g, b, 0, ctx->stream>>>(a, l.stride, c, r.stride, valid, n, out, err); break;
  default: arith_kernel<T, AR_DIV><<<g, b, 0, ctx->stream>>>(a, l.stride, c, r.stride, valid, n, out, err); break;
  }
  SQ_HIP(hipGetLastError());
}
template <class T>
static void launch_cmp(Ctx *ctx, int op, const DCol &l, const DCol &r, int64_t n, uint64_t *out) {
  int64_t n64 = (int64_t)round_up((size_t)n, 64);
  dim3 g((unsigned)ceil_div(n64, BLOCK)), b(BLOCK);
  const T *a = l.v<T>(), *c = r.v<T>();
  switch (op) {
  case CMP_GT: cmp_kernel<T, CMP_GT><<<g, b, 0, ctx->stream>>>(a, l.stride, c, r.stride, n, out); break;
  case CMP_LT: cmp_kernel<T, CMP_LT><<<g, b, 0, ctx->stream>>>(a, l.stride, c, r.stride, n, out); break;
  case CMP_GE: cmp_kernel<T, CMP_GE><<<g, b, 0, ctx->stream>>>(a, l.stride, c, r.stride, n, out); break;
  case CMP_LE: cmp_kernel<T, CMP_LE><<<g, b, 0, ctx->stream>>>(a, l.stride, c, r.stride, n, out); break;
  case CMP_EQ: cmp_kernel<T, CMP_EQ><<<g, b, 0, ctx->stream>>>(a, l.stride, c, r.stride, n, out); break;
  default: cmp_kernel<T, CMP_NE><<<g, b, 0, ctx->stream>>>(a, l.stride, c, r.stride, n, out); break;
  }
  SQ_HIP(hipGetLastError());
}

static DCol binary_op(Ctx *ctx, const DCol &l, const DCol &r, int op, int64_t rows, BufP &err) {
  DCol o;
  o.length = rows;
  int64_t n1 = std::max<int64_t>(rows, 1);
  ProfScope ps(ctx, "expr_binary");
  if (op >= SQLRS_EXPR_PLUS && op <= SQLRS_EXPR_DIVIDE) {
    if (l.dtype != r.dtype) fail(SQLRS_ERR_INTERNAL, "compute_op failed to downcast array");
    if (l.dtype != SQLRS_INT32 && l.dtype != SQLRS_INT64 && l.dtype != SQLRS_FLOAT64)
      fail(SQLRS_ERR_INTERNAL, "unsupported data type");
    o.dtype = l.dtype;
    combine_validity(ctx, l, r, rows, o);
    size_t w = width_of(o.dtype);
    o.own_values = ctx->alloc(w * (size_t)n1 + 16);
    o.values = o.own_values->p;
    if (rows == 0) return o;
    int aop = op - SQLRS_EXPR_PLUS;
    if (aop == AR_DIV && !err) err = ctx->alloc_zero(8);
    int *e = err ? err->as<int>() : nullptr;
    if (o.dtype == SQLRS_INT64)
      launch_arith<int64_t>(ctx, aop, l, r, o.validity, rows, o.own_values->as<int64_t>(), e);
    else if (o.dtype == SQLRS_INT32)
      launch_arith<int32_t>(ctx, aop, l, r, o.validity, rows, o.own_values->as<int32_t>(), e);
    else
      launch_arith<double>(ctx, aop, l, r, o.validity, rows, o.own_values->as<double>(), e);
    return o;
  }
  o.dtype = SQLRS_BOOLEAN;
  int64_t nw = ceil_div(n1, 64);
  o.own_values = ctx->alloc(8 * (size_t)nw);
  o.values = o.own_values->p;
  if (op >= SQLRS_EXPR_GT && op <= SQLRS_EXPR_NOTEQ) {
    if (l.dtype != r.dtype) fail(SQLRS_ERR_ARROW, "comparison of arrays of different types");
    int cop = op - SQLRS_EXPR_GT;
    if (l.dtype == SQLRS_BOOLEAN) {
      combine_validity(ctx, l, r, rows, o);
      bool_words_kernel<<<dim3((unsigned)ceil_div(nw, 256)), dim3(256), 0, ctx->stream>>>(
          l.v<uint64_t>(), l.stride, nullptr, 0, r.v<uint64_t>(), r.stride, nullptr, 0, cop, nw,
          o.own_values->as<uint64_t>(), nullptr);
      SQ_HIP(hipGetLastError());
      return o;
    }
    combine_validity(ctx, l, r, rows, o);
    if (rows == 0) return o;
    switch (l.dtype) {
    case SQLRS_INT64: launch_cmp<int64_t>(ctx, cop, l, r, rows, o.own_values->as<uint64_t>()); break;
    case SQLRS_INT32: launch_cmp<int32_t>(ctx, cop, l, r, rows, o.own_values->as<uint64_t>()); break;
    case SQLRS_FLOAT64: launch_cmp<double>(ctx, cop, l, r, rows, o.own_values->as<uint64_t>()); break;
    case SQLRS_UTF8: {
      int64_t n64 = (int64_t)round_up((size_t)rows, 64);
      dim3 g((unsigned)ceil_div(n64, BLOCK)), b(BLOCK);
      uint64_t *ob = o.own_values->as<uint64_t>();
#define SQ_U8(OP) cmp_utf8_kernel<OP><<<g, b, 0, ctx->stream>>>(l.v<uint8_t>(), l.offsets, l.stride, r.v<uint8_t>(), r.offsets, r.stride, rows, ob)
      switch (cop) {
      case CMP_GT: SQ_U8(CMP_GT); break;
      case CMP_LT: SQ_U8(CMP_LT); break;
      case CMP_GE: SQ_U8(CMP_GE); break;
      case CMP_LE: SQ_U8(CMP_LE); break;
      case CMP_EQ: SQ_U8(CMP_EQ); break;
      default: SQ_U8(CMP_NE); break;
      }
#undef SQ_U8
      SQ_HIP(hipGetLastError());
      break;
    }
    default:
      fail(SQLRS_ERR_ARROW, "comparison of unsupported type");
    }
    return o;
  }
  if (op == SQLRS_EXPR_AND || op == SQLRS_EXPR_OR) {
    if (l.dtype != SQLRS_BOOLEAN || r.dtype != SQLRS_BOOLEAN)
      fail(SQLRS_ERR_INTERNAL,
           "Cannot evaluate binary expression with non-Boolean types, only Boolean supported");
    const uint64_t *lv = (l.validity && l.null_count != 0) ? l.validity : nullptr;
    const uint64_t *rv = (r.validity && r.null_count != 0) ? r.validity : nullptr;
    uint64_t *ov = nullptr;
    if (lv || rv) {
      o.own_validity = ctx->alloc(8 * (size_t)nw);
      o.validity = ov = o.own_validity->as<uint64_t>();
      o.null_count = -1;
    }
    bool_words_kernel<<<dim3((unsigned)ceil_div(nw, 256)), dim3(256), 0, ctx->stream>>>(
        l.v<uint64_t>(), l.stride, lv, l.stride, r.v<uint64_t>(), r.stride, rv, r.stride,
        op == SQLRS_EXPR_AND ? 6 : 7, nw, o.own_values->as<uint64_t>(), ov);
    SQ_HIP(hipGetLastError());
    return o;
  }
  fail(SQLRS_ERR_INTERNAL, "unsupported binary operator");
}

static DCol cast_col(Ctx *ctx, const DCol &cin, int32_t to, int64_t rows) {
  if (cin.dtype == to) return cin;
  DCol c = cin.stride == 0 ? materialize_scalar(ctx, cin, rows) : cin;
  DCol o;
  o.dtype = to;
  o.length = rows;
  o.validity = c.validity;
  o.own_validity = c.own_validity;
  o.null_count = c.null_count;
  size_t w = width_of(to);
  if (!w || (to != SQLRS_INT32 && to != SQLRS_INT64 && to != SQLRS_FLOAT64))
    fail(SQLRS_ERR_ARROW, "unsupported cast");
  int64_t n1 = std::max<int64_t>(rows, 1);
  o.own_values = ctx->alloc(w * (size_t)n1 + 16);
  o.values = o.own_values->p;
  if (rows == 0) return o;
  ProfScope ps(ctx, "expr_cast");
  int64_t n64 = (int64_t)round_up((size_t)rows, 64);
  dim3 g((unsigned)ceil_div(n64, BLOCK)), b(BLOCK);
  BufP okb;
  uint64_t *ok = nullptr;
  auto need_ok = [&]() {
    okb = ctx->alloc(bitmap_bytes(n1));
    ok = okb->as<uint64_t>();
  };
#define SQ_CAST(S, D) cast_kernel<S, D><<<g, b, 0, ctx->stream>>>(c.v<S>(), rows, o.own_values->as<D>(), ok)
  if (c.dtype == SQLRS_BOOLEAN) {
    if (to == SQLRS_INT32) cast_bool_kernel<int32_t><<<g, b, 0, ctx->stream>>>(c.v<uint64_t>(), rows, o.own_values->as<int32_t>());
    else if (to == SQLRS_INT64) cast_bool_kernel<int64_t><<<g, b, 0, ctx->stream>>>(c.v<uint64_t>(), rows, o.own_values->as<int64_t>());
    else cast_bool_kernel<double><<<g, b, 0, ctx->stream>>>(c.v<uint64_t>(), rows, o.own_values->as<double>());
  } else if (c.dtype == SQLRS_INT32 && to == SQLRS_INT64) SQ_CAST(int32_t, int64_t);
  else if (c.dtype == SQLRS_INT32 && to == SQLRS_FLOAT64) SQ_CAST(int32_t, double);
  else if (c.dtype == SQLRS_INT64 && to == SQLRS_FLOAT64) SQ_CAST(int64_t, double);
  else if (c.dtype == SQLRS_INT64 && to == SQLRS_INT32) { need_ok(); SQ_CAST(int64_t, int32_t); }
  else if (c.dtype == SQLRS_FLOAT64 && to == SQLRS_INT64) { need_ok(); SQ_CAST(double, int64_t); }
  else if (c.dtype == SQLRS_FLOAT64 && to == SQLRS_INT32) { need_ok(); SQ_CAST(double, int32_t); }
  else fail(SQLRS_ERR_ARROW, "unsupported cast");
#undef SQ_CAST
  SQ_HIP(hipGetLastError());
  if (ok) { // out-of-range values become NULL
    int64_t nw = ceil_div(n1, 64);
    if (c.validity && c.null_count != 0) {
      BufP v = ctx->alloc(8 * (size_t)nw);
      and_words_kernel<<<dim3((unsigned)ceil_div(nw, 256)), dim3(256), 0, ctx->stream>>>(
          c.validity, ok, nw, v->as<uint64_t>());
      SQ_HIP(hipGetLastError());
      o.own_validity = v;
    } else
      o.own_validity = okb;
    o.validity = o.own_validity->as<uint64_t>();
    o.null_count = -1;
  }
  return o;
}

DCol eval_expr(Ctx *ctx, const Expr &e, const std::function<const DCol &(int)> &col, int64_t rows,
               bool materialize) {
  std::vector<DCol> st;
  BufP err;
  for (size_t k = 0; k < e.nodes.size(); k++) {
    const sqlrs_expr_node_t &n = e.nodes[k];
    switch (n.op) {
    case SQLRS_EXPR_INPUT_REF: {
      const DCol &c = col(n.index);
      if (c.length != rows) fail(SQLRS_ERR_ARROW, "column length != num_rows");
      st.push_back(c);
      break;
    }
    case SQLRS_EXPR_CONSTANT:
      if (n.dtype == SQLRS_NULLTYPE) fail(SQLRS_ERR_INTERNAL, "Null-typed constant array");
      st.push_back(make_scalar(ctx, n, e.strings[k], rows));
      break;
    case SQLRS_EXPR_TYPE_CAST: {
      if (st.empty()) fail(SQLRS_ERR_INTERNAL, "malformed expression");
      DCol c = cast_col(ctx, st.back(), n.dtype, rows);
      st.back() = std::move(c);
      break;
    }
    default: {
      if (st.size() < 2) fail(SQLRS_ERR_INTERNAL, "malformed expression");
      DCol r = std::move(st.back());
      st.pop_back();
      DCol l = std::move(st.back());
      st.pop_back();
      if (l.stride == 0 && r.stride == 0) l = materialize_scalar(ctx, l, rows);
      st.push_back(binary_op(ctx, l, r, n.op, rows, err));
    }
    }
  }
  if (st.size() != 1) fail(SQLRS_ERR_INTERNAL, "malformed expression");
  if (err && ctx->fetch_value(err->as<int>())) fail(SQLRS_ERR_ARROW, "Divide by zero error");
  if (materialize && st[0].stride == 0) return materialize_scalar(ctx, st[0], rows);
  return std::move(st[0]);
}

} // namespace sq
