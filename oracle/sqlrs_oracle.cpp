/*
 * sqlrs_oracle.cpp — CPU restatement of the sqlrs v1 executor hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle and the timed CPU
 * baseline ("port") for the HIP backend.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; nothing under sqlrs_amd/ links,
 * imports or calls it, and the product library has no CPU fallback.
 *
 * It follows the reference algorithm step by step (single thread, hash-only
 * matching through std::unordered_map keyed by a 64-bit row hash, per-batch
 * per-group take + accumulate, first-seen group order), citing the reference
 * file:line each function restates.  What it does NOT reproduce is the bit
 * pattern of ahash 0.8.0 (a third-party crate absent from the reference tree,
 * Cargo.lock:6-8): hash values never appear in any operator output, so an own
 * 64-bit mixer is used.  Arrow-28 kernels used by the reference
 * (take/filter/concat/lexsort/sum/min/max/cast/cmp) are restated from their
 * documented semantics.
 *
 * Pinning: tests/test_oracle_golden.py checks this oracle against every golden
 * table the reference's own tests hold for this path (tests/golden/ (.json files),
 * transcribed from hash_join.rs:442-749, hash_agg.rs:213-220,
 * executor/mod.rs:271-395, tests/slt/{aggregation,join,join_filter,order,
 * filter,distinct}.slt).
 *
 * Build: make -C oracle   (g++ -O2 -shared; no dependencies)
 */
#include "../include/sqlrs_hip.h"

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

struct Err {
  int status;
  std::string msg;
};
[[noreturn]] void fail(int status, const std::string &m) { throw Err{status, m}; }

// ------------------------------------------------------------------ columns --
struct Col {
  int32_t dtype = SQLRS_NULLTYPE;
  int64_t length = 0;
  std::vector<uint8_t> values;   // fixed width data, bitmap (BOOLEAN) or bytes (UTF8)
  std::vector<uint8_t> validity; // empty = all valid
  std::vector<int32_t> offsets;  // UTF8
  bool valid(int64_t i) const {
    return validity.empty() || ((validity[i >> 3] >> (i & 7)) & 1);
  }
  int64_t null_count() const {
    if (validity.empty()) return 0;
    int64_t n = 0;
    for (int64_t i = 0; i < length; i++) n += !valid(i);
    return n;
  }
};

struct Batch {
  int64_t rows = 0;
  std::vector<Col> cols;
};

size_t width_of(int32_t dtype) {
  switch (dtype) {
  case SQLRS_INT32:
  case SQLRS_UINT32:
    return 4;
  case SQLRS_INT64:
  case SQLRS_UINT64:
  case SQLRS_FLOAT64:
    return 8;
  default:
    return 0;
  }
}

template <class T> T getv(const Col &c, int64_t i) {
  T v;
  std::memcpy(&v, c.values.data() + i * sizeof(T), sizeof(T));
  return v;
}
bool getbool(const Col &c, int64_t i) { return (c.values[i >> 3] >> (i & 7)) & 1; }
std::string getstr(const Col &c, int64_t i) {
  return std::string((const char *)c.values.data() + c.offsets[i],
                     (size_t)(c.offsets[i + 1] - c.offsets[i]));
}

// ScalarValue  [ref: src/types/mod.rs:23-36]
struct Scalar {
  int32_t dtype = SQLRS_NULLTYPE;
  bool null = true;
  int64_t i = 0;
  double f = 0;
  std::string s;
  // Hash/Eq of the reference's ScalarValue: floats by bit pattern (types/mod.rs derives on
  // ordered bits), everything else by value; None is a value of its own.
  bool operator<(const Scalar &o) const {
    if (dtype != o.dtype) return dtype < o.dtype;
    if (null != o.null) return null;
    if (null) return false;
    switch (dtype) {
    case SQLRS_FLOAT64: {
      uint64_t a, b;
      std::memcpy(&a, &f, 8);
      std::memcpy(&b, &o.f, 8);
      return a < b;
    }
    case SQLRS_UTF8:
      return s < o.s;
    default:
      return i < o.i;
    }
  }
};

// ScalarValue::try_from_array  [ref: src/types/mod.rs:63-78]
Scalar scalar_at(const Col &c, int64_t i) {
  Scalar s;
  s.dtype = c.dtype;
  s.null = !c.valid(i);
  if (s.null) return s;
  switch (c.dtype) {
  case SQLRS_INT32:
    s.i = getv<int32_t>(c, i);
    break;
  case SQLRS_INT64:
    s.i = getv<int64_t>(c, i);
    break;
  case SQLRS_UINT32:
    s.i = getv<uint32_t>(c, i);
    break;
  case SQLRS_UINT64:
    s.i = (int64_t)getv<uint64_t>(c, i);
    break;
  case SQLRS_FLOAT64:
    s.f = getv<double>(c, i);
    break;
  case SQLRS_BOOLEAN:
    s.i = getbool(c, i);
    break;
  case SQLRS_UTF8:
    s.s = getstr(c, i);
    break;
  default:
    fail(SQLRS_ERR_INTERNAL, "unsupported scalar type");
  }
  return s;
}

// Array builders  [ref: src/types/mod.rs:225-273 build_scalar_value_builder /
// append_scalar_value_for_builder]
struct Builder {
  Col c;
  bool any_null = false;
  std::vector<uint8_t> valid_bits; // one byte per row while building
  explicit Builder(int32_t dtype) {
    c.dtype = dtype;
    if (dtype == SQLRS_NULLTYPE)
      fail(SQLRS_ERR_INTERNAL, "Null-typed builder is not supported"); // types/mod.rs:241-245
    if (dtype == SQLRS_UTF8) c.offsets.push_back(0);
  }
  void append_null() {
    any_null = true;
    valid_bits.push_back(0);
    push_default();
  }
  void push_default() {
    switch (c.dtype) {
    case SQLRS_BOOLEAN:
      push_bit(false);
      break;
    case SQLRS_UTF8:
      c.offsets.push_back(c.offsets.back());
      break;
    default:
      c.values.resize(c.values.size() + width_of(c.dtype), 0);
    }
    c.length++;
  }
  void push_bit(bool b) {
    if ((c.length & 7) == 0) c.values.push_back(0);
    if (b) c.values[c.length >> 3] |= (uint8_t)(1u << (c.length & 7));
  }
  void append(const Scalar &s) {
    if (s.null) {
      append_null();
      return;
    }
    valid_bits.push_back(1);
    switch (c.dtype) {
    case SQLRS_INT32: {
      int32_t v = (int32_t)s.i;
      put(&v, 4);
      break;
    }
    case SQLRS_UINT32: {
      uint32_t v = (uint32_t)s.i;
      put(&v, 4);
      break;
    }
    case SQLRS_INT64:
    case SQLRS_UINT64:
      put(&s.i, 8);
      break;
    case SQLRS_FLOAT64:
      put(&s.f, 8);
      break;
    case SQLRS_BOOLEAN:
      push_bit(s.i != 0);
      break;
    case SQLRS_UTF8:
      c.values.insert(c.values.end(), s.s.begin(), s.s.end());
      c.offsets.push_back((int32_t)c.values.size());
      break;
    default:
      fail(SQLRS_ERR_INTERNAL, "unsupported builder type");
    }
    c.length++;
  }
  void put(const void *p, size_t n) {
    const uint8_t *b = (const uint8_t *)p;
    c.values.insert(c.values.end(), b, b + n);
  }
  Col finish() {
    if (any_null) {
      c.validity.assign((size_t)((c.length + 7) / 8), 0);
      for (int64_t i = 0; i < c.length; i++)
        if (valid_bits[i]) c.validity[i >> 3] |= (uint8_t)(1u << (i & 7));
    }
    return std::move(c);
  }
};

Col col_from_abi(const sqlrs_column_t &a) {
  if (a.mem != SQLRS_MEM_HOST) fail(SQLRS_ERR_ARROW, "oracle accepts host memory only");
  Col c;
  c.dtype = a.dtype;
  c.length = a.length;
  size_t nb = (size_t)((a.length + 7) / 8);
  if (a.validity && a.null_count != 0) c.validity.assign(a.validity, a.validity + nb);
  // normalise: a validity bitmap with every bit set is dropped
  if (!c.validity.empty() && c.null_count() == 0) c.validity.clear();
  switch (a.dtype) {
  case SQLRS_BOOLEAN:
    c.values.assign((const uint8_t *)a.values, (const uint8_t *)a.values + nb);
    break;
  case SQLRS_UTF8: {
    if (!a.offsets) fail(SQLRS_ERR_ARROW, "utf8 column without offsets");
    c.offsets.assign(a.offsets, a.offsets + a.length + 1);
    c.values.assign((const uint8_t *)a.values, (const uint8_t *)a.values + c.offsets.back());
    break;
  }
  default: {
    size_t w = width_of(a.dtype);
    if (!w) fail(SQLRS_ERR_INTERNAL, "unsupported column dtype");
    c.values.assign((const uint8_t *)a.values, (const uint8_t *)a.values + w * (size_t)a.length);
  }
  }
  return c;
}

Batch batch_from_abi(const sqlrs_batch_t *b) {
  if (!b) fail(SQLRS_ERR_ARROW, "null batch");
  Batch out;
  out.rows = b->num_rows;
  for (int i = 0; i < b->num_columns; i++) {
    if (b->columns[i].length != b->num_rows) fail(SQLRS_ERR_ARROW, "column length != num_rows");
    out.cols.push_back(col_from_abi(b->columns[i]));
  }
  return out;
}

struct OwnedBatch {
  sqlrs_batch_t abi;
  Batch data;
  std::vector<sqlrs_column_t> descs;
};

sqlrs_batch_t *batch_to_abi(Batch &&b) {
  auto *o = new OwnedBatch();
  o->data = std::move(b);
  o->descs.resize(o->data.cols.size());
  for (size_t i = 0; i < o->data.cols.size(); i++) {
    Col &c = o->data.cols[i];
    sqlrs_column_t &d = o->descs[i];
    d.dtype = c.dtype;
    d.mem = SQLRS_MEM_HOST;
    d.length = c.length;
    d.null_count = c.null_count();
    // keep non-null pointers even for empty columns
    if (c.values.empty()) c.values.reserve(8);
    d.values = c.values.data();
    d.validity = c.validity.empty() ? nullptr : c.validity.data();
    d.offsets = c.dtype == SQLRS_UTF8 ? c.offsets.data() : nullptr;
  }
  o->abi.num_rows = o->data.rows;
  o->abi.num_columns = (int32_t)o->descs.size();
  o->abi.reserved = 0;
  o->abi.columns = o->descs.data();
  o->abi.owner = o;
  return &o->abi;
}

// ---------------------------------------------------- arrow kernels restated --
// arrow::compute::take with nullable indices (NULL index => NULL row)
// [ref call sites: hash_join.rs:35,40,305; hash_agg.rs:118; order.rs:53]
Col take(const Col &src, const std::vector<int64_t> &idx /* -1 = NULL */) {
  size_t w = width_of(src.dtype);
  if (w && src.validity.empty()) { // fixed-width, no source NULLs: plain gather loop
    Col out;
    out.dtype = src.dtype;
    out.length = (int64_t)idx.size();
    out.values.resize(w * idx.size());
    bool any_null = false;
    for (size_t k = 0; k < idx.size(); k++) {
      int64_t i = idx[k];
      if (i < 0) {
        any_null = true;
        std::memset(out.values.data() + k * w, 0, w);
        continue;
      }
      if (i >= src.length) fail(SQLRS_ERR_ARROW, "take index out of bounds");
      std::memcpy(out.values.data() + k * w, src.values.data() + (size_t)i * w, w);
    }
    if (any_null) {
      out.validity.assign((idx.size() + 7) / 8, 0);
      for (size_t k = 0; k < idx.size(); k++)
        if (idx[k] >= 0) out.validity[k >> 3] |= (uint8_t)(1u << (k & 7));
    }
    return out;
  }
  Builder b(src.dtype == SQLRS_NULLTYPE ? SQLRS_INT32 : src.dtype);
  for (int64_t i : idx) {
    if (i < 0)
      b.append_null();
    else {
      if (i >= src.length) fail(SQLRS_ERR_ARROW, "take index out of bounds");
      b.append(scalar_at(src, i));
    }
  }
  return b.finish();
}

// arrow::compute::concat_batches  [ref: hash_join.rs:187; order.rs:28]
Batch concat_batches(const std::vector<Batch> &bs) {
  Batch out;
  if (bs.empty()) return out;
  size_t nc = bs[0].cols.size();
  for (size_t c = 0; c < nc; c++) {
    Builder b(bs[0].cols[c].dtype);
    for (const Batch &x : bs) {
      if (x.cols.size() != nc || x.cols[c].dtype != bs[0].cols[c].dtype)
        fail(SQLRS_ERR_ARROW, "concat_batches: schema mismatch");
      for (int64_t i = 0; i < x.rows; i++) b.append(scalar_at(x.cols[c], i));
    }
    out.cols.push_back(b.finish());
  }
  for (const Batch &x : bs) out.rows += x.rows;
  return out;
}

// new_null_array  [ref: hash_join.rs:316]
Col null_array(int32_t dtype, int64_t n) {
  Builder b(dtype);
  for (int64_t i = 0; i < n; i++) b.append_null();
  return b.finish();
}

// ScalarValue -> N-long array  [ref: src/types/mod.rs:214-223 build_scalar_value_array,
// called from evaluator.rs:21]
Col constant_array(const sqlrs_expr_node_t &n, int64_t rows) {
  Scalar s;
  s.dtype = n.dtype;
  s.null = n.is_null != 0;
  s.i = n.i;
  s.f = n.f;
  if (n.dtype == SQLRS_UTF8 && n.s) s.s = n.s;
  if (n.dtype == SQLRS_NULLTYPE) fail(SQLRS_ERR_INTERNAL, "Null-typed constant array");
  Builder b(n.dtype);
  for (int64_t i = 0; i < rows; i++) b.append(s);
  return b.finish();
}

// total order on f64 bit patterns (ArrowNativeTypeOp::compare for floats, arrow-array 28)
int cmp_f64(double a, double b) {
  int64_t x, y;
  std::memcpy(&x, &a, 8);
  std::memcpy(&y, &b, 8);
  x ^= (int64_t)(((uint64_t)(x >> 63)) >> 1);
  y ^= (int64_t)(((uint64_t)(y >> 63)) >> 1);
  return x < y ? -1 : (x > y ? 1 : 0);
}

int cmp_scalar(const Scalar &a, const Scalar &b) {
  switch (a.dtype) {
  case SQLRS_FLOAT64:
    return cmp_f64(a.f, b.f);
  case SQLRS_UTF8:
    return a.s < b.s ? -1 : (a.s > b.s ? 1 : 0);
  case SQLRS_UINT64:
    return (uint64_t)a.i < (uint64_t)b.i ? -1 : ((uint64_t)a.i > (uint64_t)b.i ? 1 : 0);
  default:
    return a.i < b.i ? -1 : (a.i > b.i ? 1 : 0);
  }
}

// arrow::compute::cast for the numeric lattice  [ref: evaluator.rs:23; sum.rs:54]
Col cast(const Col &in, int32_t to) {
  if (in.dtype == to) return in;
  Builder b(to);
  for (int64_t i = 0; i < in.length; i++) {
    Scalar s = scalar_at(in, i);
    if (s.null) {
      b.append_null();
      continue;
    }
    Scalar o;
    o.dtype = to;
    o.null = false;
    bool from_f = in.dtype == SQLRS_FLOAT64;
    bool from_i = in.dtype == SQLRS_INT32 || in.dtype == SQLRS_INT64 || in.dtype == SQLRS_BOOLEAN;
    if (to == SQLRS_FLOAT64 && from_i)
      o.f = (double)s.i;
    else if ((to == SQLRS_INT64 || to == SQLRS_INT32) && from_i) {
      if (to == SQLRS_INT32 && (s.i > INT32_MAX || s.i < INT32_MIN)) {
        b.append_null(); // arrow safe cast: out-of-range -> NULL
        continue;
      }
      o.i = s.i;
    } else if ((to == SQLRS_INT64 || to == SQLRS_INT32) && from_f) {
      double lim = to == SQLRS_INT32 ? 2147483648.0 : 9223372036854775808.0;
      if (!(s.f > -lim - 1 && s.f < lim)) {
        b.append_null();
        continue;
      }
      o.i = (int64_t)s.f;
    } else
      fail(SQLRS_ERR_ARROW, "unsupported cast");
    b.append(o);
  }
  return b.finish();
}

// binary_op  [ref: src/executor/array_compute.rs:70-90]
Col binary_op(const Col &l, const Col &r, int op) {
  if (l.length != r.length) fail(SQLRS_ERR_ARROW, "binary op on arrays of different length");
  int64_t n = l.length;
  if (op >= SQLRS_EXPR_PLUS && op <= SQLRS_EXPR_DIVIDE) {
    // arithmetic_op!: dispatch on LEFT dtype, both sides downcast to it (array_compute.rs:37-46)
    if (l.dtype != r.dtype) fail(SQLRS_ERR_INTERNAL, "compute_op failed to downcast array");
    if (l.dtype != SQLRS_INT32 && l.dtype != SQLRS_INT64 && l.dtype != SQLRS_FLOAT64)
      fail(SQLRS_ERR_INTERNAL, "unsupported data type");
    Builder b(l.dtype);
    for (int64_t i = 0; i < n; i++) {
      if (!l.valid(i) || !r.valid(i)) {
        b.append_null();
        continue;
      }
      Scalar a = scalar_at(l, i), c = scalar_at(r, i), o;
      o.dtype = l.dtype;
      o.null = false;
      if (l.dtype == SQLRS_FLOAT64) {
        switch (op) {
        case SQLRS_EXPR_PLUS:
          o.f = a.f + c.f;
          break;
        case SQLRS_EXPR_MINUS:
          o.f = a.f - c.f;
          break;
        case SQLRS_EXPR_MULTIPLY:
          o.f = a.f * c.f;
          break;
        default:
          if (c.f == 0.0) fail(SQLRS_ERR_ARROW, "Divide by zero error");
          o.f = a.f / c.f;
        }
      } else {
        // arrow-arith 28 add/subtract/multiply on integers wrap
        uint64_t x = (uint64_t)a.i, y = (uint64_t)c.i, z;
        switch (op) {
        case SQLRS_EXPR_PLUS:
          z = x + y;
          break;
        case SQLRS_EXPR_MINUS:
          z = x - y;
          break;
        case SQLRS_EXPR_MULTIPLY:
          z = x * y;
          break;
        default:
          if (c.i == 0) fail(SQLRS_ERR_ARROW, "Divide by zero error");
          if (l.dtype == SQLRS_INT64 && a.i == INT64_MIN && c.i == -1)
            z = (uint64_t)INT64_MIN;
          else
            z = (uint64_t)(a.i / c.i);
        }
        o.i = l.dtype == SQLRS_INT32 ? (int64_t)(int32_t)(uint32_t)z : (int64_t)z;
      }
      b.append(o);
    }
    return b.finish();
  }
  if (op >= SQLRS_EXPR_GT && op <= SQLRS_EXPR_NOTEQ) {
    if (l.dtype != r.dtype) fail(SQLRS_ERR_ARROW, "comparison of arrays of different types");
    Builder b(SQLRS_BOOLEAN);
    for (int64_t i = 0; i < n; i++) {
      if (!l.valid(i) || !r.valid(i)) {
        b.append_null();
        continue;
      }
      int c = cmp_scalar(scalar_at(l, i), scalar_at(r, i));
      bool v = false;
      switch (op) {
      case SQLRS_EXPR_GT:
        v = c > 0;
        break;
      case SQLRS_EXPR_LT:
        v = c < 0;
        break;
      case SQLRS_EXPR_GTEQ:
        v = c >= 0;
        break;
      case SQLRS_EXPR_LTEQ:
        v = c <= 0;
        break;
      case SQLRS_EXPR_EQ:
        v = c == 0;
        break;
      default:
        v = c != 0;
      }
      Scalar o;
      o.dtype = SQLRS_BOOLEAN;
      o.null = false;
      o.i = v;
      b.append(o);
    }
    return b.finish();
  }
  if (op == SQLRS_EXPR_AND || op == SQLRS_EXPR_OR) {
    // boolean_op!: both must be Boolean (array_compute.rs:50-57); Kleene logic
    if (l.dtype != SQLRS_BOOLEAN || r.dtype != SQLRS_BOOLEAN)
      fail(SQLRS_ERR_INTERNAL,
           "Cannot evaluate binary expression with non-Boolean types, only Boolean supported");
    Builder b(SQLRS_BOOLEAN);
    for (int64_t i = 0; i < n; i++) {
      bool lv = l.valid(i), rv = r.valid(i);
      bool a = lv && getbool(l, i), c = rv && getbool(r, i);
      Scalar o;
      o.dtype = SQLRS_BOOLEAN;
      if (op == SQLRS_EXPR_AND) {
        if ((lv && !a) || (rv && !c)) {
          o.null = false;
          o.i = 0;
        } else if (lv && rv) {
          o.null = false;
          o.i = 1;
        } else
          o.null = true;
      } else {
        if ((lv && a) || (rv && c)) {
          o.null = false;
          o.i = 1;
        } else if (lv && rv) {
          o.null = false;
          o.i = 0;
        } else
          o.null = true;
      }
      b.append(o);
    }
    return b.finish();
  }
  fail(SQLRS_ERR_INTERNAL, "unsupported binary operator");
}

// BoundExpr::eval_column  [ref: src/executor/evaluator.rs:13-28] over the postfix encoding
struct Expr {
  std::vector<sqlrs_expr_node_t> nodes;
  std::vector<std::string> strings; // owned copies of utf8 constants
};
Expr expr_from_abi(const sqlrs_expr_t *e) {
  Expr x;
  if (!e || e->num_nodes <= 0 || !e->nodes) fail(SQLRS_ERR_INTERNAL, "empty expression");
  x.nodes.assign(e->nodes, e->nodes + e->num_nodes);
  x.strings.resize(x.nodes.size());
  for (size_t i = 0; i < x.nodes.size(); i++)
    if (x.nodes[i].s) x.strings[i] = x.nodes[i].s;
  return x;
}
Col eval_column(const Expr &e, const Batch &batch) {
  std::vector<Col> st;
  for (size_t k = 0; k < e.nodes.size(); k++) {
    sqlrs_expr_node_t n = e.nodes[k];
    n.s = e.strings[k].c_str();
    switch (n.op) {
    case SQLRS_EXPR_INPUT_REF:
      if (n.index < 0 || (size_t)n.index >= batch.cols.size())
        fail(SQLRS_ERR_INTERNAL, "input ref out of range");
      st.push_back(batch.cols[n.index]);
      break;
    case SQLRS_EXPR_CONSTANT:
      st.push_back(constant_array(n, batch.rows));
      break;
    case SQLRS_EXPR_TYPE_CAST: {
      if (st.empty()) fail(SQLRS_ERR_INTERNAL, "malformed expression");
      Col c = cast(st.back(), n.dtype);
      st.back() = std::move(c);
      break;
    }
    default: {
      if (st.size() < 2) fail(SQLRS_ERR_INTERNAL, "malformed expression");
      Col r = std::move(st.back());
      st.pop_back();
      Col l = std::move(st.back());
      st.pop_back();
      st.push_back(binary_op(l, r, n.op));
    }
    }
  }
  if (st.size() != 1) fail(SQLRS_ERR_INTERNAL, "malformed expression");
  return std::move(st[0]);
}

// arrow::compute::filter_record_batch: keep rows whose predicate is valid AND true
// [ref: filter.rs:23; hash_join.rs:66,69,84,87]
std::vector<int64_t> true_rows(const Col &mask) {
  if (mask.dtype != SQLRS_BOOLEAN)
    fail(SQLRS_ERR_INTERNAL, "filter executor expected evaluate boolean array");
  std::vector<int64_t> keep;
  for (int64_t i = 0; i < mask.length; i++)
    if (mask.valid(i) && getbool(mask, i)) keep.push_back(i);
  return keep;
}

// ------------------------------------------------------------- create_hashes --
uint64_t mix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}
// stands in for RandomState::with_seeds(0,0,0,0).hash_one(v): values are hashed at their
// native width, so an Int32 5 and an Int64 5 hash differently (hash_utils.rs:174-179)
uint64_t hash_one(const Col &c, int64_t i) {
  switch (c.dtype) {
  case SQLRS_INT32:
    return mix64((uint64_t)(uint32_t)getv<int32_t>(c, i) ^ 0x3232323200000000ULL);
  case SQLRS_INT64:
    return mix64((uint64_t)getv<int64_t>(c, i) + 0x9e3779b97f4a7c15ULL);
  case SQLRS_FLOAT64: // by bit pattern, as u64 (hash_utils.rs:124-131)
    return mix64(getv<uint64_t>(c, i) + 0x9e3779b97f4a7c15ULL);
  case SQLRS_BOOLEAN:
    return mix64((uint64_t)getbool(c, i) ^ 0x0808080808080808ULL);
  case SQLRS_UTF8: {
    uint64_t h = 0xcbf29ce484222325ULL;
    for (int32_t k = c.offsets[i]; k < c.offsets[i + 1]; k++) {
      h ^= c.values[k];
      h *= 0x100000001b3ULL;
    }
    return mix64(h ^ 0x7575757575757575ULL);
  }
  default:
    fail(SQLRS_ERR_INTERNAL, "Unsupported data type in hasher"); // hash_utils.rs:210-216
  }
}
// [ref: hash_utils.rs:13-16]
uint64_t combine_hashes(uint64_t l, uint64_t r) {
  uint64_t h = (uint64_t)(17 * 37) + l;
  return h * 37 + r;
}
// [ref: hash_utils.rs:161-220]; NULL slots leave the incoming hash untouched (:91-104)
void create_hashes(const std::vector<Col> &arrays, std::vector<uint64_t> &hashes) {
  bool multi_col = arrays.size() > 1;
  for (const Col &col : arrays) {
    for (size_t i = 0; i < hashes.size(); i++) {
      if (!col.valid((int64_t)i)) continue;
      uint64_t h = hash_one(col, (int64_t)i);
      hashes[i] = multi_col ? combine_hashes(h, hashes[i]) : h;
    }
  }
}

} // namespace

// =========================================================================== //
//                                 operators                                   //
// =========================================================================== //
struct oracle_ctx {
  std::string last_error;
  int compat_count_last_batch = 0; // count.rs:22 assigns instead of accumulating
};

namespace {

// ---- accumulators  [ref: src/executor/aggregate/mod.rs:19-49] -------------- //
struct Accumulator {
  virtual ~Accumulator() {}
  virtual void update_batch(const Col &array) = 0;
  virtual Scalar evaluate() const = 0;
};
// [ref: aggregate/count.rs:10-29]
struct CountAccumulator : Accumulator {
  int64_t result = 0;
  bool compat;
  explicit CountAccumulator(bool c) : compat(c) {}
  void update_batch(const Col &a) override {
    int64_t n = a.length - a.null_count();
    if (compat)
      result = n; // the reference's assignment (count.rs:22)
    else
      result += n; // SQL semantics; identical for single-batch input
  }
  Scalar evaluate() const override {
    Scalar s;
    s.dtype = SQLRS_INT64;
    s.null = false;
    s.i = result;
    return s;
  }
};
// [ref: aggregate/count.rs:31-58] — a NULL is a distinct value of the HashSet<ScalarValue>
struct DistinctCountAccumulator : Accumulator {
  std::set<Scalar> vals;
  void update_batch(const Col &a) override {
    for (int64_t i = 0; i < a.length; i++) vals.insert(scalar_at(a, i));
  }
  Scalar evaluate() const override {
    Scalar s;
    s.dtype = SQLRS_INT64;
    s.null = false;
    s.i = (int64_t)vals.size();
    return s;
  }
};
// sum_result  [ref: aggregate/sum.rs:25-34,64-85]
Scalar sum_result(const Scalar &l, const Scalar &r) {
  Scalar o;
  o.dtype = l.dtype;
  if (l.dtype == SQLRS_FLOAT64) {
    if (r.dtype != SQLRS_FLOAT64 && r.dtype != SQLRS_INT64 && r.dtype != SQLRS_INT32)
      fail(SQLRS_ERR_INTERNAL, "not expected types for sum");
    double rv = r.dtype == SQLRS_FLOAT64 ? r.f : (double)r.i;
    if (l.null && r.null)
      o.null = true;
    else {
      o.null = false;
      o.f = l.null ? rv : (r.null ? l.f : l.f + rv);
    }
    return o;
  }
  if (l.dtype == SQLRS_INT64 && (r.dtype == SQLRS_INT64 || r.dtype == SQLRS_INT32)) {
    if (l.null && r.null)
      o.null = true;
    else {
      o.null = false;
      o.i = l.null ? r.i : (r.null ? l.i : (int64_t)((uint64_t)l.i + (uint64_t)r.i));
    }
    return o;
  }
  // (Int32, Int32) is not in the reference's match (sum.rs:64-85): sum(int32) with
  // return_type Int32 hits unimplemented!().  Restated as an internal error.
  fail(SQLRS_ERR_INTERNAL, "not expected types for sum");
}
// [ref: aggregate/sum.rs:36-97]
struct SumAccumulator : Accumulator {
  Scalar result;
  int32_t data_type;
  explicit SumAccumulator(int32_t dt) : data_type(dt) {
    result.dtype = dt;
    result.null = true;
  }
  void update_batch(const Col &array) override {
    Col v = cast(array, data_type); // sum.rs:54
    // arrow::compute::sum: None when there is no valid value, NULLs skipped
    Scalar delta;
    delta.dtype = v.dtype;
    delta.null = true;
    for (int64_t i = 0; i < v.length; i++) {
      if (!v.valid(i)) continue;
      Scalar x = scalar_at(v, i);
      if (delta.null) {
        delta = x;
      } else if (v.dtype == SQLRS_FLOAT64)
        delta.f += x.f;
      else
        delta.i = (int64_t)((uint64_t)delta.i + (uint64_t)x.i);
    }
    if (v.dtype == SQLRS_INT32) delta.i = (int64_t)(int32_t)(uint32_t)delta.i;
    if (v.dtype != SQLRS_INT32 && v.dtype != SQLRS_INT64 && v.dtype != SQLRS_FLOAT64)
      fail(SQLRS_ERR_INTERNAL, "unsupported sum type"); // sum.rs:59
    result = sum_result(result, delta);
  }
  Scalar evaluate() const override { return result; }
};
// [ref: aggregate/sum.rs:99-132]
struct DistinctSumAccumulator : Accumulator {
  std::set<Scalar> vals;
  int32_t data_type;
  explicit DistinctSumAccumulator(int32_t dt) : data_type(dt) {}
  void update_batch(const Col &a) override {
    for (int64_t i = 0; i < a.length; i++) vals.insert(scalar_at(a, i));
  }
  Scalar evaluate() const override {
    Scalar sum;
    sum.dtype = data_type;
    sum.null = true;
    for (const Scalar &v : vals) sum = sum_result(sum, v);
    return sum;
  }
};
// [ref: aggregate/min_max.rs:47-157]
struct MinMaxAccumulator : Accumulator {
  Scalar cur;
  bool is_min;
  MinMaxAccumulator(int32_t dt, bool m) : is_min(m) {
    cur.dtype = dt;
    cur.null = true;
  }
  void update_batch(const Col &a) override {
    if (a.dtype != SQLRS_FLOAT64 && a.dtype != SQLRS_INT32 && a.dtype != SQLRS_INT64 &&
        a.dtype != SQLRS_UTF8)
      fail(SQLRS_ERR_INTERNAL, "unsupported min/max type"); // min_max.rs:41
    Scalar delta;
    delta.dtype = a.dtype;
    delta.null = true;
    for (int64_t i = 0; i < a.length; i++) {
      if (!a.valid(i)) continue;
      Scalar x = scalar_at(a, i);
      if (delta.null)
        delta = x;
      else {
        int c = cmp_scalar(x, delta);
        if (is_min ? c < 0 : c > 0) delta = x;
      }
    }
    if (cur.dtype != delta.dtype) fail(SQLRS_ERR_INTERNAL, "unsupported min_max scalar type");
    if (delta.null) return;
    if (cur.null)
      cur = delta;
    else {
      int c = cmp_scalar(delta, cur);
      if (is_min ? c < 0 : c > 0) cur = delta;
    }
  }
  Scalar evaluate() const override { return cur; }
};

struct AggSpec {
  int32_t func, distinct, return_dtype;
  Expr arg;
};
// create_accumulator  [ref: aggregate/mod.rs:27-45]
std::unique_ptr<Accumulator> create_accumulator(const AggSpec &a, bool compat_count) {
  switch (a.func) {
  case SQLRS_AGG_COUNT:
    if (a.distinct) return std::unique_ptr<Accumulator>(new DistinctCountAccumulator());
    return std::unique_ptr<Accumulator>(new CountAccumulator(compat_count));
  case SQLRS_AGG_SUM:
    if (a.distinct) return std::unique_ptr<Accumulator>(new DistinctSumAccumulator(a.return_dtype));
    return std::unique_ptr<Accumulator>(new SumAccumulator(a.return_dtype));
  case SQLRS_AGG_MIN:
    return std::unique_ptr<Accumulator>(new MinMaxAccumulator(a.return_dtype, true));
  case SQLRS_AGG_MAX:
    return std::unique_ptr<Accumulator>(new MinMaxAccumulator(a.return_dtype, false));
  }
  fail(SQLRS_ERR_INTERNAL, "unknown aggregate function");
}

} // namespace

// ---- FilterExecutor  [ref: src/executor/filter.rs:7-25] -------------------- //
struct oracle_filter {
  oracle_ctx *ctx;
  Expr expr;
};

// ---- HashJoinExecutor  [ref: src/executor/join/hash_join.rs:16-323] -------- //
struct oracle_hash_join {
  oracle_ctx *ctx;
  int join_type;
  std::vector<Expr> on_left_keys, on_right_keys;
  bool has_filter = false;
  Expr filter;
  std::vector<int32_t> right_dtypes;
  // build state (hash_join.rs:155-159)
  std::unordered_map<uint64_t, std::vector<size_t>> left_hashmap;
  size_t left_row_offset = 0;
  std::vector<Batch> left_batches;
  bool build_finished = false;
  Batch left_single_batch;
  std::vector<uint8_t> visited_left_side; // BooleanBufferBuilder (:194-206)
};

// ---- HashAggExecutor  [ref: src/executor/aggregate/hash_agg.rs:15-150] ----- //
struct oracle_hash_agg {
  oracle_ctx *ctx;
  std::vector<AggSpec> agg_funcs;
  std::vector<Expr> group_by;
  bool saw_batch = false;
  std::vector<int32_t> field_dtypes; // group_and_agg_fields (:38,47-59)
  std::vector<uint64_t> group_hashs; // first-seen order (:39,98)
  std::unordered_map<uint64_t, std::vector<Scalar>> group_hash_2_keys;
  std::unordered_map<uint64_t, std::vector<std::unique_ptr<Accumulator>>> group_hash_2_accs;
};

// ---- OrderExecutor  [ref: src/executor/order.rs:8-67] ---------------------- //
struct oracle_order {
  oracle_ctx *ctx;
  std::vector<Expr> exprs;
  std::vector<int> asc;
  std::vector<Batch> batches;
};

namespace {

// build_batch  [ref: hash_join.rs:25-45]
Batch build_batch(const Batch &left, const Batch &right, const std::vector<int64_t> &li,
                  const std::vector<int64_t> &ri) {
  Batch out;
  for (const Col &c : left.cols) out.cols.push_back(take(c, li));
  for (const Col &c : right.cols) out.cols.push_back(take(c, ri));
  out.rows = (int64_t)li.size();
  return out;
}

// the index-building loop of the probe phase  [ref: hash_join.rs:217-253]
void probe_indices(oracle_hash_join *j, const Batch &batch, std::vector<int64_t> &left_indices,
                   std::vector<int64_t> &right_indices) {
  std::vector<Col> right_keys;
  for (const Expr &e : j->on_right_keys) right_keys.push_back(eval_column(e, batch));
  std::vector<uint64_t> right_rows_hashes((size_t)batch.rows, 0);
  create_hashes(right_keys, right_rows_hashes);
  bool outer_right = j->join_type == SQLRS_JOIN_RIGHT || j->join_type == SQLRS_JOIN_FULL;
  for (size_t row = 0; row < right_rows_hashes.size(); row++) {
    auto it = j->left_hashmap.find(right_rows_hashes[row]);
    if (it != j->left_hashmap.end()) {
      for (size_t i : it->second) { // no key-equality check (TODO at :222-224)
        left_indices.push_back((int64_t)i);
        right_indices.push_back((int64_t)row);
      }
    } else if (outer_right) {
      left_indices.push_back(-1); // append_null (:243)
      right_indices.push_back((int64_t)row);
    }
  }
}

// apply_join_filter  [ref: hash_join.rs:47-127]
void apply_join_filter(oracle_hash_join *j, const Batch &intermediate,
                       std::vector<int64_t> &left_indices, std::vector<int64_t> &right_indices,
                       int64_t right_num_rows) {
  if (!j->has_filter) return;
  Col mask = eval_column(j->filter, intermediate);
  std::vector<int64_t> keep = true_rows(mask);
  std::vector<int64_t> l, r;
  for (int64_t k : keep) {
    l.push_back(left_indices[k]);
    r.push_back(right_indices[k]);
  }
  if (j->join_type == SQLRS_JOIN_RIGHT || j->join_type == SQLRS_JOIN_FULL) {
    // keep every right row: re-append the ones that lost all their matches (:73-121)
    std::vector<uint8_t> visited_right((size_t)right_num_rows, 0);
    for (int64_t x : r) visited_right[(size_t)x] = 1;
    for (int64_t v = 0; v < right_num_rows; v++)
      if (!visited_right[(size_t)v]) {
        l.push_back(-1);
        r.push_back(v);
      }
  }
  left_indices.swap(l);
  right_indices.swap(r);
}

void require_build_finished(oracle_hash_join *j) {
  if (!j->build_finished) fail(SQLRS_ERR_INTERNAL, "probe before build_finish");
}

Batch join_probe(oracle_hash_join *j, const Batch &batch) {
  std::vector<int64_t> li, ri;
  probe_indices(j, batch, li, ri);
  // 2. intermediate batch from all left and right columns (:256-262)
  Batch intermediate = build_batch(j->left_single_batch, batch, li, ri);
  // 3. join filter (:265-272)
  apply_join_filter(j, intermediate, li, ri, batch.rows);
  if (j->join_type == SQLRS_JOIN_LEFT || j->join_type == SQLRS_JOIN_FULL)
    for (int64_t x : li)
      if (x >= 0) j->visited_left_side[(size_t)x] = 1; // :274-282
  return build_batch(j->left_single_batch, batch, li, ri); // :284-291
}

template <class F> int guard(oracle_ctx *ctx, F &&f) {
  try {
    f();
    return SQLRS_OK;
  } catch (const Err &e) {
    if (ctx) ctx->last_error = e.msg;
    return e.status;
  } catch (const std::exception &e) {
    if (ctx) ctx->last_error = e.what();
    return SQLRS_ERR_INTERNAL;
  }
}

} // namespace

// =========================================================================== //
//                     C entry points (mirror include/sqlrs_hip.h)             //
// =========================================================================== //
extern "C" {

int oracle_ctx_create(int compat_count_last_batch, oracle_ctx **out) {
  *out = new oracle_ctx();
  (*out)->compat_count_last_batch = compat_count_last_batch;
  return SQLRS_OK;
}
void oracle_ctx_destroy(oracle_ctx *ctx) { delete ctx; }
const char *oracle_last_error(const oracle_ctx *ctx) { return ctx->last_error.c_str(); }
void oracle_batch_release(sqlrs_batch_t *b) {
  if (b && b->owner) delete (OwnedBatch *)b->owner;
}

// ------------------------------------------------------------------ Filter --
int oracle_filter_create(oracle_ctx *ctx, const sqlrs_expr_t *expr, oracle_filter **out) {
  return guard(ctx, [&] {
    auto *f = new oracle_filter();
    f->ctx = ctx;
    f->expr = expr_from_abi(expr);
    *out = f;
  });
}
// one iteration of the for_await loop  [ref: filter.rs:16-24]
int oracle_filter_push(oracle_filter *f, const sqlrs_batch_t *in, int, sqlrs_batch_t **out) {
  return guard(f->ctx, [&] {
    Batch batch = batch_from_abi(in);
    Col mask = eval_column(f->expr, batch);
    std::vector<int64_t> keep = true_rows(mask);
    Batch o;
    for (const Col &c : batch.cols) o.cols.push_back(take(c, keep));
    o.rows = (int64_t)keep.size();
    *out = batch_to_abi(std::move(o));
  });
}
void oracle_filter_destroy(oracle_filter *f) { delete f; }

int oracle_eval_expr(oracle_ctx *ctx, const sqlrs_expr_t *expr, const sqlrs_batch_t *in, int,
                     sqlrs_batch_t **out) {
  return guard(ctx, [&] {
    Batch batch = batch_from_abi(in);
    Batch o;
    o.cols.push_back(eval_column(expr_from_abi(expr), batch));
    o.rows = batch.rows;
    *out = batch_to_abi(std::move(o));
  });
}

// ---------------------------------------------------------------- HashJoin --
int oracle_hash_join_create(oracle_ctx *ctx, int join_type, int num_keys,
                            const sqlrs_expr_t *left_keys, const sqlrs_expr_t *right_keys,
                            const sqlrs_expr_t *filter, int num_right_columns,
                            const int32_t *right_dtypes, oracle_hash_join **out) {
  return guard(ctx, [&] {
    if (num_keys < 1) fail(SQLRS_ERR_INTERNAL, "HashJoin must has on condition"); // :132
    auto j = std::unique_ptr<oracle_hash_join>(new oracle_hash_join());
    j->ctx = ctx;
    j->join_type = join_type;
    for (int i = 0; i < num_keys; i++) {
      j->on_left_keys.push_back(expr_from_abi(&left_keys[i]));
      j->on_right_keys.push_back(expr_from_abi(&right_keys[i]));
    }
    if (filter && filter->num_nodes > 0) {
      j->has_filter = true;
      j->filter = expr_from_abi(filter);
    }
    j->right_dtypes.assign(right_dtypes, right_dtypes + num_right_columns);
    *out = j.release();
  });
}
// [ref: hash_join.rs:161-181]
int oracle_hash_join_build_push(oracle_hash_join *j, const sqlrs_batch_t *left) {
  return guard(j->ctx, [&] {
    Batch batch = batch_from_abi(left);
    std::vector<Col> left_keys;
    for (const Expr &e : j->on_left_keys) left_keys.push_back(eval_column(e, batch));
    std::vector<uint64_t> every_rows_hashes((size_t)batch.rows, 0);
    create_hashes(left_keys, every_rows_hashes);
    for (size_t row = 0; row < every_rows_hashes.size(); row++)
      j->left_hashmap[every_rows_hashes[row]].push_back(row + j->left_row_offset);
    j->left_row_offset += (size_t)batch.rows;
    j->left_batches.push_back(std::move(batch));
  });
}
// [ref: hash_join.rs:183-206]
int oracle_hash_join_build_finish(oracle_hash_join *j) {
  return guard(j->ctx, [&] {
    j->build_finished = true;
    if (j->left_batches.empty()) return; // join emits nothing (:183-185)
    j->left_single_batch = concat_batches(j->left_batches);
    if (j->join_type == SQLRS_JOIN_LEFT || j->join_type == SQLRS_JOIN_FULL)
      j->visited_left_side.assign((size_t)j->left_single_batch.rows, 0);
  });
}
// [ref: hash_join.rs:207-292]
int oracle_hash_join_probe_push(oracle_hash_join *j, const sqlrs_batch_t *right, int,
                                sqlrs_batch_t **out) {
  return guard(j->ctx, [&] {
    require_build_finished(j);
    *out = nullptr;
    if (j->left_batches.empty()) return;
    Batch batch = batch_from_abi(right);
    *out = batch_to_abi(join_probe(j, batch));
  });
}
// [ref: hash_join.rs:217-253]
int oracle_hash_join_probe_indices(oracle_hash_join *j, const sqlrs_batch_t *right, int,
                                   sqlrs_batch_t **out) {
  return guard(j->ctx, [&] {
    require_build_finished(j);
    *out = nullptr;
    if (j->left_batches.empty()) return;
    Batch batch = batch_from_abi(right);
    std::vector<int64_t> li, ri;
    probe_indices(j, batch, li, ri);
    Builder bl(SQLRS_UINT64), br(SQLRS_UINT32);
    for (size_t k = 0; k < li.size(); k++) {
      Scalar s;
      s.dtype = SQLRS_UINT64;
      s.null = li[k] < 0;
      s.i = li[k];
      bl.append(s);
      Scalar t;
      t.dtype = SQLRS_UINT32;
      t.null = false;
      t.i = ri[k];
      br.append(t);
    }
    Batch o;
    o.cols.push_back(bl.finish());
    o.cols.push_back(br.finish());
    o.rows = (int64_t)li.size();
    *out = batch_to_abi(std::move(o));
  });
}
// [ref: hash_join.rs:296-322]
int oracle_hash_join_finish(oracle_hash_join *j, int, sqlrs_batch_t **out) {
  return guard(j->ctx, [&] {
    require_build_finished(j);
    *out = nullptr;
    if (j->left_batches.empty()) return;
    if (j->join_type != SQLRS_JOIN_LEFT && j->join_type != SQLRS_JOIN_FULL) return;
    std::vector<int64_t> indices;
    for (size_t v = 0; v < j->visited_left_side.size(); v++)
      if (!j->visited_left_side[v]) indices.push_back((int64_t)v);
    Batch o;
    for (const Col &c : j->left_single_batch.cols) o.cols.push_back(take(c, indices));
    for (int32_t dt : j->right_dtypes) o.cols.push_back(null_array(dt, (int64_t)indices.size()));
    o.rows = (int64_t)indices.size();
    *out = batch_to_abi(std::move(o));
  });
}
void oracle_hash_join_destroy(oracle_hash_join *j) { delete j; }

// ----------------------------------------------------------------- HashAgg --
int oracle_hash_agg_create(oracle_ctx *ctx, int num_group_by, const sqlrs_expr_t *group_by,
                           int num_aggs, const sqlrs_agg_func_t *aggs, oracle_hash_agg **out) {
  return guard(ctx, [&] {
    auto a = std::unique_ptr<oracle_hash_agg>(new oracle_hash_agg());
    a->ctx = ctx;
    for (int i = 0; i < num_group_by; i++) a->group_by.push_back(expr_from_abi(&group_by[i]));
    for (int i = 0; i < num_aggs; i++) {
      AggSpec s;
      s.func = aggs[i].func;
      s.distinct = aggs[i].distinct;
      s.return_dtype = aggs[i].return_dtype;
      s.arg = expr_from_abi(&aggs[i].arg);
      a->agg_funcs.push_back(std::move(s));
    }
    *out = a.release();
  });
}
// one iteration of the for_await loop  [ref: hash_agg.rs:44-122]
int oracle_hash_agg_push(oracle_hash_agg *a, const sqlrs_batch_t *in) {
  return guard(a->ctx, [&] {
    Batch batch = batch_from_abi(in);
    // 2.1 agg argument columns (exprs[0] only, :63-66)
    std::vector<Col> columns;
    for (const AggSpec &s : a->agg_funcs) columns.push_back(eval_column(s.arg, batch));
    // 2.2 group key columns (:69-73)
    std::vector<Col> group_keys;
    for (const Expr &e : a->group_by) group_keys.push_back(eval_column(e, batch));
    // 1. schema from the first batch (:47-59)
    if (!a->saw_batch) {
      a->saw_batch = true;
      for (const Col &k : group_keys) a->field_dtypes.push_back(k.dtype);
      for (const AggSpec &s : a->agg_funcs) a->field_dtypes.push_back(s.return_dtype);
    }
    // 3.1 row hashes (:76-77)
    std::vector<uint64_t> every_rows_hashes((size_t)batch.rows, 0);
    create_hashes(group_keys, every_rows_hashes);
    // 3.2 (:85-110)
    std::unordered_map<uint64_t, std::vector<int64_t>> group_hash_2_row_indices;
    for (size_t row = 0; row < every_rows_hashes.size(); row++) {
      uint64_t hash = every_rows_hashes[row];
      if (!a->group_hash_2_accs.count(hash)) {
        std::vector<std::unique_ptr<Accumulator>> accs;
        for (const AggSpec &s : a->agg_funcs)
          accs.push_back(create_accumulator(s, a->ctx->compat_count_last_batch != 0));
        a->group_hash_2_accs.emplace(hash, std::move(accs));
        std::vector<Scalar> keys;
        for (const Col &k : group_keys) keys.push_back(scalar_at(k, (int64_t)row));
        a->group_hash_2_keys.emplace(hash, std::move(keys));
        a->group_hashs.push_back(hash);
      }
      group_hash_2_row_indices[hash].push_back((int64_t)row);
    }
    // 4. per group: take + update_batch (:113-121)
    for (auto &kv : group_hash_2_row_indices) {
      auto &accs = a->group_hash_2_accs[kv.first];
      for (size_t k = 0; k < accs.size(); k++) {
        Col new_array = take(columns[k], kv.second);
        accs[k]->update_batch(new_array);
      }
    }
  });
}
// [ref: hash_agg.rs:124-149]
int oracle_hash_agg_finish(oracle_hash_agg *a, int, sqlrs_batch_t **out) {
  return guard(a->ctx, [&] {
    if (!a->saw_batch) // group_and_agg_fields.unwrap() on None panics (:125)
      fail(SQLRS_ERR_INTERNAL, "hash agg finished without any input batch");
    std::vector<Builder> builders;
    for (int32_t dt : a->field_dtypes) builders.emplace_back(dt);
    for (uint64_t hash : a->group_hashs) {
      const auto &group_values = a->group_hash_2_keys[hash];
      for (size_t idx = 0; idx < group_values.size(); idx++) builders[idx].append(group_values[idx]);
      const auto &accs = a->group_hash_2_accs[hash];
      for (size_t idx = 0; idx < accs.size(); idx++) {
        Scalar v = accs[idx]->evaluate();
        Builder &b = builders[idx + group_values.size()];
        if (!v.null && v.dtype != b.c.dtype) {
          // append_scalar_value_for_builder downcasts by builder type (types/mod.rs:248-273)
          fail(SQLRS_ERR_INTERNAL, "accumulator result type does not match return type");
        }
        b.append(v);
      }
    }
    Batch o;
    for (Builder &b : builders) o.cols.push_back(b.finish());
    o.rows = (int64_t)a->group_hashs.size();
    *out = batch_to_abi(std::move(o));
  });
}
void oracle_hash_agg_destroy(oracle_hash_agg *a) { delete a; }

// ------------------------------------------------------------------- Order --
int oracle_order_create(oracle_ctx *ctx, int num_keys, const sqlrs_order_by_t *order_by,
                        oracle_order **out) {
  return guard(ctx, [&] {
    auto o = std::unique_ptr<oracle_order>(new oracle_order());
    o->ctx = ctx;
    for (int i = 0; i < num_keys; i++) {
      o->exprs.push_back(expr_from_abi(&order_by[i].expr));
      o->asc.push_back(order_by[i].asc);
    }
    *out = o.release();
  });
}
int oracle_order_push(oracle_order *o, const sqlrs_batch_t *in) {
  return guard(o->ctx, [&] { o->batches.push_back(batch_from_abi(in)); });
}
// [ref: order.rs:27-66]; lexsort_to_indices with nulls_first = true, ties kept in input
// order (arrow-ord 28 leaves tie order unspecified; no reference test pins it)
int oracle_order_finish(oracle_order *o, int, sqlrs_batch_t **out) {
  return guard(o->ctx, [&] {
    if (o->batches.empty()) fail(SQLRS_ERR_INTERNAL, "order finished without any input batch"); // schema.unwrap() :27
    Batch batch = concat_batches(o->batches);
    std::vector<Col> sort_cols;
    for (const Expr &e : o->exprs) sort_cols.push_back(eval_column(e, batch));
    std::vector<int64_t> indices((size_t)batch.rows);
    for (size_t i = 0; i < indices.size(); i++) indices[i] = (int64_t)i;
    std::stable_sort(indices.begin(), indices.end(), [&](int64_t x, int64_t y) {
      for (size_t k = 0; k < sort_cols.size(); k++) {
        const Col &c = sort_cols[k];
        bool vx = c.valid(x), vy = c.valid(y);
        if (!vx || !vy) {
          if (vx == vy) continue;
          return !vx; // nulls first regardless of direction
        }
        int cmp = cmp_scalar(scalar_at(c, x), scalar_at(c, y));
        if (cmp == 0) continue;
        return o->asc[k] ? cmp < 0 : cmp > 0;
      }
      return false;
    });
    Batch r;
    for (const Col &c : batch.cols) r.cols.push_back(take(c, indices));
    r.rows = batch.rows;
    *out = batch_to_abi(std::move(r));
  });
}
void oracle_order_destroy(oracle_order *o) { delete o; }

// ----------------------------------------------------------------- Project --
// [ref: src/executor/project.rs:6-29] one output batch per input batch, column i = exprs[i].eval_column
struct oracle_project {
  oracle_ctx *ctx;
  std::vector<Expr> exprs;
};
int oracle_project_create(oracle_ctx *ctx, int num_exprs, const sqlrs_expr_t *exprs, oracle_project **out) {
  return guard(ctx, [&] {
    auto p = std::unique_ptr<oracle_project>(new oracle_project());
    p->ctx = ctx;
    for (int i = 0; i < num_exprs; i++) p->exprs.push_back(expr_from_abi(&exprs[i]));
    *out = p.release();
  });
}
int oracle_project_push(oracle_project *p, const sqlrs_batch_t *in, int, sqlrs_batch_t **out) {
  return guard(p->ctx, [&] {
    Batch batch = batch_from_abi(in);
    Batch o;
    o.rows = batch.rows;
    for (const Expr &e : p->exprs) o.cols.push_back(eval_column(e, batch)); // project.rs:17-21
    *out = batch_to_abi(std::move(o));
  });
}
void oracle_project_destroy(oracle_project *p) { delete p; }

// --------------------------------------------------------------- CrossJoin --
// [ref: src/executor/join/cross_join.rs:8-58] left child collected + concatenated (:30-36); per right batch and per
// left ROW one output batch = that row's scalars repeated right.num_rows times | the right columns (:39-55).  The ABI
// returns the batches of one right batch as ONE batch in the same row order (the host mirror slices it back).
struct oracle_cross_join {
  oracle_ctx *ctx;
  std::vector<Batch> left;
};
int oracle_cross_join_create(oracle_ctx *ctx, oracle_cross_join **out) {
  return guard(ctx, [&] {
    auto j = std::unique_ptr<oracle_cross_join>(new oracle_cross_join());
    j->ctx = ctx;
    *out = j.release();
  });
}
int oracle_cross_join_build_push(oracle_cross_join *j, const sqlrs_batch_t *left) {
  return guard(j->ctx, [&] { j->left.push_back(batch_from_abi(left)); });
}
int oracle_cross_join_probe_push(oracle_cross_join *j, const sqlrs_batch_t *right, int, sqlrs_batch_t **out) {
  return guard(j->ctx, [&] {
    *out = nullptr;
    if (j->left.empty()) return; // cross_join.rs:32-34
    const Batch left = concat_batches(j->left); // cross_join.rs:36
    const Batch r = batch_from_abi(right);
    if (left.rows == 0) return;
    std::vector<Batch> per_row; // the reference's output stream for this right batch
    for (int64_t row = 0; row < left.rows; row++) { // cross_join.rs:43
      Batch o;
      o.rows = r.rows;
      const std::vector<int64_t> rep((size_t)r.rows, row); // build_scalar_value_array(scalar, right rows) (:47-50)
      for (const Col &c : left.cols) o.cols.push_back(take(c, rep));
      for (const Col &c : r.cols) o.cols.push_back(c); // :53
      per_row.push_back(std::move(o));
    }
    *out = batch_to_abi(concat_batches(per_row));
  });
}
void oracle_cross_join_destroy(oracle_cross_join *j) { delete j; }

// ------------------------------------------------------------------- Limit --
// [ref: src/executor/limit.rs:4-81]
struct oracle_limit {
  oracle_ctx *ctx;
  bool has_limit;
  uint64_t limit, offset_val;
  uint64_t returned_count = 0;
  bool done = false;
};
int oracle_limit_create(oracle_ctx *ctx, int has_limit, int64_t limit, int has_offset, int64_t offset, oracle_limit **out) {
  return guard(ctx, [&] {
    auto l = std::unique_ptr<oracle_limit>(new oracle_limit());
    l->ctx = ctx;
    l->has_limit = has_limit != 0;
    l->limit = (uint64_t)limit;
    l->offset_val = has_offset ? (uint64_t)offset : 0; // limit.rs:21-27
    if (l->has_limit && l->limit == 0) l->done = true;  // limit.rs:29-31: nothing is ever yielded
    *out = l.release();
  });
}
int oracle_limit_push(oracle_limit *l, const sqlrs_batch_t *in, int, sqlrs_batch_t **out, int *done) {
  return guard(l->ctx, [&] {
    *out = nullptr;
    if (l->done) {
      *done = 1;
      return;
    }
    Batch batch = batch_from_abi(in);
    const uint64_t cardinality = (uint64_t)batch.rows;
    const uint64_t limit_val = l->has_limit ? l->limit : cardinality;                       // :39
    const uint64_t start = std::max(l->returned_count, l->offset_val) - l->returned_count;   // :41
    const uint64_t total_end = l->offset_val + limit_val;                                    // :45
    const uint64_t current_batch_end = l->returned_count + cardinality;                      // :46
    const uint64_t end = std::min(total_end, current_batch_end) - l->returned_count;         // :49-51
    l->returned_count += cardinality;                                                        // :54
    if (start < end) {                                                                       // :61-63
      if (start == 0 && end == cardinality) {
        *out = batch_to_abi(std::move(batch));                                               // :65-66
      } else {
        std::vector<int64_t> idx;
        for (uint64_t i = start; i < end; i++) idx.push_back((int64_t)i);
        Batch o;
        o.rows = (int64_t)(end - start);
        for (const Col &c : batch.cols) o.cols.push_back(take(c, idx));                      // batch.slice(start, length)
        *out = batch_to_abi(std::move(o));
      }
      if (l->returned_count >= l->offset_val + limit_val) l->done = true;                    // :76-78
    }
    *done = l->done ? 1 : 0;
  });
}
void oracle_limit_destroy(oracle_limit *l) { delete l; }

// --------------------------------------------------------------- SimpleAgg --
// [ref: src/executor/aggregate/simple_agg.rs:10-66] one accumulator per aggregate over ALL rows,
// one output row; without any input batch `agg_fileds.unwrap()` panics (:63)
struct oracle_simple_agg {
  oracle_ctx *ctx;
  std::vector<AggSpec> agg_funcs;
  std::vector<std::unique_ptr<Accumulator>> accs;
  bool saw_batch = false;
};
int oracle_simple_agg_create(oracle_ctx *ctx, int num_aggs, const sqlrs_agg_func_t *aggs, oracle_simple_agg **out) {
  return guard(ctx, [&] {
    auto a = std::unique_ptr<oracle_simple_agg>(new oracle_simple_agg());
    a->ctx = ctx;
    for (int i = 0; i < num_aggs; i++) {
      AggSpec s;
      s.func = aggs[i].func;
      s.distinct = aggs[i].distinct;
      s.return_dtype = aggs[i].return_dtype;
      s.arg = expr_from_abi(&aggs[i].arg);
      a->accs.push_back(create_accumulator(s, ctx->compat_count_last_batch != 0)); // :29
      a->agg_funcs.push_back(std::move(s));
    }
    *out = a.release();
  });
}
int oracle_simple_agg_push(oracle_simple_agg *a, const sqlrs_batch_t *in) {
  return guard(a->ctx, [&] {
    Batch batch = batch_from_abi(in);
    a->saw_batch = true;
    for (size_t k = 0; k < a->agg_funcs.size(); k++)
      a->accs[k]->update_batch(eval_column(a->agg_funcs[k].arg, batch)); // :38-56
  });
}
int oracle_simple_agg_finish(oracle_simple_agg *a, int, sqlrs_batch_t **out) {
  return guard(a->ctx, [&] {
    if (!a->saw_batch) fail(SQLRS_ERR_INTERNAL, "simple agg finished without any input batch");
    Batch o;
    o.rows = 1;
    for (size_t k = 0; k < a->accs.size(); k++) {
      Builder b(a->agg_funcs[k].return_dtype);
      b.append(a->accs[k]->evaluate()); // build_scalar_value_array(&res, 1)  (:59-62)
      o.cols.push_back(b.finish());
    }
    *out = batch_to_abi(std::move(o));
  });
}
void oracle_simple_agg_destroy(oracle_simple_agg *a) { delete a; }

// ----------------------------------------------------- record_batch_to_string --
// [ref: src/util/mod.rs:53-80] one line per row, columns separated by one blank, NULL -> "NULL",
// empty Utf8 -> "(empty)", everything else arrow's array_value_to_string = Rust's Display of the
// value (floats: shortest digits that round-trip, positional notation, no trailing ".0").
// Returns a malloc'd NUL-terminated string in *out (caller frees with oracle_string_free).
int oracle_batch_to_string(oracle_ctx *ctx, const sqlrs_batch_t *in, char **out) {
  return guard(ctx, [&] {
    Batch b = batch_from_abi(in);
    std::string s;
    for (int64_t row = 0; row < b.rows; row++) {
      for (size_t c = 0; c < b.cols.size(); c++) {
        if (c) s.push_back(' ');
        const Col &col = b.cols[c];
        if (!col.valid(row)) {
          s += "NULL";
          continue;
        }
        Scalar v = scalar_at(col, row);
        switch (col.dtype) {
        case SQLRS_UTF8:
          s += v.s.empty() ? std::string("(empty)") : v.s;
          break;
        case SQLRS_BOOLEAN:
          s += v.i ? "true" : "false";
          break;
        case SQLRS_FLOAT64: {
          if (std::isnan(v.f)) s += "NaN";
          else if (std::isinf(v.f)) s += v.f < 0 ? "-inf" : "inf";
          else {
            char buf[400];
            auto r = std::to_chars(buf, buf + sizeof(buf), v.f, std::chars_format::fixed);
            s.append(buf, r.ptr);
          }
          break;
        }
        case SQLRS_UINT64:
          s += std::to_string((uint64_t)v.i);
          break;
        default:
          s += std::to_string(v.i);
        }
      }
      s.push_back('\n');
    }
    char *p = (char *)std::malloc(s.size() + 1);
    if (!p) fail(SQLRS_ERR_INTERNAL, "allocation failed");
    std::memcpy(p, s.c_str(), s.size() + 1);
    *out = p;
  });
}
void oracle_string_free(char *s) { std::free(s); }

const char *oracle_version(void) { return "sqlrs-oracle 0.1 (cpu restatement, test only)"; }

} // extern "C"
