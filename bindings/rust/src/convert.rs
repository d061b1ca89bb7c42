//! Arrow <-> `sqlrs_column_t`, `BoundExpr` -> postfix `sqlrs_expr_t`, status -> `ExecutorError`.
use std::ffi::{CStr, CString};
use std::os::raw::c_int;
use std::sync::Arc;

use arrow::array::{make_array, ArrayData, ArrayRef};
use arrow::buffer::Buffer;
use arrow::datatypes::{DataType, SchemaRef};
use arrow::error::ArrowError;
use arrow::record_batch::RecordBatch;

use crate::binder::{BoundExpr, BoundTypeCast};
use crate::executor::ExecutorError;
use crate::ffi::*;
use crate::types::ScalarValue;

/// One ctx = one HIP stream on one GPU; the handle the replacement `ExecutorBuilder` holds.
pub struct HipCtx(*mut sqlrs_ctx_t);
unsafe impl Send for HipCtx {}
unsafe impl Sync for HipCtx {}

impl HipCtx {
    pub fn new(device_id: i32) -> Result<Arc<Self>, ExecutorError> {
        let mut raw = std::ptr::null_mut();
        match unsafe { sqlrs_ctx_create(device_id, &mut raw) } {
            SQLRS_OK => Ok(Arc::new(HipCtx(raw))),
            _ => Err(ExecutorError::InternalError("no usable gfx950 device (libsqlrs_hip has no CPU fallback)".into())),
        }
    }
    pub fn raw(&self) -> *mut sqlrs_ctx_t { self.0 }
    pub fn last_error(&self) -> String {
        unsafe { CStr::from_ptr(sqlrs_last_error(self.0)) }.to_string_lossy().into_owned()
    }
    /// status codes -> `ExecutorError` (executor/mod.rs:67-85): 1 Arrow, 2 InternalError, 4 (device) has no analogue
    pub fn check(&self, status: c_int) -> Result<(), ExecutorError> {
        match status {
            SQLRS_OK => Ok(()),
            SQLRS_ERR_ARROW => Err(ExecutorError::Arrow(ArrowError::ComputeError(self.last_error()))),
            _ => Err(ExecutorError::InternalError(self.last_error())),
        }
    }
}
impl Drop for HipCtx {
    fn drop(&mut self) { unsafe { sqlrs_ctx_destroy(self.0) } }
}

pub fn dtype_of(t: &DataType) -> Result<i32, ExecutorError> {
    Ok(match t {
        DataType::Int32 => SQLRS_INT32,
        DataType::Int64 => SQLRS_INT64,
        DataType::Float64 => SQLRS_FLOAT64,
        DataType::Boolean => SQLRS_BOOLEAN,
        DataType::Utf8 => SQLRS_UTF8,
        t => return Err(ExecutorError::InternalError(format!("unsupported data type {t}"))),
    })
}

/// zero-copy view of one arrow array (arrow 28 `ArrayData`); slices are materialised first (ABI: offset 0)
pub fn column_of(a: &ArrayRef) -> Result<(sqlrs_column_t, ArrayRef), ExecutorError> {
    let a = if a.data().offset() != 0 { arrow::compute::concat(&[a.as_ref()])? } else { a.clone() };
    let d = a.data();
    let dtype = dtype_of(a.data_type())?;
    let (values, offsets) = match a.data_type() {
        DataType::Utf8 => (d.buffers()[1].as_ptr(), d.buffers()[0].as_ptr() as *const i32),
        _ => (d.buffers()[0].as_ptr(), std::ptr::null()),
    };
    let col = sqlrs_column_t {
        dtype, mem: SQLRS_MEM_HOST, length: d.len() as i64, null_count: d.null_count() as i64,
        values: values as _, validity: d.null_buffer().map_or(std::ptr::null(), |b| b.as_ptr()), offsets,
    };
    Ok((col, a)) // the returned ArrayRef keeps the buffers alive for the duration of the call
}

/// A `RecordBatch` as a caller-built `sqlrs_batch_t` (HOST columns are read before the call returns).
pub struct AbiBatch { pub raw: sqlrs_batch_t, _cols: Vec<sqlrs_column_t>, _keep: Vec<ArrayRef> }
impl AbiBatch {
    pub fn new(batch: &RecordBatch) -> Result<Self, ExecutorError> {
        let mut cols = Vec::new();
        let mut keep = Vec::new();
        for a in batch.columns() {
            let (c, k) = column_of(a)?;
            cols.push(c);
            keep.push(k);
        }
        let raw = sqlrs_batch_t { num_rows: batch.num_rows() as i64, num_columns: cols.len() as i32, reserved: 0,
                                  columns: cols.as_mut_ptr(), owner: std::ptr::null_mut() };
        Ok(AbiBatch { raw, _cols: cols, _keep: keep })
    }
}

/// The same hand-over through the Arrow C Data Interface (`sqlrs_batch_import_arrow`, round 6): the batch is exported as a
/// struct array with arrow's own `arrow::ffi` and MOVED into the library — no descriptor is built by hand and nothing is
/// copied; the library releases the arrow buffers when the returned batch is released.
pub struct FfiBatch(pub *mut sqlrs_batch_t);
impl FfiBatch {
    pub fn new(ctx: &HipCtx, batch: &RecordBatch) -> Result<Self, ExecutorError> {
        let data = arrow::array::StructArray::from(batch.clone()).into_data();
        let mut array = arrow::ffi::FFI_ArrowArray::new(&data);
        let mut schema = arrow::ffi::FFI_ArrowSchema::try_from(data.data_type())?;
        let mut out = std::ptr::null_mut();
        // (on failure nothing was consumed: `array` / `schema` release their exports when they drop)
        ctx.check(unsafe { sqlrs_batch_import_arrow(ctx.raw(), &mut array, &mut schema, &mut out) })?;
        std::mem::forget(array); // moved: their `release` fields are NULL now, nothing left to drop
        std::mem::forget(schema);
        Ok(FfiBatch(out))
    }
}
impl Drop for FfiBatch {
    fn drop(&mut self) { unsafe { sqlrs_batch_release(self.0) } }
}
/// ... and back: a library batch MOVED out as (FFI_ArrowArray, FFI_ArrowSchema) and imported by arrow without a copy
/// (`sqlrs_batch_export_arrow`; device-resident batches are downloaded by the call).
pub fn export_batch_ffi(ctx: &HipCtx, schema: SchemaRef, out: *mut sqlrs_batch_t) -> Result<RecordBatch, ExecutorError> {
    let names: Vec<CString> = schema.fields().iter().map(|f| CString::new(f.name().as_str()).unwrap()).collect();
    let ptrs: Vec<*const std::os::raw::c_char> = names.iter().map(|n| n.as_ptr()).collect();
    let mut array = arrow::ffi::FFI_ArrowArray::empty();
    let mut fschema = arrow::ffi::FFI_ArrowSchema::empty();
    ctx.check(unsafe { sqlrs_batch_export_arrow(ctx.raw(), out, ptrs.as_ptr(), &mut array, &mut fschema) })?;
    let data = arrow::ffi::ArrowArray::new(array, fschema).to_data()?;
    Ok(RecordBatch::from(&arrow::array::StructArray::from(data)))
}

/// Copies a library batch (out_mem = HOST) into arrow buffers, then releases it.
pub fn import_batch(schema: SchemaRef, out: *mut sqlrs_batch_t) -> Result<RecordBatch, ExecutorError> {
    let b = unsafe { &*out };
    let mut arrays: Vec<ArrayRef> = Vec::with_capacity(b.num_columns as usize);
    for i in 0..b.num_columns as usize {
        let c = unsafe { &*b.columns.add(i) };
        let n = c.length as usize;
        let bitmap_bytes = (n + 7) / 8;
        let slice = |p: *const u8, len: usize| unsafe { Buffer::from_slice_ref(std::slice::from_raw_parts(p, len)) };
        let t = schema.field(i).data_type().clone();
        let mut builder = ArrayData::builder(t.clone()).len(n);
        if !c.validity.is_null() && c.null_count != 0 {
            builder = builder.null_bit_buffer(Some(slice(c.validity, bitmap_bytes)));
        }
        builder = match t {
            DataType::Utf8 => {
                let offs = unsafe { std::slice::from_raw_parts(c.offsets, n + 1) };
                builder.add_buffer(Buffer::from_slice_ref(offs)).add_buffer(slice(c.values as _, offs[n] as usize))
            }
            DataType::Boolean => builder.add_buffer(slice(c.values as _, bitmap_bytes)),
            DataType::Int32 => builder.add_buffer(slice(c.values as _, 4 * n)),
            _ => builder.add_buffer(slice(c.values as _, 8 * n)),
        };
        arrays.push(make_array(builder.build()?));
    }
    unsafe { sqlrs_batch_release(out) };
    Ok(RecordBatch::try_new(schema, arrays)?)
}

/// `BoundExpr` lowered to the postfix `sqlrs_expr_node_t` encoding: exactly the cases of evaluator.rs:14-27.
pub struct Lowered { nodes: Vec<sqlrs_expr_node_t>, _strings: Vec<CString> }
impl Lowered {
    pub fn abi(&self) -> sqlrs_expr_t { sqlrs_expr_t { nodes: self.nodes.as_ptr(), num_nodes: self.nodes.len() as i32, reserved: 0 } }
}
pub fn lower(e: &BoundExpr) -> Result<Lowered, ExecutorError> {
    fn node(op: i32) -> sqlrs_expr_node_t {
        sqlrs_expr_node_t { op, dtype: 0, index: 0, is_null: 0, i: 0, f: 0.0, s: std::ptr::null() }
    }
    fn push(e: &BoundExpr, out: &mut Lowered) -> Result<(), ExecutorError> {
        match e {
            BoundExpr::InputRef(r) => { let mut n = node(SQLRS_EXPR_INPUT_REF); n.index = r.index as i32; out.nodes.push(n); }
            BoundExpr::Constant(v) => {
                let mut n = node(SQLRS_EXPR_CONSTANT);
                match v {
                    ScalarValue::Null => { n.dtype = SQLRS_NULLTYPE; n.is_null = 1; }
                    ScalarValue::Boolean(x) => { n.dtype = SQLRS_BOOLEAN; n.is_null = x.is_none() as i32; n.i = x.unwrap_or(false) as i64; }
                    ScalarValue::Int32(x) => { n.dtype = SQLRS_INT32; n.is_null = x.is_none() as i32; n.i = x.unwrap_or(0) as i64; }
                    ScalarValue::Int64(x) => { n.dtype = SQLRS_INT64; n.is_null = x.is_none() as i32; n.i = x.unwrap_or(0); }
                    ScalarValue::Float64(x) => { n.dtype = SQLRS_FLOAT64; n.is_null = x.is_none() as i32; n.f = x.unwrap_or(0.0); }
                    ScalarValue::String(x) => {
                        n.dtype = SQLRS_UTF8;
                        n.is_null = x.is_none() as i32;
                        let c = CString::new(x.clone().unwrap_or_default()).map_err(|e| ExecutorError::InternalError(e.to_string()))?;
                        n.s = c.as_ptr();
                        out._strings.push(c); // (CString's heap buffer does not move with the Vec)
                    }
                }
                out.nodes.push(n);
            }
            BoundExpr::BinaryOp(b) => {
                push(&b.left, out)?;
                push(&b.right, out)?;
                use sqlparser::ast::BinaryOperator as Op;
                out.nodes.push(node(match b.op {
                    Op::Plus => SQLRS_EXPR_PLUS, Op::Minus => SQLRS_EXPR_MINUS, Op::Multiply => SQLRS_EXPR_MULTIPLY,
                    Op::Divide => SQLRS_EXPR_DIVIDE, Op::Gt => SQLRS_EXPR_GT, Op::Lt => SQLRS_EXPR_LT,
                    Op::GtEq => SQLRS_EXPR_GTEQ, Op::LtEq => SQLRS_EXPR_LTEQ, Op::Eq => SQLRS_EXPR_EQ,
                    Op::NotEq => SQLRS_EXPR_NOTEQ, Op::And => SQLRS_EXPR_AND, Op::Or => SQLRS_EXPR_OR,
                    ref o => return Err(ExecutorError::InternalError(format!("unsupported binary operator {o}"))),
                }));
            }
            BoundExpr::TypeCast(BoundTypeCast { expr, cast_type }) => {
                push(expr, out)?;
                let mut n = node(SQLRS_EXPR_TYPE_CAST);
                n.dtype = dtype_of(cast_type)?;
                out.nodes.push(n);
            }
            BoundExpr::Alias(a) => push(&a.expr, out)?,
            BoundExpr::ColumnRef(_) | BoundExpr::AggFunc(_) | BoundExpr::Subquery(_) => {
                return Err(ExecutorError::InternalError("ColumnRef / AggFunc / Subquery never reach eval_column (evaluator.rs:17,26)".into()))
            }
        }
        Ok(())
    }
    let mut out = Lowered { nodes: Vec::new(), _strings: Vec::new() };
    push(e, &mut out)?;
    Ok(out)
}
