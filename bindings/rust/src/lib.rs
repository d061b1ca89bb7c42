//! MI355X backend for the sqlrs v1 executor hot path — Rust shim over `libsqlrs_hip.so`.
//!
//! Inside sqlrs this crate is `src/executor/hip/` and `crate::...` paths below resolve to sqlrs's own
//! modules (`crate::binder::BoundExpr`, `crate::optimizer::Physical*`, `crate::executor::{BoxedExecutor,
//! ExecutorError}`); the four `visit_physical_*` bodies of `src/executor/mod.rs:103-114,139-149,163-174,
//! 189-199` call the `Hip*Executor` structs of [`executors`] instead of the CPU ones.
#![feature(generators, proc_macro_hygiene, stmt_expr_attributes)] // what futures-async-stream needs (sqlrs: nightly-2022-07-29)

pub mod convert;
pub mod executors;
pub mod ffi;
