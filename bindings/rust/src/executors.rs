//! The executors `ExecutorBuilder` instantiates instead of the CPU ones, and the `visit_physical_*` bodies that do
//! it (src/executor/mod.rs:87-200).  Same struct fields as the reference operators (filter.rs:7-10,
//! hash_join.rs:16-23, cross_join.rs:8-13, hash_agg.rs:15-19, order.rs:8-11, project.rs:6-9, limit.rs:4-8, simple_agg.rs:9-12) plus
//! the `HipCtx` handle; every `execute` yields exactly the batches the CPU operator yields.
use std::sync::Arc;

use arrow::datatypes::{Field, Schema, SchemaRef};
use arrow::record_batch::RecordBatch;
use futures_async_stream::try_stream;

use crate::binder::{AggFunc, BoundAggFunc, BoundExpr, BoundOrderBy, JoinCondition, JoinType};
use crate::catalog::ColumnCatalog;
use crate::convert::{dtype_of, import_batch, lower, AbiBatch, HipCtx, Lowered};
use crate::executor::{BoxedExecutor, ExecutorBuilder, ExecutorError};
use crate::ffi::*;
use crate::optimizer::{
    PhysicalCrossJoin, PhysicalFilter, PhysicalHashAgg, PhysicalHashJoin, PhysicalLimit, PhysicalOrder, PhysicalProject,
    PhysicalSimpleAgg, PlanRef, PlanTreeNode,
};

/// destroys the operator handle when the stream is dropped (also on an early error)
struct Guard<T>(*mut T, unsafe extern "C" fn(*mut T));
impl<T> Drop for Guard<T> {
    fn drop(&mut self) { if !self.0.is_null() { unsafe { (self.1)(self.0) } } }
}
unsafe impl<T> Send for Guard<T> {}
/// the tickets an async stream has in flight: every one is waited for (and its batch released) when the stream is dropped
/// early — a ticket must be consumed before its ctx goes (sqlrs_hip.h) — declared AFTER the operator's Guard so that it is
/// dropped first
struct Tickets(std::collections::VecDeque<(*mut sqlrs_ticket_t, SchemaRef)>);
impl Drop for Tickets {
    fn drop(&mut self) {
        for (t, _) in self.0.drain(..) {
            let mut out = std::ptr::null_mut();
            if unsafe { sqlrs_batch_wait(t, &mut out) } == SQLRS_OK && !out.is_null() { unsafe { sqlrs_batch_release(out) } }
        }
    }
}
unsafe impl Send for Tickets {}

fn lower_all(exprs: &[BoundExpr]) -> Result<(Vec<Lowered>, Vec<sqlrs_expr_t>), ExecutorError> {
    let low: Vec<Lowered> = exprs.iter().map(lower).collect::<Result<_, _>>()?;
    let abi = low.iter().map(|l| l.abi()).collect();
    Ok((low, abi))
}
fn lower_aggs(aggs: &[BoundAggFunc]) -> Result<(Vec<Lowered>, Vec<sqlrs_agg_func_t>), ExecutorError> {
    // only exprs[0] is read by the reference (hash_agg.rs:65, simple_agg.rs:38-41)
    let low: Vec<Lowered> = aggs.iter().map(|a| lower(&a.exprs[0])).collect::<Result<_, _>>()?;
    let mut out = Vec::new();
    for (a, l) in aggs.iter().zip(&low) {
        let func = match a.func { AggFunc::Count => SQLRS_AGG_COUNT, AggFunc::Sum => SQLRS_AGG_SUM, AggFunc::Min => SQLRS_AGG_MIN, AggFunc::Max => SQLRS_AGG_MAX };
        out.push(sqlrs_agg_func_t { func, distinct: a.distinct as i32, return_dtype: dtype_of(&a.return_type)?, reserved: 0, arg: l.abi() });
    }
    Ok((low, out))
}
/// output schema of an aggregation: eval_field names / types (evaluator.rs:30-64), as the CPU operators build it
fn agg_schema(group_by: &[BoundExpr], aggs: &[BoundAggFunc], input: &RecordBatch) -> SchemaRef {
    let mut fields: Vec<Field> = group_by.iter().map(|g| g.eval_field(input)).collect(); // hash_agg.rs:49-60: keys, then aggregates
    fields.extend(aggs.iter().map(|a| BoundExpr::AggFunc(a.clone()).eval_field(input)));
    Arc::new(Schema::new(fields))
}

// ------------------------------------------------------------------ Filter --
pub struct HipFilterExecutor { pub ctx: Arc<HipCtx>, pub expr: BoundExpr, pub child: BoxedExecutor }
/// The same operator pulling `group` batches of its child at a time and handing them to `sqlrs_filter_push_many`: the same
/// stream of output batches (one per input batch, filter.rs:15-24) at one upload / launch sequence / download per GROUP —
/// 1024-row CSV batches (storage/csv.rs:105): 23 -> 934 Mrows/s from a native caller.
pub struct HipFilterManyExecutor { pub ctx: Arc<HipCtx>, pub expr: BoundExpr, pub child: BoxedExecutor, pub group: usize }
impl HipFilterManyExecutor {
    #[try_stream(boxed, ok = RecordBatch, error = ExecutorError)]
    pub async fn execute(self) {
        let expr = lower(&self.expr)?;
        let mut f = std::ptr::null_mut();
        self.ctx.check(unsafe { sqlrs_filter_create(self.ctx.raw(), &expr.abi(), &mut f) })?;
        let _g = Guard(f, sqlrs_filter_destroy);
        let mut pending: Vec<RecordBatch> = Vec::with_capacity(self.group);
        let mut child = self.child;
        loop {
            // (advisor r05: the end of the child's stream is `None`, not a short group — testing the group's fill right after
            //  the first push ended the loop after ONE batch and dropped the rest of the stream)
            let next = futures::StreamExt::next(&mut child).await;
            let end = next.is_none();
            if let Some(b) = next { pending.push(b?); }
            if pending.len() >= self.group.max(1) || (end && !pending.is_empty()) {
                let views: Vec<AbiBatch> = pending.iter().map(AbiBatch::new).collect::<Result<_, _>>()?;
                let ins: Vec<*const sqlrs_batch_t> = views.iter().map(|v| &v.raw as *const _).collect();
                let mut outs: Vec<*mut sqlrs_batch_t> = vec![std::ptr::null_mut(); ins.len()];
                self.ctx.check(unsafe { sqlrs_filter_push_many(f, ins.len() as i32, ins.as_ptr(), SQLRS_MEM_HOST, outs.as_mut_ptr()) })?;
                for (b, o) in pending.iter().zip(outs) { yield import_batch(b.schema(), o)?; }
                pending.clear();
            }
            if end { break; }
        }
    }
}
/// The same operator polled ONE batch at a time, as the reference's loop reads, through `sqlrs_filter_push_async`: `depth`
/// tickets in flight, the batch of the input read `depth` polls ago handed out by `sqlrs_batch_wait` — no stream
/// synchronisation per batch and no regrouping of the child's stream (22 -> 168 Mrows/s at 1024-row batches).
pub struct HipFilterAsyncExecutor { pub ctx: Arc<HipCtx>, pub expr: BoundExpr, pub child: BoxedExecutor, pub depth: usize }
impl HipFilterAsyncExecutor {
    #[try_stream(boxed, ok = RecordBatch, error = ExecutorError)]
    pub async fn execute(self) {
        let expr = lower(&self.expr)?;
        let mut f = std::ptr::null_mut();
        self.ctx.check(unsafe { sqlrs_filter_create(self.ctx.raw(), &expr.abi(), &mut f) })?;
        let _g = Guard(f, sqlrs_filter_destroy);
        let mut inflight = Tickets(std::collections::VecDeque::new());
        let mut child = self.child;
        let mut ended = false;
        loop {
            while !ended && inflight.0.len() <= self.depth.max(1) {
                match futures::StreamExt::next(&mut child).await {
                    None => ended = true,
                    Some(b) => {
                        let b = b?;
                        let inb = AbiBatch::new(&b)?; // read completely before push_async returns
                        let mut t = std::ptr::null_mut();
                        self.ctx.check(unsafe { sqlrs_filter_push_async(f, &inb.raw, &mut t) })?;
                        inflight.0.push_back((t, b.schema()));
                    }
                }
            }
            match inflight.0.pop_front() {
                None => break,
                Some((t, schema)) => {
                    let mut out = std::ptr::null_mut();
                    self.ctx.check(unsafe { sqlrs_batch_wait(t, &mut out) })?; // consumes the ticket, also on error
                    yield import_batch(schema, out)?;
                }
            }
        }
    }
}
impl HipFilterExecutor {
    #[try_stream(boxed, ok = RecordBatch, error = ExecutorError)]
    pub async fn execute(self) {
        let expr = lower(&self.expr)?;
        let mut f = std::ptr::null_mut();
        self.ctx.check(unsafe { sqlrs_filter_create(self.ctx.raw(), &expr.abi(), &mut f) })?;
        let _g = Guard(f, sqlrs_filter_destroy);
        #[for_await]
        for batch in self.child { // filter.rs:15-24: one output batch per input batch, empty ones included
            let batch = batch?;
            let inb = AbiBatch::new(&batch)?;
            let mut out = std::ptr::null_mut();
            self.ctx.check(unsafe { sqlrs_filter_push(f, &inb.raw, SQLRS_MEM_HOST, &mut out) })?;
            yield import_batch(batch.schema(), out)?;
        }
    }
}

// ---------------------------------------------------------------- HashJoin --
pub struct HipHashJoinExecutor {
    pub ctx: Arc<HipCtx>,
    pub left_child: BoxedExecutor,
    pub right_child: BoxedExecutor,
    pub join_type: JoinType,
    pub join_condition: JoinCondition,
    pub join_output_schema: Vec<ColumnCatalog>,
}
fn join_keys(cond: &JoinCondition) -> Result<(Vec<BoundExpr>, Vec<BoundExpr>, Option<BoundExpr>), ExecutorError> {
    match cond { // hash_join.rs:129-134
        JoinCondition::On { on, filter } => Ok((on.iter().map(|(l, _)| l.clone()).collect(), on.iter().map(|(_, r)| r.clone()).collect(), filter.clone())),
        JoinCondition::None => Err(ExecutorError::InternalError("HashJoin must has on condition".into())),
    }
}
impl HipHashJoinExecutor {
    #[try_stream(boxed, ok = RecordBatch, error = ExecutorError)]
    pub async fn execute(self) {
        let (lk, rk, filter) = join_keys(&self.join_condition)?;
        let (_l1, lke) = lower_all(&lk)?;
        let (_l2, rke) = lower_all(&rk)?;
        let flt = filter.as_ref().map(lower).transpose()?;
        let fe = flt.as_ref().map(|f| f.abi());
        let schema: SchemaRef = Arc::new(Schema::new(self.join_output_schema.iter().map(|c| c.to_arrow_field()).collect::<Vec<_>>())); // :136-143
        let mut left_batches = 0usize;
        let mut pending_left: Vec<RecordBatch> = vec![];
        #[for_await]
        for batch in self.left_child { pending_left.push(batch?); left_batches += 1; }
        if left_batches == 0 { return Ok(()); } // hash_join.rs:183-185
        let nleft = pending_left[0].num_columns();
        let right_dtypes: Vec<i32> = self.join_output_schema[nleft..].iter().map(|c| dtype_of(&c.desc.data_type)).collect::<Result<_, _>>()?;
        let mut j = std::ptr::null_mut();
        self.ctx.check(unsafe {
            sqlrs_hash_join_create(self.ctx.raw(), self.join_type as i32, lke.len() as i32, lke.as_ptr(), rke.as_ptr(),
                                   fe.as_ref().map_or(std::ptr::null(), |f| f as *const _), right_dtypes.len() as i32, right_dtypes.as_ptr(), &mut j)
        })?;
        let _g = Guard(j, sqlrs_hash_join_destroy);
        for batch in &pending_left { // build phase (:161-187)
            let inb = AbiBatch::new(batch)?;
            self.ctx.check(unsafe { sqlrs_hash_join_build_push(j, &inb.raw) })?;
        }
        drop(pending_left);
        self.ctx.check(unsafe { sqlrs_hash_join_build_finish(j) })?;
        #[for_await]
        for batch in self.right_child { // probe phase (:207-292): one joined batch per probe batch
            let batch = batch?;
            let inb = AbiBatch::new(&batch)?;
            let mut out = std::ptr::null_mut();
            self.ctx.check(unsafe { sqlrs_hash_join_probe_push(j, &inb.raw, SQLRS_MEM_HOST, &mut out) })?;
            if !out.is_null() { yield import_batch(schema.clone(), out)?; }
        }
        let mut tail = std::ptr::null_mut(); // unvisited left rows of Left / Full joins (:296-322)
        self.ctx.check(unsafe { sqlrs_hash_join_finish(j, SQLRS_MEM_HOST, &mut tail) })?;
        if !tail.is_null() { yield import_batch(schema, tail)?; }
    }
}

// ----------------------------------------------------------------- HashAgg --
pub struct HipHashAggExecutor {
    pub ctx: Arc<HipCtx>,
    pub agg_funcs: Vec<BoundExpr>,
    pub group_by: Vec<BoundExpr>,
    pub child: BoxedExecutor, // the Filter's child when `child_filter` is set
    /// FilterExecutor{expr, child} directly below the operator (filter.rs:7-25): handed to the library, which evaluates
    /// `column OP constant` inside its first partition pass and runs the Filter operator itself otherwise
    pub child_filter: Option<BoundExpr>,
}
fn agg_funcs_of(exprs: &[BoundExpr]) -> Vec<BoundAggFunc> {
    exprs.iter().filter_map(|e| if let BoundExpr::AggFunc(a) = e { Some(a.clone()) } else { None }).collect() // hash_agg.rs:58-63
}
impl HipHashAggExecutor {
    #[try_stream(boxed, ok = RecordBatch, error = ExecutorError)]
    pub async fn execute(self) {
        let aggs = agg_funcs_of(&self.agg_funcs);
        let (_l1, ge) = lower_all(&self.group_by)?;
        let (_l2, af) = lower_aggs(&aggs)?;
        let mut a = std::ptr::null_mut();
        self.ctx.check(unsafe { sqlrs_hash_agg_create(self.ctx.raw(), ge.len() as i32, ge.as_ptr(), af.len() as i32, af.as_ptr(), &mut a) })?;
        let _g = Guard(a, sqlrs_hash_agg_destroy);
        if let Some(p) = &self.child_filter {
            let low = lower(p)?;
            self.ctx.check(unsafe { sqlrs_hash_agg_set_filter(a, &low.abi()) })?; // (the library copies the expression)
        }
        let mut schema = None;
        #[for_await]
        for batch in self.child { // hash_agg.rs:44-122
            let batch = batch?;
            if schema.is_none() { schema = Some(agg_schema(&self.group_by, &aggs, &batch)); }
            let inb = AbiBatch::new(&batch)?;
            self.ctx.check(unsafe { sqlrs_hash_agg_push(a, &inb.raw) })?;
        }
        let mut out = std::ptr::null_mut();
        self.ctx.check(unsafe { sqlrs_hash_agg_finish(a, SQLRS_MEM_HOST, &mut out) })?; // no input: InternalError (the reference panics, :125)
        yield import_batch(schema.expect("finish succeeded, so a batch was pushed"), out)?; // :147-149
    }
}

// ------------------------------------------------------------------- Order --
/// `limit_hint`: a LimitExecutor sits directly above (PhysicalLimit(PhysicalOrder(child))) and reads only the first
/// offset + limit rows: the operator may return a prefix of the sorted result (sqlrs_order_set_limit); 0 = no hint
pub struct HipOrderExecutor { pub ctx: Arc<HipCtx>, pub order_by: Vec<BoundOrderBy>, pub child: BoxedExecutor, pub limit_hint: i64 }
impl HipOrderExecutor {
    #[try_stream(boxed, ok = RecordBatch, error = ExecutorError)]
    pub async fn execute(self) {
        let low: Vec<Lowered> = self.order_by.iter().map(|o| lower(&o.expr)).collect::<Result<_, _>>()?;
        let ob: Vec<sqlrs_order_by_t> = self.order_by.iter().zip(&low).map(|(o, l)| sqlrs_order_by_t { expr: l.abi(), asc: o.asc as i32, reserved: 0 }).collect();
        let mut h = std::ptr::null_mut();
        self.ctx.check(unsafe { sqlrs_order_create(self.ctx.raw(), ob.len() as i32, ob.as_ptr(), &mut h) })?;
        let _g = Guard(h, sqlrs_order_destroy);
        if self.limit_hint > 0 { self.ctx.check(unsafe { sqlrs_order_set_limit(h, self.limit_hint) })?; }
        let mut schema = None;
        #[for_await]
        for batch in self.child { // order.rs:19-26
            let batch = batch?;
            schema.get_or_insert(batch.schema());
            let inb = AbiBatch::new(&batch)?;
            self.ctx.check(unsafe { sqlrs_order_push(h, &inb.raw) })?;
        }
        let Some(schema) = schema else { return Ok(()) };
        let mut out = std::ptr::null_mut();
        self.ctx.check(unsafe { sqlrs_order_finish(h, SQLRS_MEM_HOST, &mut out) })?;
        yield import_batch(schema, out)?; // order.rs:66
    }
}

// -------------------------------------------- Project / Limit / SimpleAgg --
pub struct HipProjectExecutor { pub ctx: Arc<HipCtx>, pub exprs: Vec<BoundExpr>, pub child: BoxedExecutor }
/// The same operator pulling `group` batches of its child at a time and handing them to `sqlrs_project_push_many`: the same
/// stream of output batches (one per input batch, project.rs:15-27) at one upload / launch sequence / download per GROUP.
pub struct HipProjectManyExecutor { pub ctx: Arc<HipCtx>, pub exprs: Vec<BoundExpr>, pub child: BoxedExecutor, pub group: usize }
impl HipProjectManyExecutor {
    #[try_stream(boxed, ok = RecordBatch, error = ExecutorError)]
    pub async fn execute(self) {
        let (_l, ex) = lower_all(&self.exprs)?;
        let mut p = std::ptr::null_mut();
        self.ctx.check(unsafe { sqlrs_project_create(self.ctx.raw(), ex.len() as i32, ex.as_ptr(), &mut p) })?;
        let _g = Guard(p, sqlrs_project_destroy);
        let mut pending: Vec<RecordBatch> = Vec::with_capacity(self.group);
        let mut child = self.child;
        loop {
            // (advisor r05: the end of the child's stream is `None`, not a short group — testing the group's fill right after
            //  the first push ended the loop after ONE batch and dropped the rest of the stream)
            let next = futures::StreamExt::next(&mut child).await;
            let end = next.is_none();
            if let Some(b) = next { pending.push(b?); }
            if pending.len() >= self.group.max(1) || (end && !pending.is_empty()) {
                let views: Vec<AbiBatch> = pending.iter().map(AbiBatch::new).collect::<Result<_, _>>()?;
                let ins: Vec<*const sqlrs_batch_t> = views.iter().map(|v| &v.raw as *const _).collect();
                let mut outs: Vec<*mut sqlrs_batch_t> = vec![std::ptr::null_mut(); ins.len()];
                self.ctx.check(unsafe { sqlrs_project_push_many(p, ins.len() as i32, ins.as_ptr(), SQLRS_MEM_HOST, outs.as_mut_ptr()) })?;
                for (b, o) in pending.iter().zip(outs) {
                    let schema: SchemaRef = Arc::new(Schema::new(self.exprs.iter().map(|e| e.eval_field(b)).collect::<Vec<_>>()));
                    yield import_batch(schema, o)?;
                }
                pending.clear();
            }
            if end { break; }
        }
    }
}
/// ... and polled ONE batch at a time through `sqlrs_project_push_async` with `depth` tickets in flight (see
/// HipFilterAsyncExecutor; 11 -> 250 Mrows/s at 1024-row batches from the native caller).
pub struct HipProjectAsyncExecutor { pub ctx: Arc<HipCtx>, pub exprs: Vec<BoundExpr>, pub child: BoxedExecutor, pub depth: usize }
impl HipProjectAsyncExecutor {
    #[try_stream(boxed, ok = RecordBatch, error = ExecutorError)]
    pub async fn execute(self) {
        let (_l, ex) = lower_all(&self.exprs)?;
        let mut p = std::ptr::null_mut();
        self.ctx.check(unsafe { sqlrs_project_create(self.ctx.raw(), ex.len() as i32, ex.as_ptr(), &mut p) })?;
        let _g = Guard(p, sqlrs_project_destroy);
        let mut inflight = Tickets(std::collections::VecDeque::new());
        let mut child = self.child;
        let mut ended = false;
        loop {
            while !ended && inflight.0.len() <= self.depth.max(1) {
                match futures::StreamExt::next(&mut child).await {
                    None => ended = true,
                    Some(b) => {
                        let b = b?;
                        let schema: SchemaRef = Arc::new(Schema::new(self.exprs.iter().map(|e| e.eval_field(&b)).collect::<Vec<_>>()));
                        let inb = AbiBatch::new(&b)?; // read completely before push_async returns
                        let mut t = std::ptr::null_mut();
                        self.ctx.check(unsafe { sqlrs_project_push_async(p, &inb.raw, &mut t) })?;
                        inflight.0.push_back((t, schema));
                    }
                }
            }
            match inflight.0.pop_front() {
                None => break,
                Some((t, schema)) => {
                    let mut out = std::ptr::null_mut();
                    self.ctx.check(unsafe { sqlrs_batch_wait(t, &mut out) })?; // consumes the ticket, also on error
                    yield import_batch(schema, out)?;
                }
            }
        }
    }
}
impl HipProjectExecutor {
    #[try_stream(boxed, ok = RecordBatch, error = ExecutorError)]
    pub async fn execute(self) {
        let (_l, ex) = lower_all(&self.exprs)?;
        let mut p = std::ptr::null_mut();
        self.ctx.check(unsafe { sqlrs_project_create(self.ctx.raw(), ex.len() as i32, ex.as_ptr(), &mut p) })?;
        let _g = Guard(p, sqlrs_project_destroy);
        #[for_await]
        for batch in self.child { // project.rs:15-27
            let batch = batch?;
            let schema: SchemaRef = Arc::new(Schema::new(self.exprs.iter().map(|e| e.eval_field(&batch)).collect::<Vec<_>>()));
            let inb = AbiBatch::new(&batch)?;
            let mut out = std::ptr::null_mut();
            self.ctx.check(unsafe { sqlrs_project_push(p, &inb.raw, SQLRS_MEM_HOST, &mut out) })?;
            yield import_batch(schema, out)?;
        }
    }
}
/// `CrossJoinExecutor { left_child, right_child, join_output_schema }` (cross_join.rs:8-13): what the binder makes of an
/// uncorrelated scalar subquery (binder/table/subquery.rs:120-167).
pub struct HipCrossJoinExecutor { pub ctx: Arc<HipCtx>, pub left_child: BoxedExecutor, pub right_child: BoxedExecutor, pub join_output_schema: Vec<ColumnCatalog> }
impl HipCrossJoinExecutor {
    #[try_stream(boxed, ok = RecordBatch, error = ExecutorError)]
    pub async fn execute(self) {
        let schema: SchemaRef = Arc::new(Schema::new(self.join_output_schema.iter().map(|c| c.to_arrow_field()).collect::<Vec<_>>())); // cross_join.rs:16-23
        let mut j = std::ptr::null_mut();
        self.ctx.check(unsafe { sqlrs_cross_join_create(self.ctx.raw(), &mut j) })?;
        let _g = Guard(j, sqlrs_cross_join_destroy);
        #[for_await]
        for batch in self.left_child { // cross_join.rs:30: the whole left side first
            let inb = AbiBatch::new(&batch?)?;
            self.ctx.check(unsafe { sqlrs_cross_join_build_push(j, &inb.raw) })?;
        }
        #[for_await]
        for batch in self.right_child { // cross_join.rs:38-56
            let batch = batch?;
            let inb = AbiBatch::new(&batch)?;
            let mut out = std::ptr::null_mut();
            self.ctx.check(unsafe { sqlrs_cross_join_probe_push(j, &inb.raw, SQLRS_MEM_HOST, &mut out) })?;
            if out.is_null() { continue; } // no left batch / no left row
            // the library returns the batches of this right batch as ONE batch, left row major: one slice per left row
            let whole = import_batch(schema.clone(), out)?;
            let r = batch.num_rows();
            for i in 0..(if r == 0 { 0 } else { whole.num_rows() / r }) { yield whole.slice(i * r, r); }
        }
    }
}
/// The exchange step of a partitioned plan: hash-partitions every batch of `child` on `key` and trades the partitions
/// with the other ranks (one process per GPU; `sqlrs_exchange_*`: RCCL all-to-all over xGMI on the ctx stream).  No
/// reference analogue — sqlrs is a single process — but it slots exactly where the builder instantiates the children
/// of a join / aggregate (executor/mod.rs:103-114,163-174): `left_child: exchange(visit(plan.left()), left_key)`,
/// `right_child: exchange(visit(plan.right()), right_key)`; rows with equal keys meet on one rank, so the operators
/// above run unchanged and their per-rank results are disjoint.  `ExchangeComm` is created once per process from an id
/// rank 0 made (`sqlrs_exchange_unique_id`) and handed round by whatever launched the ranks.
pub struct ExchangeComm { pub ctx: Arc<HipCtx>, pub raw: *mut sqlrs_exchange_t, pub world: usize }
unsafe impl Send for ExchangeComm {}
unsafe impl Sync for ExchangeComm {}
impl ExchangeComm {
    pub fn new(ctx: Arc<HipCtx>, unique_id: &[u8; 128], rank: usize, world: usize) -> Result<Self, ExecutorError> {
        let mut raw = std::ptr::null_mut();
        ctx.check(unsafe { sqlrs_exchange_create(ctx.raw(), unique_id.as_ptr() as *const _, rank as i32, world as i32, &mut raw) })?;
        Ok(ExchangeComm { ctx, raw, world })
    }
}
impl Drop for ExchangeComm { fn drop(&mut self) { unsafe { sqlrs_exchange_destroy(self.raw) } } }
pub struct HipExchangeExecutor { pub comm: Arc<ExchangeComm>, pub key: BoundExpr, pub child: BoxedExecutor }
impl HipExchangeExecutor {
    #[try_stream(boxed, ok = RecordBatch, error = ExecutorError)]
    pub async fn execute(self) {
        let ctx = self.comm.ctx.clone();
        let key = lower(&self.key)?;
        #[for_await]
        for batch in self.child {
            let batch = batch?;
            let inb = AbiBatch::new(&batch)?;
            let w = self.comm.world;
            let (mut parts, mut offs) = (std::ptr::null_mut(), vec![0i64; w + 1]);
            ctx.check(unsafe { sqlrs_hash_partition(ctx.raw(), &inb.raw, &key.abi(), w as i32, SQLRS_MEM_DEVICE, &mut parts, offs.as_mut_ptr()) })?;
            let rows: Vec<i64> = (0..w).map(|p| offs[p + 1] - offs[p]).collect();
            let mut got = std::ptr::null_mut();
            let st = unsafe { sqlrs_exchange_all_to_all(self.comm.raw, parts, offs.as_ptr(), rows.as_ptr(), &mut got, std::ptr::null_mut()) };
            unsafe { sqlrs_batch_release(parts) }; // (read stream-ordered: the pool hands the block to LATER work of the same stream)
            ctx.check(st)?;
            let mut host = std::ptr::null_mut();
            let st = unsafe { sqlrs_batch_copy(ctx.raw(), got, SQLRS_MEM_HOST, &mut host) };
            unsafe { sqlrs_batch_release(got) };
            ctx.check(st)?;
            yield import_batch(batch.schema(), host)?; // every rank yields the rows whose key hashes to it
        }
    }
}
pub struct HipLimitExecutor { pub ctx: Arc<HipCtx>, pub limit: Option<usize>, pub offset: Option<usize>, pub child: BoxedExecutor }
impl HipLimitExecutor {
    #[try_stream(boxed, ok = RecordBatch, error = ExecutorError)]
    pub async fn execute(self) {
        if self.limit == Some(0) { return Ok(()); } // limit.rs:29-31
        let mut l = std::ptr::null_mut();
        self.ctx.check(unsafe {
            sqlrs_limit_create(self.ctx.raw(), self.limit.is_some() as i32, self.limit.unwrap_or(0) as i64,
                               self.offset.is_some() as i32, self.offset.unwrap_or(0) as i64, &mut l)
        })?;
        let _g = Guard(l, sqlrs_limit_destroy);
        #[for_await]
        for batch in self.child { // limit.rs:33-79
            let batch = batch?;
            let inb = AbiBatch::new(&batch)?;
            let (mut out, mut done) = (std::ptr::null_mut(), 0);
            self.ctx.check(unsafe { sqlrs_limit_push(l, &inb.raw, SQLRS_MEM_HOST, &mut out, &mut done) })?;
            if !out.is_null() { yield import_batch(batch.schema(), out)?; }
            if done != 0 { break; } // :76-78
        }
    }
}
pub struct HipSimpleAggExecutor { pub ctx: Arc<HipCtx>, pub agg_funcs: Vec<BoundExpr>, pub child: BoxedExecutor }
impl HipSimpleAggExecutor {
    #[try_stream(boxed, ok = RecordBatch, error = ExecutorError)]
    pub async fn execute(self) {
        let aggs = agg_funcs_of(&self.agg_funcs);
        let (_l, af) = lower_aggs(&aggs)?;
        let mut a = std::ptr::null_mut();
        self.ctx.check(unsafe { sqlrs_simple_agg_create(self.ctx.raw(), af.len() as i32, af.as_ptr(), &mut a) })?;
        let _g = Guard(a, sqlrs_simple_agg_destroy);
        let mut schema = None;
        #[for_await]
        for batch in self.child { // simple_agg.rs:35-57
            let batch = batch?;
            if schema.is_none() { schema = Some(agg_schema(&[], &aggs, &batch)); }
            let inb = AbiBatch::new(&batch)?;
            self.ctx.check(unsafe { sqlrs_simple_agg_push(a, &inb.raw) })?;
        }
        let mut out = std::ptr::null_mut();
        self.ctx.check(unsafe { sqlrs_simple_agg_finish(a, SQLRS_MEM_HOST, &mut out) })?;
        let Some(schema) = schema else { unsafe { sqlrs_batch_release(out) }; return Ok(()) };
        yield import_batch(schema, out)?; // simple_agg.rs:59-65: exactly one row
    }
}

// --------------------------- HashAgg directly over an Inner HashJoin (rewrite) --
/// `PhysicalHashAgg(PhysicalHashJoin[Inner, no join filter](left, PhysicalFilter?(right)))` as ONE operator
/// (`sqlrs_join_agg_*`): same batches as the three executors back to back; the library takes its fused route
/// (no joined batch, Filter evaluated inside the first partition pass) when the plan's data allow it.
pub struct HipHashJoinAggExecutor {
    pub ctx: Arc<HipCtx>,
    pub left_child: BoxedExecutor,
    pub right_child: BoxedExecutor, // the Filter's child when `probe_filter` is set
    pub join_condition: JoinCondition,
    pub join_output_schema: Vec<ColumnCatalog>,
    pub agg_funcs: Vec<BoundExpr>,
    pub group_by: Vec<BoundExpr>,
    pub probe_filter: Option<BoundExpr>,
}
impl HipHashJoinAggExecutor {
    #[try_stream(boxed, ok = RecordBatch, error = ExecutorError)]
    pub async fn execute(self) {
        let (lk, rk, _) = join_keys(&self.join_condition)?;
        let (_l1, lke) = lower_all(&lk)?;
        let (_l2, rke) = lower_all(&rk)?;
        let aggs = agg_funcs_of(&self.agg_funcs);
        let (_l3, ge) = lower_all(&self.group_by)?;
        let (_l4, af) = lower_aggs(&aggs)?;
        let join_schema: SchemaRef = Arc::new(Schema::new(self.join_output_schema.iter().map(|c| c.to_arrow_field()).collect::<Vec<_>>()));
        let mut left: Vec<RecordBatch> = vec![];
        #[for_await]
        for batch in self.left_child { left.push(batch?); }
        if left.is_empty() { // the join yields nothing (hash_join.rs:183-185) and HashAgg over nothing panics (hash_agg.rs:125)
            return Err(ExecutorError::InternalError("HashAgg without input".into()));
        }
        let nleft = left[0].num_columns();
        let right_dtypes: Vec<i32> = self.join_output_schema[nleft..].iter().map(|c| dtype_of(&c.desc.data_type)).collect::<Result<_, _>>()?;
        let mut ja = std::ptr::null_mut();
        self.ctx.check(unsafe {
            sqlrs_join_agg_create(self.ctx.raw(), lke.len() as i32, lke.as_ptr(), rke.as_ptr(), nleft as i32, right_dtypes.len() as i32,
                                  right_dtypes.as_ptr(), ge.len() as i32, ge.as_ptr(), af.len() as i32, af.as_ptr(), &mut ja)
        })?;
        let _g = Guard(ja, sqlrs_join_agg_destroy);
        if let Some(p) = &self.probe_filter {
            let low = lower(p)?;
            self.ctx.check(unsafe { sqlrs_join_agg_set_probe_filter(ja, &low.abi()) })?; // (the library copies the expression)
        }
        for batch in &left {
            let inb = AbiBatch::new(batch)?;
            self.ctx.check(unsafe { sqlrs_join_agg_build_push(ja, &inb.raw) })?;
        }
        drop(left);
        self.ctx.check(unsafe { sqlrs_join_agg_build_finish(ja) })?;
        #[for_await]
        for batch in self.right_child {
            let batch = batch?;
            let inb = AbiBatch::new(&batch)?;
            self.ctx.check(unsafe { sqlrs_join_agg_probe_push(ja, &inb.raw) })?;
        }
        let mut out = std::ptr::null_mut();
        self.ctx.check(unsafe { sqlrs_join_agg_finish(ja, SQLRS_MEM_HOST, &mut out) })?;
        // eval_field only reads the schema of its input: an empty batch of the join's output schema will do
        let schema = agg_schema(&self.group_by, &aggs, &RecordBatch::new_empty(join_schema));
        yield import_batch(schema, out)?;
    }
}

// ------------------------------------------- the visit_physical_* bodies --
/// `ExecutorBuilder` gains one field, `hip: Arc<HipCtx>` (src/executor/mod.rs:36-43); these bodies replace the
/// ones at mod.rs:103-114, 127-137, 139-149, 151-161, 163-174, 176-187, 189-199.
impl ExecutorBuilder {
    pub fn hip_visit_physical_filter(&mut self, plan: &PhysicalFilter) -> Option<BoxedExecutor> {
        Some(HipFilterExecutor { ctx: self.hip.clone(), expr: plan.logical().expr(), child: self.visit(plan.children().first().unwrap().clone()).unwrap() }.execute())
    }
    pub fn hip_visit_physical_hash_join(&mut self, plan: &PhysicalHashJoin) -> Option<BoxedExecutor> {
        Some(HipHashJoinExecutor {
            ctx: self.hip.clone(),
            left_child: self.visit(plan.left()).unwrap(),
            right_child: self.visit(plan.right()).unwrap(),
            join_type: plan.join_type(),
            join_condition: plan.join_condition(),
            join_output_schema: plan.join_output_columns(),
        }.execute())
    }
    /// HashAgg, with the two peepholes that reach the fused forms from a reference-shaped plan:
    /// HashAgg(HashJoin[Inner](l, Filter?(r))) -> sqlrs_join_agg_*, HashAgg(Filter(c)) -> sqlrs_hash_agg_set_filter
    pub fn hip_visit_physical_hash_agg(&mut self, plan: &PhysicalHashAgg) -> Option<BoxedExecutor> {
        let child: PlanRef = plan.children().first().unwrap().clone();
        if let Some(join) = child.as_physical_hash_join() {
            if let (JoinType::Inner, JoinCondition::On { filter: None, .. }) = (join.join_type(), join.join_condition()) {
                let (right_child, probe_filter) = match join.right().as_physical_filter() {
                    // FilterExecutor directly below the probe side: handed to the operator (filter.rs:13-25)
                    Some(f) => (self.visit(f.children().first().unwrap().clone()).unwrap(), Some(f.logical().expr())),
                    None => (self.visit(join.right()).unwrap(), None),
                };
                return Some(HipHashJoinAggExecutor {
                    ctx: self.hip.clone(),
                    left_child: self.visit(join.left()).unwrap(),
                    right_child,
                    join_condition: join.join_condition(),
                    join_output_schema: join.join_output_columns(),
                    agg_funcs: plan.logical().agg_funcs(),
                    group_by: plan.logical().group_by(),
                    probe_filter,
                }.execute());
            }
        }
        let (child, child_filter) = match child.as_physical_filter() {
            Some(f) => (self.visit(f.children().first().unwrap().clone()).unwrap(), Some(f.logical().expr())),
            None => (self.visit(child).unwrap(), None),
        };
        Some(HipHashAggExecutor { ctx: self.hip.clone(), agg_funcs: plan.logical().agg_funcs(), group_by: plan.logical().group_by(), child, child_filter }.execute())
    }
    pub fn hip_visit_physical_order(&mut self, plan: &PhysicalOrder) -> Option<BoxedExecutor> {
        Some(HipOrderExecutor { ctx: self.hip.clone(), order_by: plan.logical().order_by(), child: self.visit(plan.children().first().unwrap().clone()).unwrap(), limit_hint: 0 }.execute())
    }
    pub fn hip_visit_physical_project(&mut self, plan: &PhysicalProject) -> Option<BoxedExecutor> {
        Some(HipProjectExecutor { ctx: self.hip.clone(), exprs: plan.logical().exprs(), child: self.visit(plan.children().first().unwrap().clone()).unwrap() }.execute())
    }
    pub fn hip_visit_physical_cross_join(&mut self, plan: &PhysicalCrossJoin) -> Option<BoxedExecutor> { // executor/mod.rs:116-125
        Some(HipCrossJoinExecutor { ctx: self.hip.clone(), left_child: self.visit(plan.left()).unwrap(), right_child: self.visit(plan.right()).unwrap(),
                                    join_output_schema: plan.join_output_columns() }.execute())
    }
    pub fn hip_visit_physical_limit(&mut self, plan: &PhysicalLimit) -> Option<BoxedExecutor> {
        // both bounds are Constants in the reference (limit.rs:14-27); the binder has already folded them
        let as_usize = |e: Option<BoundExpr>| e.and_then(|e| match e { BoundExpr::Constant(v) => v.as_usize(), e => unreachable!("expr: {:?} not allowed in limit", e) });
        let (limit, offset) = (as_usize(plan.logical().limit()), as_usize(plan.logical().offset()));
        let below = plan.children().first().unwrap().clone();
        // peephole: PhysicalLimit(PhysicalOrder(x)) hands offset + limit down to the sort (ORDER BY ... LIMIT k)
        let child = match (below.as_physical_order(), limit) {
            (Ok(order), Some(l)) => HipOrderExecutor { ctx: self.hip.clone(), order_by: order.logical().order_by(),
                                                      child: self.visit(order.children().first().unwrap().clone()).unwrap(),
                                                      limit_hint: (l + offset.unwrap_or(0)) as i64 }.execute(),
            _ => self.visit(below).unwrap(),
        };
        Some(HipLimitExecutor { ctx: self.hip.clone(), limit, offset, child }.execute())
    }
    pub fn hip_visit_physical_simple_agg(&mut self, plan: &PhysicalSimpleAgg) -> Option<BoxedExecutor> {
        Some(HipSimpleAggExecutor { ctx: self.hip.clone(), agg_funcs: plan.logical().agg_funcs(), child: self.visit(plan.children().first().unwrap().clone()).unwrap() }.execute())
    }
}
