"""The `combine` exchange strategy of `bench.py --gpus W`, one rank's device work emulated on ONE GPU and
timed phase by phase — everything but the wire: (1) HashAgg(Filter(fact slice)) by key = the rank's partial aggregates,
(2) hash partition of the partials W ways, (3) the merge on the owning rank: HashJoinAgg(dim partition, the partials that
W ranks send to partition 0).  The other ranks' partials for (3) are computed from their own fact slices the same way.
  python tools/c5_combine_rank.py [W ...]       (default 2 4 8)"""
import os as _os; _os.environ.setdefault("SQLRS_HOOKS", "1")  # the SQLRS_* tuning hooks are consulted only in a process that opts in (common.hpp: hook)
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, sqlrs_amd
from sqlrs_amd import abi, datagen
from sqlrs_amd.expr import AggFunc, Constant, InputRef

dev = torch.device("cuda", 0)
be = sqlrs_amd.new_ctx(0)
n_fact, n_dim = int(float(os.environ.get("SQLRS_BENCH_ROWS", 1e9))), int(float(os.environ.get("SQLRS_BENCH_DIM", 1e7)))
worlds = [int(a) for a in sys.argv[1:]] or [2, 4, 8]
fact_key = datagen.fill_chunks(torch.empty(n_fact, dtype=torch.int64, device=dev), lambda i: datagen.key_t(0xF1, i, n_dim))
fact_val = datagen.fill_chunks(torch.empty(n_fact, dtype=torch.float64, device=dev), lambda i: datagen.val_t(0xF2, i))
dim_key = datagen.fill_chunks(torch.empty(n_dim, dtype=torch.int64, device=dev), lambda i: datagen.dim_key_t(i, n_dim))
torch.cuda.synchronize()
pred = (InputRef(1) > Constant(0.5, abi.FLOAT64)).pack()
gb, _k = abi.pack_exprs([InputRef(0)])
keep = []
p_aggs = (abi.AggFunc * 2)(AggFunc("count", InputRef(1), abi.INT64).abi_struct(keep), AggFunc("sum", InputRef(1), abi.FLOAT64).abi_struct(keep))
m_aggs = (abi.AggFunc * 2)(AggFunc("sum", InputRef(2), abi.INT64).abi_struct(keep), AggFunc("sum", InputRef(3), abi.FLOAT64).abi_struct(keep))
lk, _k1 = abi.pack_exprs([InputRef(0)]); rk, _k2 = abi.pack_exprs([InputRef(0)]); mgb, _k3 = abi.pack_exprs([InputRef(0)])
rd = (C.c_int32 * 3)(abi.INT64, abi.INT64, abi.FLOAT64)
T3 = (torch.int64, torch.int64, torch.float64)

def timed(fn, reps=4):
    fn(); be.synchronize(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        out = fn()
    be.synchronize()
    return (time.perf_counter() - t) * 1e3 / reps, out

def partial_of(lo, hi, keep_out=True):
    a = C.c_void_p()
    be.check(be.fn("hash_agg_create")(be.ctx, 1, gb, 2, p_aggs, C.byref(a)))
    be.check(be.fn("hash_agg_set_group_order")(a, abi.GROUP_ORDER_ANY))
    be.check(be.fn("hash_agg_set_filter")(a, C.byref(pred.abi)))
    fb = bench.device_batch(abi, [fact_key[lo:hi], fact_val[lo:hi]], [abi.INT64, abi.FLOAT64])
    be.check(be.fn("hash_agg_push")(a, fb.ptr))
    po = C.POINTER(abi.Batch)()
    be.check(be.fn("hash_agg_finish")(a, abi.MEM_DEVICE, C.byref(po)))
    be.fn("hash_agg_destroy")(a)
    w = be.wrap(po)
    if not keep_out:
        w.release()
        return None
    return w

for W in worlds:
    sl = n_fact // W
    ms_partial, part0 = timed(lambda: partial_of(0, sl, keep_out=False))
    part = partial_of(0, sl)
    g = part.num_rows
    cols = [bench._tensor_view(torch, part.column(i).values, g, t, dev) for i, t in enumerate(T3)]
    def do_partition():
        pb = bench.device_batch(abi, cols, [abi.INT64, abi.INT64, abi.FLOAT64])
        parts, offs = be.hash_partition(pb, InputRef(0), W, abi.MEM_DEVICE)
        parts.release()
        return offs
    ms_part, offs = timed(do_partition)
    # what partition 0 receives: its slice of every rank's partials
    recv = [[], [], []]
    for r in range(W):
        pr = part if r == 0 else partial_of(r * sl, (r + 1) * sl)
        gr = pr.num_rows
        cr = [bench._tensor_view(torch, pr.column(i).values, gr, t, dev) for i, t in enumerate(T3)]
        be.synchronize()
        parts, o = be.hash_partition(bench.device_batch(abi, cr, [abi.INT64, abi.INT64, abi.FLOAT64]), InputRef(0), W, abi.MEM_DEVICE)
        be.synchronize()
        for i, t in enumerate(T3):
            recv[i].append(bench._tensor_view(torch, parts.column(i).values, parts.column(i).length, t, dev)[o[0]:o[1]].clone())
        torch.cuda.synchronize()
        parts.release()
        if r:
            pr.release()
    rk_, rc_, rs_ = [torch.cat(x) for x in recv]
    dparts, doffs = be.hash_partition(bench.device_batch(abi, [dim_key], [abi.INT64]), InputRef(0), W, abi.MEM_DEVICE)
    be.synchronize()
    dk = bench._tensor_view(torch, dparts.column(0).values, dparts.column(0).length, torch.int64, dev)[doffs[0]:doffs[1]].clone()
    torch.cuda.synchronize()
    dparts.release()
    def merge():
        ja = C.c_void_p()
        be.check(be.fn("join_agg_create")(be.ctx, 1, lk, rk, 1, 3, rd, 1, mgb, 2, m_aggs, C.byref(ja)))
        be.check(be.fn("join_agg_set_group_order")(ja, abi.GROUP_ORDER_ANY))
        db = bench.device_batch(abi, [dk], [abi.INT64])
        mb = bench.device_batch(abi, [rk_, rc_, rs_], [abi.INT64, abi.INT64, abi.FLOAT64])
        be.check(be.fn("join_agg_build_push")(ja, db.ptr)); be.check(be.fn("join_agg_build_finish")(ja))
        be.check(be.fn("join_agg_probe_push")(ja, mb.ptr))
        ao = C.POINTER(abi.Batch)()
        be.check(be.fn("join_agg_finish")(ja, abi.MEM_DEVICE, C.byref(ao)))
        fused = be.fn("join_agg_fused_batches")(ja)
        be.fn("join_agg_destroy")(ja)
        w = be.wrap(ao); n_out = w.num_rows; w.release()
        return n_out, fused
    ms_merge, (groups, fused) = timed(merge)
    part.release()
    bytes_out = 24 * g * (W - 1) // W
    print(W, json.dumps({"fact_rows_per_rank": sl, "partial_groups": g, "partial_agg_ms": round(ms_partial, 3), "partition_ms": round(ms_part, 3),
                         "merge_rows": int(rk_.numel()), "merge_groups": groups, "merge_fused": bool(fused), "merge_ms": round(ms_merge, 3),
                         "device_ms_total": round(ms_partial + ms_part + ms_merge, 3), "bytes_off_rank": bytes_out,
                         "wire_ms_at_50GBps_per_link": round(bytes_out / max(W - 1, 1) / 50e6, 3)}), flush=True)
    del rk_, rc_, rs_, dk
