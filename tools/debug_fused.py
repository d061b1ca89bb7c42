import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, pyarrow as pa
import sqlrs_amd
from sqlrs_amd import abi
from sqlrs_amd.executor import HashJoinAggExecutor, HashJoinExecutor, HashAggExecutor
from sqlrs_amd.expr import AggFunc, InputRef, JoinCondition
from oracle_backend import load_oracle
hip, oracle = sqlrs_amd.hip(0), load_oracle()
rng = np.random.default_rng(1)
nb, np_, keyrange = 1000, 140_000, 1500
lkeys = rng.permutation(keyrange)[:nb].astype(np.int64)
lb = pa.RecordBatch.from_arrays([pa.array(lkeys), pa.array(rng.random(nb))], names=["c0", "c1"])
rb = pa.RecordBatch.from_arrays([pa.array(rng.random(np_)), pa.array(rng.integers(0, keyrange, np_, dtype=np.int64))], names=["v", "k"])
cond = JoinCondition([(InputRef(0), InputRef(1))])
sch = pa.schema([("l.c0", pa.int64()), ("l.c1", pa.float64()), ("r.v", pa.float64()), ("r.k", pa.int64())])
for aggs in ([AggFunc("count", InputRef(2), abi.INT64)], [AggFunc("count", InputRef(2), abi.INT64), AggFunc("sum", InputRef(2), abi.FLOAT64)]):
    for nbatches in (1, 2):
        rbs = [rb] if nbatches == 1 else [rb.slice(0, np_ // 2), rb.slice(np_ // 2)]
        ex = HashJoinAggExecutor(hip, [lb], rbs, cond, sch, 2, aggs, [InputRef(0)])
        got = list(ex.execute())[0]
        join = HashJoinExecutor(oracle, [lb], rbs, "inner", cond, sch, 2)
        exp = list(HashAggExecutor(oracle, aggs, [InputRef(0)], join.execute()).execute())[0]
        gk, ek = np.array(got.column(0).to_pylist()), np.array(exp.column(0).to_pylist())
        print("aggs", len(aggs), "batches", nbatches, "fused", ex.fused_batches, "groups", len(gk), len(ek), "same order", bool(len(gk) == len(ek) and (gk == ek).all()),
              "same set", set(gk.tolist()) == set(ek.tolist()))
        gd = dict(zip(gk.tolist(), got.column(1).to_pylist())); ed = dict(zip(ek.tolist(), exp.column(1).to_pylist()))
        bad = [k for k in ed if gd.get(k) != ed[k]]
        print("   count mismatches:", len(bad), bad[:5], [(gd.get(k), ed[k]) for k in bad[:5]], "first got", gk[:5], "first exp", ek[:5])
print("---- single halves")
aggs = [AggFunc("count", InputRef(2), abi.INT64)]
h1, h2 = rb.slice(0, np_ // 2), rb.slice(np_ // 2)
for name, rbs in (("b1", [h1]), ("b2", [h2]), ("b2copy", [pa.RecordBatch.from_pydict(h2.to_pydict())]), ("b1,b2", [h1, h2]), ("b2,b1", [h2, h1])):
    for trial in range(2):
        ex = HashJoinAggExecutor(hip, [lb], rbs, cond, sch, 2, aggs, [InputRef(0)])
        got = list(ex.execute())[0]
        join = HashJoinExecutor(oracle, [lb], rbs, "inner", cond, sch, 2)
        exp = list(HashAggExecutor(oracle, aggs, [InputRef(0)], join.execute()).execute())[0]
        gd = dict(zip(got.column(0).to_pylist(), got.column(1).to_pylist())); ed = dict(zip(exp.column(0).to_pylist(), exp.column(1).to_pylist()))
        bad = [(k, gd.get(k), ed[k]) for k in ed if gd.get(k) != ed[k]]
        print(name, "trial", trial, "fused", ex.fused_batches, "bad", bad[:6], "sum got", sum(gd.values()), "sum exp", sum(ed.values()))
print("---- marshalling of the sliced batch")
hb = abi.HostBatch(h2)
for i, arr in enumerate(hb.arrays):
    print(i, "offset", arr.offset, "len", len(arr), "equal to slice:", arr.equals(h2.column(i)), "buf size", arr.buffers()[1].size, "addr%64", arr.buffers()[1].address % 64)
rt = hip.to_host(hip.to_device(h2)).to_arrow(["v", "k"])
print("roundtrip equal:", rt.column(0).equals(h2.column(0)), rt.column(1).equals(h2.column(1)))
k_host = np.array(h2.column(1).to_pylist()); k_rt = np.array(rt.column(1).to_pylist())
d = np.nonzero(k_host != k_rt)[0]
print("diff rows", d[:10], k_host[d[:10]], k_rt[d[:10]])
