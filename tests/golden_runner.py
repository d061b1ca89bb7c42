"""Runs the physical plans of tests/golden/reference_goldens.json on an abi.Backend
(the CPU oracle, or the HIP library) through the operator classes of sqlrs_amd.executor.

Project / Limit / scan are metadata-only plumbing in the reference (project.rs:13-28,
limit.rs:12-80, table_scan.rs:16-34): they stay on the host here.
SimpleAgg (simple_agg.rs:27-65) is run as a HashAgg over one constant key with the key
column dropped — same accumulators, one group.
"""
import json
import os

import pyarrow as pa

from sqlrs_amd import abi
from sqlrs_amd.executor import (FilterExecutor, HashAggExecutor, HashJoinExecutor, OrderExecutor,
                                eval_column)
from sqlrs_amd.expr import (AggFunc, BinaryOp, BoundExpr, Constant, InputRef, JoinCondition,
                            OrderBy, TypeCast)

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden", "reference_goldens.json")

_TYPES = {"int32": pa.int32(), "int64": pa.int64(), "float64": pa.float64(), "utf8": pa.string(),
          "bool": pa.bool_()}
_DT = {"int32": abi.INT32, "int64": abi.INT64, "float64": abi.FLOAT64, "utf8": abi.UTF8,
       "bool": abi.BOOLEAN}


def load():
    with open(GOLDEN) as f:
        return json.load(f)


def table_batches(tdef):
    schema = pa.schema([(n, _TYPES[t]) for n, t in tdef["schema"]])
    out = []
    for rows in tdef["batches"]:
        cols = [pa.array([r[i] for r in rows], type=schema.field(i).type) for i in range(len(schema))]
        out.append(pa.RecordBatch.from_arrays(cols, schema=schema))
    return out


def build_expr(e) -> BoundExpr:
    head = e[0]
    if head == "ref":
        return InputRef(e[1])
    if head == "const":
        return Constant(e[1], _DT[e[2]])
    if head == "cast":
        return TypeCast(build_expr(e[1]), _DT[e[2]])
    return BinaryOp(head, build_expr(e[1]), build_expr(e[2]))


def _rename(batches, prefix):
    return batches


class Runner:
    def __init__(self, backend: abi.Backend, fixture=None):
        self.be = backend
        self.fx = fixture or load()

    def run(self, node):
        """-> list of pyarrow RecordBatch (the operator's output stream)."""
        op = node["op"]
        be = self.be
        if op == "scan":
            return table_batches(self.fx["tables"][node["table"]])
        if op == "filter":
            return list(FilterExecutor(be, build_expr(node["expr"]), self.run(node["child"])).execute())
        if op == "project":
            out = []
            for b in self.run(node["child"]):
                cols = [eval_column(be, build_expr(e), b).column(0) for e in node["exprs"]]
                out.append(pa.RecordBatch.from_arrays(cols, names=[f"p{i}" for i in range(len(cols))]))
            return out
        if op == "limit":
            t = pa.Table.from_batches(self.run(node["child"]))
            t = t.slice(node["offset"], node["limit"])
            return t.combine_chunks().to_batches() if t.num_rows else []
        if op == "hash_join":
            left, right = self.run(node["left"]), self.run(node["right"])
            ls = left[0].schema if left else pa.schema([])
            rs = right[0].schema if right else pa.schema([])
            schema = pa.schema([pa.field(f"l.{f.name}", f.type) for f in ls] +
                               [pa.field(f"r.{f.name}", f.type) for f in rs])
            cond = JoinCondition([(build_expr(l), build_expr(r)) for l, r in node["on"]],
                                 build_expr(node["filter"]) if node.get("filter") else None)
            ex = HashJoinExecutor(be, left, right, node["join_type"], cond, schema, len(ls))
            return list(ex.execute())
        if op in ("hash_agg", "simple_agg"):
            aggs = [AggFunc(a["func"], build_expr(a["expr"]), _DT[a["return_type"]],
                            bool(a.get("distinct", False))) for a in node["aggs"]]
            if op == "hash_agg":
                gb = [build_expr(e) for e in node["group_by"]]
                return list(HashAggExecutor(be, aggs, gb, self.run(node["child"])).execute())
            out = list(HashAggExecutor(be, aggs, [Constant(0, abi.INT64)], self.run(node["child"])).execute())
            return [b.select(list(range(1, b.num_columns))) for b in out]
        if op == "order":
            ob = [OrderBy(build_expr(e), bool(asc)) for e, asc in node["order_by"]]
            return list(OrderExecutor(be, ob, self.run(node["child"])).execute())
        raise ValueError(op)

    def rows(self, node):
        out = []
        for b in self.run(node):
            cols = [b.column(i).to_pylist() for i in range(b.num_columns)]
            out.extend([list(r) for r in zip(*cols)] if cols else [])
        return out
