"""Runs the physical plans of tests/golden/reference_goldens.json on an abi.Backend
(the CPU oracle, or the HIP library) through the operator classes of sqlrs_amd.executor.

Every operator of a plan — Filter, HashJoin, HashAgg, Order and the plumbing around them
(Project project.rs:13-28, Limit limit.rs:12-80, SimpleAgg simple_agg.rs:27-65) — runs through
the backend's C ABI; only the table scan is host data (table_scan.rs:16-34).  `text()` renders the
output with the backend's record_batch_to_string (util/mod.rs:53-80), the form the reference's
sqllogictest harness compares; `render_rows()` is that rule applied to a golden's typed rows.
"""
import json
import os

import pyarrow as pa

from sqlrs_amd import abi
from sqlrs_amd.executor import (CrossJoinExecutor, FilterExecutor, HashAggExecutor, HashJoinExecutor, LimitExecutor, OrderExecutor,
                                ProjectExecutor, SimpleAggExecutor)
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


def render_rows(rows) -> str:
    """record_batch_to_string (util/mod.rs:53-80) over typed rows: NULL, (empty), Rust Display"""
    def cell(v):
        if v is None:
            return "NULL"
        if isinstance(v, bool):
            return "true" if v else "false"
        if isinstance(v, float):
            return str(int(v)) if v == int(v) and abs(v) < 1e15 else repr(v)
        if isinstance(v, str):
            return v if v else "(empty)"
        return str(v)
    return "".join(" ".join(cell(v) for v in r) + "\n" for r in rows)


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
            exprs = [build_expr(e) for e in node["exprs"]]
            return list(ProjectExecutor(be, exprs, self.run(node["child"]),
                                        output_names=[f"p{i}" for i in range(len(exprs))]).execute())
        if op == "limit":
            return list(LimitExecutor(be, node.get("limit"), node.get("offset"), self.run(node["child"])).execute())
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
        if op == "cross_join":
            left, right = self.run(node["left"]), self.run(node["right"])
            ls = left[0].schema if left else pa.schema([])
            rs = right[0].schema if right else pa.schema([])
            schema = pa.schema([pa.field(f"l.{f.name}", f.type) for f in ls] + [pa.field(f"r.{f.name}", f.type) for f in rs])
            return list(CrossJoinExecutor(be, left, right, schema).execute())
        if op in ("hash_agg", "simple_agg"):
            aggs = [AggFunc(a["func"], build_expr(a["expr"]), _DT[a["return_type"]],
                            bool(a.get("distinct", False))) for a in node["aggs"]]
            if op == "hash_agg":
                gb = [build_expr(e) for e in node["group_by"]]
                return list(HashAggExecutor(be, aggs, gb, self.run(node["child"])).execute())
            return list(SimpleAggExecutor(be, aggs, self.run(node["child"])).execute())
        if op == "order":
            ob = [OrderBy(build_expr(e), bool(asc)) for e, asc in node["order_by"]]
            return list(OrderExecutor(be, ob, self.run(node["child"])).execute())
        raise ValueError(op)

    def text(self, node) -> str:
        """the plan's output in the reference's sqllogictest text form"""
        return "".join(self.be.batch_to_string(b) for b in self.run(node))

    def types(self, node):
        """arrow type names of the output columns (the first output batch's schema)"""
        bs = self.run(node)
        return [str(f.type) for f in bs[0].schema] if bs else None

    def rows(self, node):
        out = []
        for b in self.run(node):
            cols = [b.column(i).to_pylist() for i in range(b.num_columns)]
            out.extend([list(r) for r in zip(*cols)] if cols else [])
        return out
