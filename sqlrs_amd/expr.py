"""Bound expressions of the executor hot path and their postfix ABI encoding.

Mirrors the ``BoundExpr`` variants the reference evaluates on this path
(src/binder/expression/mod.rs:18-27, src/executor/evaluator.rs:13-28) and the
aggregate / order-by / join-condition descriptors that parameterise the operators:

* ``BoundInputRef{index, return_type}``      src/binder/expression/mod.rs:131-135
* ``BoundBinaryOp{op, left, right}``         src/binder/expression/binary_op.rs
* ``BoundTypeCast{expr, cast_type}``         evaluator.rs:23
* ``BoundAggFunc{func, exprs, return_type, distinct}``  src/binder/expression/agg_func.rs:29-34
* ``BoundOrderBy{expr, asc}``                src/binder/statement/mod.rs:26-29
* ``JoinCondition::On{on, filter}``          src/binder/table/join.rs:40-48
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

from . import abi

_BINOPS = {
    "+": abi.EXPR_PLUS, "-": abi.EXPR_MINUS, "*": abi.EXPR_MULTIPLY, "/": abi.EXPR_DIVIDE,
    ">": abi.EXPR_GT, "<": abi.EXPR_LT, ">=": abi.EXPR_GTEQ, "<=": abi.EXPR_LTEQ,
    "=": abi.EXPR_EQ, "!=": abi.EXPR_NOTEQ, "<>": abi.EXPR_NOTEQ, "and": abi.EXPR_AND,
    "or": abi.EXPR_OR,
}


class BoundExpr:
    def nodes(self) -> List[abi.ExprNode]:
        raise NotImplementedError

    def pack(self) -> abi.PackedExpr:
        return abi.PackedExpr(self.nodes())

    # small DSL so tests read like SQL
    def _bin(self, op, other):
        return BinaryOp(op, self, other if isinstance(other, BoundExpr) else Constant.of(other))

    def __gt__(self, o): return self._bin(">", o)
    def __lt__(self, o): return self._bin("<", o)
    def __ge__(self, o): return self._bin(">=", o)
    def __le__(self, o): return self._bin("<=", o)
    def eq(self, o): return self._bin("=", o)
    def ne(self, o): return self._bin("!=", o)
    def __add__(self, o): return self._bin("+", o)
    def __sub__(self, o): return self._bin("-", o)
    def __mul__(self, o): return self._bin("*", o)
    def __truediv__(self, o): return self._bin("/", o)
    def __and__(self, o): return self._bin("and", o)
    def __or__(self, o): return self._bin("or", o)


@dataclass
class InputRef(BoundExpr):
    index: int

    def nodes(self):
        return [abi.ExprNode(abi.EXPR_INPUT_REF, 0, self.index, 0, 0, 0.0, None)]


def build_bound_input_ref(index: int) -> InputRef:
    """Same helper name as the reference's test util (src/binder/mod.rs:400-413)."""
    return InputRef(index)


@dataclass
class Constant(BoundExpr):
    """``BoundExpr::Constant(ScalarValue)``; ``value=None`` is ``ScalarValue::X(None)``."""
    value: object
    dtype: int

    @staticmethod
    def of(v) -> "Constant":
        if isinstance(v, bool):
            return Constant(v, abi.BOOLEAN)
        if isinstance(v, int):
            return Constant(v, abi.INT64)
        if isinstance(v, float):
            return Constant(v, abi.FLOAT64)
        if isinstance(v, str):
            return Constant(v, abi.UTF8)
        raise TypeError(f"unsupported constant {v!r}")

    def nodes(self):
        n = abi.ExprNode(abi.EXPR_CONSTANT, self.dtype, 0, int(self.value is None), 0, 0.0, None)
        if self.value is not None:
            if self.dtype == abi.FLOAT64:
                n.f = float(self.value)
            elif self.dtype == abi.UTF8:
                n.s = str(self.value).encode()
            else:
                n.i = int(self.value)
        return [n]


@dataclass
class BinaryOp(BoundExpr):
    op: str
    left: BoundExpr
    right: BoundExpr

    def nodes(self):
        return self.left.nodes() + self.right.nodes() + [
            abi.ExprNode(_BINOPS[self.op.lower()], 0, 0, 0, 0, 0.0, None)]


@dataclass
class TypeCast(BoundExpr):
    expr: BoundExpr
    cast_type: int

    def nodes(self):
        return self.expr.nodes() + [abi.ExprNode(abi.EXPR_TYPE_CAST, self.cast_type, 0, 0, 0, 0.0, None)]


@dataclass
class Alias(BoundExpr):
    """``BoundExpr::Alias`` evaluates its inner expression (evaluator.rs:25)."""
    expr: BoundExpr
    name: str = ""

    def nodes(self):
        return self.expr.nodes()


_AGG = {"count": abi.AGG_COUNT, "sum": abi.AGG_SUM, "min": abi.AGG_MIN, "max": abi.AGG_MAX}


@dataclass
class AggFunc:
    """``BoundAggFunc`` (agg_func.rs:29-34).  ``return_type`` follows the binder's rule
    (agg_func.rs:54-80): COUNT -> Int64, SUM/MIN/MAX -> argument type."""
    func: str
    expr: BoundExpr
    return_type: int
    distinct: bool = False

    def abi_struct(self, keep: list) -> abi.AggFunc:
        p = self.expr.pack()
        keep.append(p)
        return abi.AggFunc(_AGG[self.func.lower()], int(self.distinct), self.return_type, 0, p.abi)

    @property
    def display(self) -> str:  # evaluator.rs:52-56  "{func}({inner_name})"
        return self.func.capitalize()


@dataclass
class OrderBy:
    expr: BoundExpr
    asc: bool = True


@dataclass
class JoinCondition:
    """``JoinCondition::On { on, filter }`` (join.rs:40-48)."""
    on: List[Tuple[BoundExpr, BoundExpr]]
    filter: Optional[BoundExpr] = None
