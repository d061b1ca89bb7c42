"""Single-batch streaming without a synchronisation per call (sqlrs_filter_push_async / sqlrs_hash_join_probe_push_async /
sqlrs_batch_wait, review r05 #6): the reference's calling shape — one 1024-row batch per poll, filter.rs:15-24,
hash_join.rs:284-291, storage/csv.rs:105.  The async stream must be the synchronous stream, batch for batch."""
import numpy as np
import pyarrow as pa
import pytest

from sqlrs_amd import abi
from sqlrs_amd.executor import FilterExecutor, HashJoinExecutor
from sqlrs_amd.expr import Constant, InputRef, JoinCondition, TypeCast
from test_gpu_parity import join_schema

pytestmark = pytest.mark.gpu


def fast_batches(be):
    return be.profile_read().get("async_fast_batches", (0, 0))[1]


def same_batches(got, exp):
    assert [g.num_rows for g in got] == [e.num_rows for e in exp]
    for g, e in zip(got, exp):
        assert g.schema.names == e.schema.names
        for c in range(g.num_columns):
            assert g.column(c).equals(e.column(c)), (g.schema.names[c], g.column(c).null_count, e.column(c).null_count)


def batches(rng, n, sizes, nulls, with_str=False):
    out = []
    for rows in sizes:
        cols = [pa.array(rng.integers(-100, 100, rows), mask=(rng.random(rows) < nulls) if nulls else None),
                pa.array(rng.random(rows), mask=(rng.random(rows) < nulls) if nulls else None),
                pa.array(rng.integers(-5, 5, rows).astype(np.int32), mask=(rng.random(rows) < nulls) if nulls else None)]
        names = ["a", "b", "c"]
        if with_str:
            cols.append(pa.array([None if i % 7 == 3 else ("" if i % 5 == 0 else "s" * (i % 11) + str(i)) for i in range(rows)], type=pa.string()))
            names.append("s")
        out.append(pa.RecordBatch.from_arrays(cols, names=names))
    return out


@pytest.mark.parametrize("pred", ["i64_gt", "f64_le", "i64_ne", "pred_nulls", "compound", "i32", "three_terms", "or", "col_col", "arith",
                                  "cast", "div", "null_const", "bool_cmp"])
@pytest.mark.parametrize("depth", [1, 4, 40])
def test_filter_push_async_yields_the_batches_of_push(hip, oracle, monkeypatch, pred, depth):
    """fast path (conjunctions of up to four `column OP constant` terms over int32 / int64 / float64, a row with a NULL term
    dropped, validity re-packed), Utf8 payload columns with NULL and empty strings, the slow path inside the same stream (OR, column-to-column, > 4096 rows) and
    more tickets than ring slots (depth 40): always the batches of sqlrs_filter_push and of the oracle"""
    rng = np.random.default_rng(5 + depth)
    nulls = 0.2 if pred == "pred_nulls" else 0.1
    bs = batches(rng, 0, [1024] * 20 + [0, 1, 63, 64, 65, 1023, 1025, 4096, 4097, 20000] + [1024] * 45, nulls)
    bs += batches(rng, 0, [1024, 100, 0, 4096, 3000], nulls, with_str=True)  # (Utf8 payload columns travel through the kernel too)
    e = {"i64_gt": InputRef(0) > Constant(3, abi.INT64), "f64_le": InputRef(1) <= Constant(0.37, abi.FLOAT64),
         "i64_ne": InputRef(0).ne(Constant(-7, abi.INT64)), "pred_nulls": InputRef(1) > Constant(0.5, abi.FLOAT64),
         "compound": (InputRef(0) > Constant(3, abi.INT64)) & (InputRef(1) < Constant(0.9, abi.FLOAT64)),
         "i32": InputRef(2) > Constant(0, abi.INT32),
         "three_terms": ((InputRef(0) >= Constant(-50, abi.INT64)) & (InputRef(1) < Constant(0.8, abi.FLOAT64))) & InputRef(2).ne(Constant(1, abi.INT32)),
         "or": (InputRef(0) > Constant(90, abi.INT64)) | (InputRef(1) < Constant(0.1, abi.FLOAT64)),  # (not a conjunction: the synchronous operator)
         "col_col": InputRef(0) > InputRef(0),
         # the postfix program evaluated per row inside the kernel: arithmetic (wrapping), casts (out of range -> NULL), division,
         # NULL constants, Kleene OR, comparisons of comparison results
         "arith": ((InputRef(0) + Constant(5, abi.INT64)) * InputRef(0)) > (InputRef(0) - Constant(7, abi.INT64)),
         "cast": (TypeCast(InputRef(2), abi.INT64) * Constant(3_000_000_000, abi.INT64) + InputRef(0)).ne(Constant(0, abi.INT64))
                 & (TypeCast(TypeCast(InputRef(1), abi.INT64) * Constant(1 << 40, abi.INT64) + InputRef(0) * Constant(1 << 26, abi.INT64), abi.INT32) > Constant(-1, abi.INT32)),
         "div": (InputRef(1) / Constant(0.25, abi.FLOAT64)) > (TypeCast(InputRef(0), abi.FLOAT64) / Constant(50.0, abi.FLOAT64)),
         "null_const": (InputRef(0) > Constant(None, abi.INT64)) | (InputRef(1) < Constant(0.3, abi.FLOAT64)),
         "bool_cmp": (InputRef(0) > Constant(0, abi.INT64)).eq(InputRef(1) > Constant(0.5, abi.FLOAT64))}[pred]
    before = fast_batches(hip)
    got = list(FilterExecutor(hip, e, bs, depth=depth).execute())
    took = fast_batches(hip) - before
    # every <= 4096-row batch takes the one-launch kernel (conjunctions through the specialised path, everything else over fixed-width
    # columns through the in-kernel postfix program); with more tickets than ring slots some run the synchronous operator
    eligible = sum(1 for b in bs if b.num_rows <= 4096)
    if depth < 32:
        assert took == eligible, (took, eligible)
    else:
        assert 0 < took <= eligible
    same_batches(got, list(FilterExecutor(hip, e, bs).execute()))
    same_batches(got, list(FilterExecutor(oracle, e, bs).execute()))
    monkeypatch.setenv("SQLRS_ASYNC_FAST", "0")  # every batch through the synchronous operator inside push_async
    same_batches(list(FilterExecutor(hip, e, bs, depth=depth).execute()), got)


@pytest.mark.parametrize("keys", ["dense", "sparse", "f64", "i32"])
@pytest.mark.parametrize("depth", [1, 8])
def test_hash_join_probe_push_async_yields_the_batches_of_probe_push(hip, oracle, keys, depth):
    """Inner join over unique build keys: direct-address table, 16-byte-slot hash table, f64 keys by bit pattern, int32 keys;
    NULLs in the build and probe PAYLOAD columns travel; batches with NULL probe keys, big batches and empty ones take the
    synchronous operator inside the same stream"""
    rng = np.random.default_rng(11 + depth)
    nb = 20_000
    raw = rng.permutation(3 * nb)[:nb]
    conv = {"dense": lambda x: x.astype(np.int64), "sparse": lambda x: x.astype(np.int64) * 7919 - 5,
            "f64": lambda x: x.astype(np.float64) * 0.25, "i32": lambda x: x.astype(np.int32)}[keys]
    lb = pa.RecordBatch.from_arrays([pa.array(conv(raw)), pa.array(rng.random(nb), mask=rng.random(nb) < 0.1),
                                     pa.array(np.arange(nb, dtype=np.int32))], names=["k", "x", "i"])
    rbs = []
    for rows in [1024] * 30 + [0, 1, 64, 1023, 4096, 5000, 1024, 1024]:
        rbs.append(pa.RecordBatch.from_arrays([pa.array(rng.random(rows), mask=rng.random(rows) < 0.2),
                                               pa.array(conv(rng.integers(0, 3 * nb, rows)))], names=["v", "k"]))
    pk = conv(rng.integers(0, 3 * nb, 1024))
    rbs.insert(7, pa.RecordBatch.from_arrays([pa.array(rng.random(1024)), pa.array(pk, mask=rng.random(1024) < 0.1)], names=["v", "k"]))
    cond = JoinCondition([(InputRef(0), InputRef(1))])
    sch = join_schema(lb, rbs[0])
    before = fast_batches(hip)
    got = list(HashJoinExecutor(hip, [lb], rbs, "inner", cond, sch, 3, depth=depth).execute())
    assert fast_batches(hip) - before == sum(1 for b in rbs if b.num_rows <= 4096 and b.column(1).null_count == 0)
    same_batches(got, list(HashJoinExecutor(hip, [lb], rbs, "inner", cond, sch, 3).execute()))
    same_batches(got, list(HashJoinExecutor(oracle, [lb], rbs, "inner", cond, sch, 3).execute()))


def test_async_shapes_that_only_the_synchronous_operator_takes(hip, oracle):
    """Left join, a join filter, duplicate build keys, an empty build side: push_async runs the synchronous operator and
    parks its batch in the ticket — the stream (including the Left join's tail batch) is unchanged"""
    rng = np.random.default_rng(3)
    nb = 3000
    lb = pa.RecordBatch.from_arrays([pa.array(rng.integers(0, 2000, nb)), pa.array(rng.random(nb))], names=["k", "x"])  # duplicates
    rbs = [pa.RecordBatch.from_arrays([pa.array(rng.integers(0, 2500, 1024)), pa.array(rng.random(1024))], names=["k", "v"]) for _ in range(6)]
    sch = join_schema(lb, rbs[0])
    for jt, cond, lbs in (("inner", JoinCondition([(InputRef(0), InputRef(0))]), [lb]),
                          ("left", JoinCondition([(InputRef(0), InputRef(0))]), [lb.slice(0, 500)]),
                          ("inner", JoinCondition([(InputRef(0), InputRef(0))], InputRef(1) > InputRef(3)), [lb.slice(0, 500)]),
                          ("inner", JoinCondition([(InputRef(0), InputRef(0))]), [lb.slice(0, 0)])):
        before = fast_batches(hip)
        got = list(HashJoinExecutor(hip, lbs, rbs, jt, cond, sch, 2, depth=3).execute())
        assert fast_batches(hip) == before or lbs[0].num_rows == 0  # (a build side of one 0-row batch is unique: its probes may take the kernel)
        same_batches(got, list(HashJoinExecutor(oracle, lbs, rbs, jt, cond, sch, 2).execute()))


@pytest.mark.parametrize("group", ["1", "2", "3"])
def test_async_group_sizes(oracle, monkeypatch, group):
    """the launches of the async path carry up to four batches; a ticket waited for before its group is full flushes it — here
    with groups of 1 / 2 / 3 batches (SQLRS_ASYNC_GROUP, read when a ctx's ring is created) and depths below, at and above them"""
    import sqlrs_amd
    monkeypatch.setenv("SQLRS_ASYNC_GROUP", group)
    be = sqlrs_amd.new_ctx(0)
    try:
        rng = np.random.default_rng(int(group))
        bs = batches(rng, 0, [1024] * 11 + [5000] + [1024] * 6, 0.1)
        e = InputRef(0) > Constant(3, abi.INT64)
        exp = list(FilterExecutor(oracle, e, bs).execute())
        for depth in (1, 2, 5):
            same_batches(list(FilterExecutor(be, e, bs, depth=depth).execute()), exp)
    finally:
        be.close()


def test_async_filter_keeps_a_full_utf8_batch_whole(hip, oracle):
    """4096 rows, every one kept: the Utf8 end offset (entry 4096 of the offsets) is written too"""
    n = 4096
    b = pa.RecordBatch.from_arrays([pa.array(np.arange(n)), pa.array(["s" * (i % 11) + str(i) for i in range(n)], type=pa.string()),
                                    pa.array([None if i % 5 == 0 else "t" + str(i) for i in range(n)], type=pa.string())], names=["a", "s", "t"])
    for e in (InputRef(0) >= Constant(0, abi.INT64), (InputRef(0) + Constant(1, abi.INT64)) > Constant(0, abi.INT64)):
        before = fast_batches(hip)
        got = list(FilterExecutor(hip, e, [b, b.slice(0, 4095), b], depth=4).execute())
        assert fast_batches(hip) - before == 3
        same_batches(got, list(FilterExecutor(oracle, e, [b, b.slice(0, 4095), b]).execute()))


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_async_filter_and_probe(hip, oracle, seed):
    """random schemas (1-9 fixed-width columns, NULL rates 0 / 5 / 50 / 100 %), batch sizes 0-4096, conjunctions of 1-4 random terms;
    random Inner joins over unique build keys (dense or sparse, int32 / int64) with NULL-bearing payload columns on both sides —
    the async stream equals the oracle's, batch for batch"""
    rng = np.random.default_rng(1000 + seed)
    ncols = int(rng.integers(1, 10))
    kinds = [str(rng.choice(["i64", "f64", "i32", "str"], p=[0.3, 0.3, 0.2, 0.2])) for _ in range(ncols)]
    if all(k == "str" for k in kinds):
        kinds[0] = "i64"
    nullp = [float(rng.choice([0.0, 0.05, 0.5, 1.0], p=[0.5, 0.3, 0.15, 0.05])) for _ in range(ncols)]

    def col(kind, rows, p):
        if kind == "str":
            m = rng.random(rows) < p
            return pa.array([None if m[i] else "x" * int(rng.integers(0, 9)) + str(i) for i in range(rows)], type=pa.string())
        vals = {"i64": lambda: rng.integers(-20, 20, rows), "f64": lambda: np.round(rng.random(rows), 2),
                "i32": lambda: rng.integers(-20, 20, rows).astype(np.int32)}[kind]()
        return pa.array(vals, mask=(rng.random(rows) < p) if p else None)
    sizes = [int(x) for x in rng.choice([0, 1, 7, 64, 100, 1000, 1024, 2047, 4096], size=14)]
    bs = [pa.RecordBatch.from_arrays([col(k, n, p) for k, p in zip(kinds, nullp)], names=[f"c{i}" for i in range(ncols)]) for n in sizes]
    e = None
    for _ in range(int(rng.integers(1, 5))):
        c = int(rng.choice([i for i, k in enumerate(kinds) if k != "str"]))
        const = {"i64": Constant(int(rng.integers(-10, 10)), abi.INT64), "f64": Constant(float(np.round(rng.random(), 2)), abi.FLOAT64),
                 "i32": Constant(int(rng.integers(-10, 10)), abi.INT32)}[kinds[c]]
        op = str(rng.choice([">", "<", ">=", "<=", "=", "!="]))
        t = {">": lambda a, b: a > b, "<": lambda a, b: a < b, ">=": lambda a, b: a >= b, "<=": lambda a, b: a <= b,
             "=": lambda a, b: a.eq(b), "!=": lambda a, b: a.ne(b)}[op](InputRef(c), const)
        e = t if e is None else (e & t)
    same_batches(list(FilterExecutor(hip, e, bs, depth=int(rng.integers(1, 7))).execute()), list(FilterExecutor(oracle, e, bs).execute()))
    # ---- a random predicate TREE (arithmetic, casts, comparisons, Kleene AND / OR, NULL constants): the in-kernel postfix program
    DT = {"i64": abi.INT64, "f64": abi.FLOAT64, "i32": abi.INT32}

    def rand_num(kind, depth):
        cols_k = [i for i, k in enumerate(kinds) if k == kind]
        r = rng.random()
        if depth == 0 or r < 0.3:
            if cols_k and rng.random() < 0.7:
                return InputRef(int(rng.choice(cols_k)))
            if rng.random() < 0.1:
                return Constant(None, DT[kind])
            return Constant(float(np.round(rng.random() * 4 - 2, 2)) if kind == "f64" else int(rng.integers(-5, 6)), DT[kind])
        if r < 0.45:
            other = str(rng.choice([k for k in DT if k != kind]))
            return TypeCast(rand_num(other, depth - 1), DT[kind])
        a, b = rand_num(kind, depth - 1), rand_num(kind, depth - 1)
        return {0: lambda: a + b, 1: lambda: a - b, 2: lambda: a * b}[int(rng.integers(0, 3))]()

    def rand_bool(depth):
        if depth == 0 or rng.random() < 0.5:
            kind = str(rng.choice(list(DT)))
            a, b = rand_num(kind, 2), rand_num(kind, 2)
            return {0: lambda: a > b, 1: lambda: a < b, 2: lambda: a >= b, 3: lambda: a <= b, 4: lambda: a.eq(b), 5: lambda: a.ne(b)}[int(rng.integers(0, 6))]()
        a, b = rand_bool(depth - 1), rand_bool(depth - 1)
        return (a & b) if rng.random() < 0.5 else (a | b)
    for _ in range(3):
        e2 = rand_bool(2)
        if len(e2.nodes()) > 24:
            continue
        before = fast_batches(hip)
        got2 = list(FilterExecutor(hip, e2, bs, depth=3).execute())
        assert fast_batches(hip) - before == len(bs)
        same_batches(got2, list(FilterExecutor(oracle, e2, bs).execute()))
    # ---- join
    nb = int(rng.choice([50, 3000, 40000]))
    kk = str(rng.choice(["i64", "i32"]))
    mul = int(rng.choice([1, 7919]))
    raw = rng.permutation(3 * nb)[:nb] * mul
    kconv = (lambda x: x.astype(np.int64)) if kk == "i64" else (lambda x: x.astype(np.int32))
    lb = pa.RecordBatch.from_arrays([col("f64", nb, 0.2), pa.array(kconv(raw)), col("i32", nb, 0.1)], names=["x", "k", "y"])
    rbs = [pa.RecordBatch.from_arrays([pa.array(kconv(rng.integers(0, 3 * nb, n) * mul)), col("i64", n, 0.3), col("f64", n, 0.0)], names=["k", "v", "w"])
           for n in sizes]
    cond = JoinCondition([(InputRef(1), InputRef(0))])
    sch = join_schema(lb, rbs[0])
    same_batches(list(HashJoinExecutor(hip, [lb], rbs, "inner", cond, sch, 3, depth=int(rng.integers(1, 7))).execute()),
                 list(HashJoinExecutor(oracle, [lb], rbs, "inner", cond, sch, 3).execute()))


def test_async_filter_over_the_reference_csv_table(hip, oracle):
    """tests/csv/employee.csv of the reference (Utf8 and Int64 columns) in 3-row batches, `salary > 10000 AND id < 4` polled one
    batch at a time through push_async: the oracle's stream, and every batch took the one-launch kernel"""
    import os
    import pyarrow.csv as pacsv
    root = os.path.dirname(os.path.abspath(__file__))
    t = pacsv.read_csv(os.path.join(root, "golden", "csv", "employee.csv"))
    t = t.cast(pa.schema([pa.field(f.name, pa.int64() if pa.types.is_integer(f.type) else pa.string()) for f in t.schema]))
    rb = t.combine_chunks().to_batches()[0]
    bs = [rb.slice(i, 3) for i in range(0, rb.num_rows, 3)]
    names = rb.schema.names
    e = (InputRef(names.index("salary")) > Constant(10000, abi.INT64)) & (InputRef(names.index("id")) < Constant(4, abi.INT64))
    before = fast_batches(hip)
    got = list(FilterExecutor(hip, e, bs, depth=2).execute())
    assert fast_batches(hip) - before == len(bs)
    exp = list(FilterExecutor(oracle, e, bs).execute())
    same_batches(got, exp)
    assert sum(g.num_rows for g in got) > 0


def test_async_divide_by_zero_is_the_evaluators_error(hip, oracle):
    """x / 0 on a VALID row is Arrow's DivideByZero (array_compute.rs via evaluator.rs:19-27): the synchronous push raises it at the
    push, the async path at the wait for that batch's ticket; a NULL divisor or dividend divides nothing; the operator-side state
    survives (the next stream runs)"""
    rng = np.random.default_rng(9)
    n = 1000
    num = pa.array(rng.integers(1, 100, n))
    zero_where_null = pa.array(np.where(np.arange(n) % 10 == 0, 0, rng.integers(1, 9, n)), mask=(np.arange(n) % 10 == 0))
    zero_valid = pa.array(np.where(np.arange(n) == 777, 0, rng.integers(1, 9, n)))
    ok = pa.RecordBatch.from_arrays([num, zero_where_null], names=["a", "b"])
    bad = pa.RecordBatch.from_arrays([num, zero_valid], names=["a", "b"])
    e = (InputRef(0) / InputRef(1)) > Constant(3, abi.INT64)
    same_batches(list(FilterExecutor(hip, e, [ok, ok], depth=2).execute()), list(FilterExecutor(oracle, e, [ok, ok]).execute()))
    for be, kw in ((oracle, {}), (hip, {}), (hip, {"depth": 2})):
        with pytest.raises(abi.ExecutorError) as ei:
            list(FilterExecutor(be, e, [ok, bad, ok], **kw).execute())
        assert ei.value.status == abi.ERR_ARROW and "ivide by zero" in str(ei.value), (kw, str(ei.value))
    same_batches(list(FilterExecutor(hip, e, [ok], depth=1).execute()), list(FilterExecutor(oracle, e, [ok]).execute()))


# ---- Project ---------------------------------------------------------------------------------------------------------
def bool_batches(rng, sizes):
    out = []
    for rows in sizes:
        out.append(pa.RecordBatch.from_arrays(
            [pa.array(rng.integers(-100, 100, rows), mask=rng.random(rows) < 0.1), pa.array(rng.random(rows), mask=rng.random(rows) < 0.1),
             pa.array(rng.integers(-5, 5, rows).astype(np.int32), mask=rng.random(rows) < 0.1),
             pa.array([None if i % 7 == 3 else ("" if i % 5 == 0 else "s" * (i % 11) + str(i)) for i in range(rows)], type=pa.string()),
             pa.array(rng.random(rows) < 0.5, mask=rng.random(rows) < 0.2)], names=["a", "b", "c", "s", "t"]))
    return out


PROJECTIONS = {
    "refs": lambda: [InputRef(3), InputRef(0), InputRef(4), InputRef(1), InputRef(2), InputRef(0)],
    "arith": lambda: [InputRef(0) + Constant(1, abi.INT64), InputRef(1) * InputRef(1), InputRef(2) - Constant(3, abi.INT32), InputRef(3)],
    "bool_results": lambda: [InputRef(0) > Constant(0, abi.INT64), (InputRef(1) < Constant(0.5, abi.FLOAT64)) | (InputRef(0).eq(Constant(7, abi.INT64))),
                             InputRef(4)],
    "casts": lambda: [TypeCast(InputRef(1) * Constant(1e10, abi.FLOAT64), abi.INT32), TypeCast(InputRef(2), abi.FLOAT64) / Constant(4.0, abi.FLOAT64),
                      TypeCast(InputRef(0) > Constant(0, abi.INT64), abi.INT64)],
    "constants": lambda: [Constant(5, abi.INT64), Constant(None, abi.FLOAT64), InputRef(0), Constant(True, abi.BOOLEAN)],
    "six_programs": lambda: [InputRef(0) + Constant(k, abi.INT64) for k in range(6)],
}


@pytest.mark.parametrize("proj", sorted(PROJECTIONS))
@pytest.mark.parametrize("depth", [1, 5, 40])
def test_project_push_async_yields_the_batches_of_push(hip, oracle, proj, depth):
    """Round 6: sqlrs_project_push_async — bare column references copied by the host into the slot, expressions over the
    fixed-width columns evaluated per row inside ONE launch (validity and boolean results as one ballot word per wave) —
    is the synchronous ProjectExecutor's stream, and the oracle's, batch for batch (project.rs:15-27)."""
    from sqlrs_amd.executor import ProjectExecutor
    rng = np.random.default_rng(77 + depth)
    bs = bool_batches(rng, [1024] * 12 + [0, 1, 63, 64, 65, 1000, 4096, 4097, 9000] + [1024] * 30)
    exprs = PROJECTIONS[proj]()
    names = [f"o{i}" for i in range(len(exprs))]
    before = fast_batches(hip)
    got = list(ProjectExecutor(hip, exprs, bs, output_names=names, depth=depth).execute())
    took = fast_batches(hip) - before
    eligible = sum(1 for b in bs if b.num_rows <= 4096)
    if depth < 32:
        assert took == eligible, (took, eligible)
    else:
        assert took > 0
    same_batches(got, list(ProjectExecutor(hip, exprs, bs, output_names=names).execute()))
    same_batches(got, list(ProjectExecutor(oracle, exprs, bs, output_names=names).execute()))


def test_project_async_shapes_that_take_the_synchronous_operator(hip, oracle):
    """a Utf8 / Boolean column inside an expression, more than six computed columns: the synchronous operator
    inside push_async, same batches; a division by zero is the evaluator's error at the wait"""
    from sqlrs_amd.executor import ProjectExecutor
    rng = np.random.default_rng(9)
    bs = bool_batches(rng, [1024, 10, 0, 2000])
    for exprs in ([InputRef(4) & InputRef(4)], [InputRef(3).eq(InputRef(3))], [InputRef(0) + Constant(k, abi.INT64) for k in range(7)]):
        names = [f"o{i}" for i in range(len(exprs))]
        before = fast_batches(hip)
        got = list(ProjectExecutor(hip, exprs, bs, output_names=names, depth=3).execute())
        assert fast_batches(hip) == before
        same_batches(got, list(ProjectExecutor(oracle, exprs, bs, output_names=names).execute()))
    z = pa.RecordBatch.from_arrays([pa.array(np.arange(-3, 5)), pa.array(np.arange(8.0))], names=["a", "b"])
    with pytest.raises(abi.ExecutorError, match="Divide by zero"):
        list(ProjectExecutor(hip, [Constant(10, abi.INT64) / InputRef(0)], [z], output_names=["q"], depth=2).execute())
    with pytest.raises(abi.ExecutorError, match="Divide by zero"):
        list(ProjectExecutor(oracle, [Constant(10, abi.INT64) / InputRef(0)], [z], output_names=["q"]).execute())


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_async_project(hip, oracle, seed):
    """random projections (column references of every type, random expression trees over the fixed-width columns) over random
    schemas and batch sizes"""
    from sqlrs_amd.executor import ProjectExecutor
    rng = np.random.default_rng(4000 + seed)
    bs = bool_batches(rng, [int(x) for x in rng.choice([0, 1, 7, 64, 100, 1000, 1024, 2047, 4096], size=10)])
    kinds = {0: "i64", 1: "f64", 2: "i32"}
    DT = {"i64": abi.INT64, "f64": abi.FLOAT64, "i32": abi.INT32}

    def rand_num(kind, depth):
        cols_k = [i for i, k in kinds.items() if k == kind]
        r = rng.random()
        if depth == 0 or r < 0.3:
            if rng.random() < 0.7:
                return InputRef(int(rng.choice(cols_k)))
            return Constant(None if rng.random() < 0.1 else (float(np.round(rng.random() * 4 - 2, 2)) if kind == "f64" else int(rng.integers(-5, 6))), DT[kind])
        if r < 0.45:
            other = str(rng.choice([k for k in DT if k != kind]))
            return TypeCast(rand_num(other, depth - 1), DT[kind])
        a, b = rand_num(kind, depth - 1), rand_num(kind, depth - 1)
        return {0: lambda: a + b, 1: lambda: a - b, 2: lambda: a * b}[int(rng.integers(0, 3))]()

    def rand_bool(depth):
        if depth == 0 or rng.random() < 0.5:
            kind = str(rng.choice(list(DT)))
            a, b = rand_num(kind, 2), rand_num(kind, 2)
            return {0: lambda: a > b, 1: lambda: a <= b, 2: lambda: a.eq(b), 3: lambda: a.ne(b)}[int(rng.integers(0, 4))]()
        a, b = rand_bool(depth - 1), rand_bool(depth - 1)
        return (a & b) if rng.random() < 0.5 else (a | b)
    exprs = []
    for _ in range(int(rng.integers(1, 9))):
        r = rng.random()
        e = InputRef(int(rng.integers(0, 5))) if r < 0.4 else (rand_bool(2) if r < 0.6 else rand_num(str(rng.choice(list(DT))), 3))
        if len(e.nodes()) <= 24 and (len(e.nodes()) == 1 or sum(1 for x in exprs if len(x.nodes()) > 1) < 6):
            exprs.append(e)
    if not exprs:
        exprs = [InputRef(0)]
    names = [f"o{i}" for i in range(len(exprs))]
    before = fast_batches(hip)
    got = list(ProjectExecutor(hip, exprs, bs, output_names=names, depth=int(rng.integers(1, 7))).execute())
    assert fast_batches(hip) - before == len(bs)
    same_batches(got, list(ProjectExecutor(oracle, exprs, bs, output_names=names).execute()))


@pytest.mark.parametrize("depth", [1, 2, 4, 7])
def test_async_operators_chained_on_one_ctx(hip, oracle, depth):
    """Filter -> Project -> HashJoin probe, every operator through its push_async on ONE ctx: a push that flushes another operator's
    pending group must not mark its own ticket as launched (the C++ mirror's replay found the last ticket of such a chain
    waiting for a launch that never came)"""
    from sqlrs_amd.executor import ProjectExecutor
    rng = np.random.default_rng(300 + depth)
    bs = batches(rng, 0, [2, 2, 2, 1024, 1, 0, 3000, 7, 1024, 64, 2, 2, 2], 0.0)
    pred = InputRef(0) > Constant(-20, abi.INT64)
    exprs = [InputRef(0), InputRef(1) * Constant(2.0, abi.FLOAT64), InputRef(2)]
    names = ["a", "b2", "c"]
    lb = pa.RecordBatch.from_arrays([pa.array(np.arange(-100, 100, dtype=np.int64)), pa.array(np.arange(200, dtype=np.int64) * 10)], names=["k", "p"])
    cond = JoinCondition([(InputRef(0), InputRef(0))])

    def plan(be, d):
        f = FilterExecutor(be, pred, bs, depth=d).execute()
        pr = ProjectExecutor(be, exprs, f, output_names=names, depth=d).execute()
        rb0 = pa.RecordBatch.from_arrays([pa.array([], type=pa.int64()), pa.array([], type=pa.float64()), pa.array([], type=pa.int32())], names=names)
        return list(HashJoinExecutor(be, [lb], pr, "inner", cond, join_schema(lb, rb0), 2, depth=d).execute())
    before = fast_batches(hip)
    got = plan(hip, depth)
    assert fast_batches(hip) - before == 3 * len(bs)
    same_batches(got, plan(oracle, 0))
