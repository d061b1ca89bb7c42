"""Pins the CPU oracle against every golden table the reference's own tests hold for the
hot path.  CPU only."""
import pytest

from golden_runner import Runner, load, render_rows

FX = load()


@pytest.mark.parametrize("case", FX["cases"], ids=[c["name"] for c in FX["cases"]])
def test_oracle_matches_reference_golden(oracle, case):
    got = Runner(oracle, FX).rows(case["plan"])
    assert got == case["expected"], f"{case['name']} ({case['source']})"
    if "expected_types" in case:
        assert Runner(oracle, FX).types(case["plan"]) == case["expected_types"]


@pytest.mark.parametrize("case", FX["cases"], ids=[c["name"] for c in FX["cases"]])
def test_oracle_text_form_matches_reference_golden(oracle, case):
    """the same goldens compared in the reference's own text form (record_batch_to_string,
    util/mod.rs:53-80: NULL / (empty) / Display), as its sqllogictest harness does"""
    assert Runner(oracle, FX).text(case["plan"]) == render_rows(case["expected"])


LIMIT_CASES = [  # limit.rs:93-98 #[test_case(inputs, offset, limit, outputs)]
    ([(0, 6)], 1, 4, [(1, 5)]),
    ([(0, 6)], 0, 10, [(0, 6)]),
    ([(0, 6)], 10, 0, []),
    ([(0, 2), (2, 4), (4, 6)], 1, 4, [(1, 2), (2, 4), (4, 5)]),
    ([(0, 2), (2, 4), (4, 6)], 1, 2, [(1, 2), (2, 3)]),
    ([(0, 2), (2, 4), (4, 6)], 3, 0, []),
]


def limit_case(backend, inputs, offset, limit):
    import pyarrow as pa
    from sqlrs_amd.executor import LimitExecutor
    chunks = [pa.RecordBatch.from_arrays([pa.array(list(range(a, b)), type=pa.int32())], names=["a"]) for a, b in inputs]
    pulled = []

    def child():
        for c in chunks:
            pulled.append(1)
            yield c
    out = list(LimitExecutor(backend, limit, offset, child()).execute())
    return [b.column(0).to_pylist() for b in out], len(pulled)


@pytest.mark.parametrize("inputs,offset,limit,outputs", LIMIT_CASES)
def test_oracle_limit_unit_cases(oracle, inputs, offset, limit, outputs):
    got, _ = limit_case(oracle, inputs, offset, limit)
    assert got == [list(range(a, b)) for a, b in outputs]  # one output batch per contributing input batch


def test_oracle_limit_stops_pulling_the_child(oracle):
    got, pulled = limit_case(oracle, [(0, 2), (2, 4), (4, 6)], 1, 2)
    assert pulled == 2  # limit.rs:76-78 breaks after the batch that reaches offset + limit
    _, pulled0 = limit_case(oracle, [(0, 2), (2, 4)], 0, 0)
    assert pulled0 == 0  # limit.rs:29-31 returns before polling the child


def test_oracle_simple_agg_edge_cases(oracle):
    import pyarrow as pa
    from sqlrs_amd import abi
    from sqlrs_amd.executor import SimpleAggExecutor
    from sqlrs_amd.expr import AggFunc, InputRef
    aggs = [AggFunc("count", InputRef(0), abi.INT64), AggFunc("sum", InputRef(0), abi.INT64),
            AggFunc("max", InputRef(0), abi.INT64)]
    empty = pa.RecordBatch.from_arrays([pa.array([], type=pa.int64())], names=["a"])
    (out,) = list(SimpleAggExecutor(oracle, aggs, [empty]).execute())
    assert [c.to_pylist() for c in out.columns] == [[0], [None], [None]]
    with pytest.raises(abi.ExecutorError):
        list(SimpleAggExecutor(oracle, aggs, []).execute())  # simple_agg.rs:63 unwraps a None


def symmetric_combine_case(backend):
    """create_hashes folds the key columns with combine_hashes (hash_utils.rs:13-16): for two columns of the
    same type the fold is symmetric — (a, b) and (b, a) get the same row hash — and the join / group-by match
    on the hash alone (hash_join.rs:222-224, hash_agg.rs:87).  So (1, 2) joins (2, 1) and both fall into one
    group.  Pinned here so that nobody "fixes" it on one side only."""
    import pyarrow as pa
    from sqlrs_amd import abi
    from sqlrs_amd.executor import HashAggExecutor, HashJoinExecutor
    from sqlrs_amd.expr import AggFunc, InputRef, JoinCondition
    lb = pa.RecordBatch.from_arrays([pa.array([1, 5], type=pa.int64()), pa.array([2, 6], type=pa.int64())], names=["a", "b"])
    rb = pa.RecordBatch.from_arrays([pa.array([2, 5, 7], type=pa.int64()), pa.array([1, 6, 7], type=pa.int64())], names=["a", "b"])
    sch = pa.schema([("l.a", pa.int64()), ("l.b", pa.int64()), ("r.a", pa.int64()), ("r.b", pa.int64())])
    cond = JoinCondition([(InputRef(0), InputRef(0)), (InputRef(1), InputRef(1))])
    joined = [tuple(c.to_pylist() for c in b.columns) for b in HashJoinExecutor(backend, [lb], [rb], "inner", cond, sch, 2).execute()]
    both = pa.RecordBatch.from_arrays([pa.array([1, 2, 9], type=pa.int64()), pa.array([2, 1, 9], type=pa.int64()),
                                       pa.array([10, 20, 30], type=pa.int64())], names=["a", "b", "v"])
    (g,) = list(HashAggExecutor(backend, [AggFunc("sum", InputRef(2), abi.INT64)], [InputRef(0), InputRef(1)], [both]).execute())
    return joined, [c.to_pylist() for c in g.columns]


def test_two_column_keys_inherit_symmetric_combine(oracle):
    joined, groups = symmetric_combine_case(oracle)
    assert joined == [([1, 5], [2, 6], [2, 5], [1, 6])]   # (1,2) x (2,1) match by hash; (5,6) x (5,6) by value
    assert groups == [[1, 9], [2, 9], [30, 30]]           # (1,2) and (2,1) are ONE group, keyed by its first row
