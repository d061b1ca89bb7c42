"""The reference's golden tables replayed on the HIP path through the C ABI (MI355X)."""
import pytest

from golden_runner import Runner, load, render_rows

FX = load()
GPU_CASES = [c for c in FX["cases"] if c["gpu"]]


@pytest.mark.gpu
@pytest.mark.parametrize("case", GPU_CASES, ids=[c["name"] for c in GPU_CASES])
def test_hip_matches_reference_golden(hip, oracle, case):
    got = Runner(hip, FX).rows(case["plan"])
    assert got == case["expected"], f"{case['name']} ({case['source']})"
    assert got == Runner(oracle, FX).rows(case["plan"])
    if "expected_types" in case:
        assert Runner(hip, FX).types(case["plan"]) == case["expected_types"]
    # and in the reference's own text form (record_batch_to_string, util/mod.rs:53-80)
    assert Runner(hip, FX).text(case["plan"]) == render_rows(case["expected"])


@pytest.mark.gpu
def test_cpp_host_mirror_replays_reference_unit_tests():
    """host/test_reference_executor.cpp: the reference's operator unit tests (same child streams,
    same operator structs, same expected pretty-printed tables) through the C++ host mirror."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "host", "test_reference_executor")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "host"), "-s"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PASSED" in r.stdout


@pytest.mark.gpu
def test_hip_limit_unit_cases(hip, oracle):
    """limit.rs:93-98 test cases + the arithmetic across batches on device-resident input"""
    from test_oracle_golden import LIMIT_CASES, limit_case
    for inputs, offset, limit, outputs in LIMIT_CASES:
        got, pulled = limit_case(hip, inputs, offset, limit)
        assert got == [list(range(a, b)) for a, b in outputs]
        assert (got, pulled) == limit_case(oracle, inputs, offset, limit)
    # None = no LIMIT / no OFFSET clause; NULLs and strings through the device slice
    import pyarrow as pa
    from sqlrs_amd import abi
    from sqlrs_amd.executor import LimitExecutor
    b = pa.RecordBatch.from_arrays([pa.array([1, None, 3, 4, None, 6, 7], type=pa.int64()),
                                    pa.array(["a", "", None, "dd", "e", "ff", None]),
                                    pa.array([True, None, False, True, True, None, False])], names=["x", "s", "t"])
    for limit, offset in [(None, 2), (3, None), (2, 3), (100, 1), (None, None)]:
        for be_out in (abi.MEM_HOST,):
            got = [[c.to_pylist() for c in r.columns] for r in LimitExecutor(hip, limit, offset, [hip.to_device(b), b]).execute()]
            exp = [[c.to_pylist() for c in r.columns] for r in LimitExecutor(oracle, limit, offset, [b, b]).execute()]
            assert got == exp, (limit, offset)


@pytest.mark.gpu
def test_hip_project_and_simple_agg(hip, oracle):
    import numpy as np
    import pyarrow as pa
    from sqlrs_amd import abi
    from sqlrs_amd.executor import ProjectExecutor, SimpleAggExecutor
    from sqlrs_amd.expr import AggFunc, Constant, InputRef, TypeCast
    rng = np.random.default_rng(3)
    n = 70_000
    b = pa.RecordBatch.from_arrays([pa.array(rng.integers(-50, 50, n), mask=rng.random(n) < 0.1),
                                    pa.array(rng.random(n), mask=rng.random(n) < 0.1),
                                    pa.array([None if i % 7 == 0 else f"s{i % 13}" for i in range(n)])], names=["a", "b", "c"])
    exprs = [InputRef(2), InputRef(0) + Constant(1, abi.INT64), TypeCast(InputRef(0), abi.FLOAT64) * InputRef(1),
             InputRef(1) > Constant(0.5, abi.FLOAT64), Constant(7, abi.INT64)]
    for child in ([b], [b.slice(0, 1000), b.slice(1000)], [hip.to_device(b)]):
        got = [[c.to_pylist() for c in r.columns] for r in ProjectExecutor(hip, exprs, child).execute()]
        exp = [[c.to_pylist() for c in r.columns] for r in
               ProjectExecutor(oracle, exprs, [c if isinstance(c, pa.RecordBatch) else b for c in child]).execute()]
        assert got == exp
    aggs = [AggFunc("count", InputRef(0), abi.INT64), AggFunc("sum", InputRef(0), abi.INT64), AggFunc("sum", InputRef(1), abi.FLOAT64),
            AggFunc("min", InputRef(1), abi.FLOAT64), AggFunc("max", InputRef(2), abi.UTF8), AggFunc("count", InputRef(0), abi.INT64, distinct=True)]
    for child in ([b], [b.slice(0, 999), b.slice(999, 0), b.slice(999)]):
        (g,) = list(SimpleAggExecutor(hip, aggs, child).execute())
        (e,) = list(SimpleAggExecutor(oracle, aggs, child).execute())
        assert g.num_rows == e.num_rows == 1
        gl, el = [c[0].as_py() for c in g.columns], [c[0].as_py() for c in e.columns]
        assert gl[:2] == el[:2] and gl[3:] == el[3:] and abs(gl[2] - el[2]) <= 1e-9 * abs(el[2])
    empty = b.slice(0, 0)
    (g,) = list(SimpleAggExecutor(hip, aggs, [empty]).execute())
    (e,) = list(SimpleAggExecutor(oracle, aggs, [empty]).execute())
    assert [c.to_pylist() for c in g.columns] == [c.to_pylist() for c in e.columns] == [[0], [None], [None], [None], [None], [0]]
    with pytest.raises(abi.ExecutorError):
        list(SimpleAggExecutor(hip, aggs, []).execute())


@pytest.mark.gpu
def test_hip_text_form_matches_oracle(hip, oracle):
    """record_batch_to_string on values the goldens do not hold: floats (Rust Display), bools, NULL, empty strings"""
    import pyarrow as pa
    b = pa.RecordBatch.from_arrays([
        pa.array([12000.0, 0.1, 1e21, 1e-7, -0.0, 1100.2, None, float("inf"), float("nan"), 2.5e-320]),
        pa.array([1, -2, None, 4, 5, 6, 7, 8, 9, 1 << 62], type=pa.int64()),
        pa.array(["x", "", None, "a b", "\u00e9", "f", "g", "h", "i", "j"]),
        pa.array([True, False, None, True, True, False, False, True, None, True]),
        pa.array([1, 2, 3, None, 5, 6, 7, 8, 9, -10], type=pa.int32())], names=list("abcde"))
    exp = oracle.batch_to_string(b)
    assert exp.splitlines()[0] == "12000 1 x true 1" and exp.splitlines()[1] == "0.1 -2 (empty) false 2"
    assert exp.splitlines()[2].startswith("1000000000000000000000 NULL NULL NULL 3")
    assert hip.batch_to_string(b) == exp
    assert hip.batch_to_string(hip.to_device(b)) == exp


@pytest.mark.gpu
def test_two_column_keys_inherit_symmetric_combine(hip, oracle):
    from test_oracle_golden import symmetric_combine_case
    assert symmetric_combine_case(hip) == symmetric_combine_case(oracle)


@pytest.mark.gpu
@pytest.mark.parametrize("per_call", [1, 7, 1000, (1 << 31) - 1])
def test_cross_join_left_rows_in_ranges(hip, oracle, per_call):
    """CrossJoinExecutor (cross_join.rs:8-58): one output batch per (right batch, left row).  A library batch holds < 2^31
    rows, so a large product takes the left rows in ranges (sqlrs_cross_join_probe_push_range) — forced here with a small
    per-call row budget; an empty right batch still yields one empty batch per left row.  Against the oracle."""
    import numpy as np
    import pyarrow as pa
    from sqlrs_amd.executor import CrossJoinExecutor
    rng = np.random.default_rng(per_call % 97)
    lbs = [pa.RecordBatch.from_arrays([pa.array(rng.integers(0, 99, n, dtype=np.int64)), pa.array(rng.random(n))], names=["a", "b"]) for n in (5, 0, 12)]
    rbs = [pa.RecordBatch.from_arrays([pa.array(rng.integers(0, 9, n, dtype=np.int64), mask=rng.random(n) < 0.2)], names=["c"]) for n in (3, 0, 40)]
    sch = pa.schema([("l.a", pa.int64()), ("l.b", pa.float64()), ("r.c", pa.int64())])
    exp = list(CrossJoinExecutor(oracle, lbs, rbs, sch).execute())
    got = list(CrossJoinExecutor(hip, lbs, rbs, sch, max_rows_per_call=per_call).execute())
    assert [b.num_rows for b in got] == [b.num_rows for b in exp] and len(got) == 17 * 3
    for g, e in zip(got, exp):
        assert g.equals(e)
