"""CSV ingest (storage/csv.rs:92-241) and config C1 of BASELINE.json end to end on the HIP backend:
employee.csv -> scan -> HashAgg(state; count(state), sum(salary)) -> record_batch_to_string.

The reference reads CSV through arrow-csv 28 (absent here); pyarrow.csv (Arrow C++) is the
differential comparator for the reader, the reference's own golden (aggregation.slt:30-34) pins C1."""
import os

import pyarrow as pa
import pyarrow.csv as pacsv
import pytest

from sqlrs_amd import abi
from sqlrs_amd.executor import CsvScan, FilterExecutor, HashAggExecutor, LimitExecutor, OrderExecutor, ProjectExecutor
from sqlrs_amd.expr import AggFunc, Constant, InputRef, OrderBy

pytestmark = pytest.mark.gpu
CSV_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "csv")


def read_all(scan):
    batches = list(scan.execute())
    return batches, (pa.Table.from_batches(batches) if batches else None)


@pytest.mark.parametrize("name", ["employee", "department", "state", "t1", "t2"])
def test_csv_reader_matches_pyarrow(hip, name):
    path = os.path.join(CSV_DIR, name + ".csv")
    scan = CsvScan(hip, path)
    batches, got = read_all(scan)
    exp = pacsv.read_csv(path, convert_options=pacsv.ConvertOptions(strings_can_be_null=False))
    assert scan.names == [c.lower() for c in exp.column_names]
    assert got.num_rows == exp.num_rows
    for i, col in enumerate(exp.columns):
        assert got.column(i).to_pylist() == col.to_pylist(), exp.column_names[i]


def test_csv_dialect_batches_bounds_projection(hip, tmp_path):
    p = tmp_path / "x.csv"
    rows = [f'{i},{i * 0.5},{"true" if i % 3 else "false"},"s,{i}","he said ""hi"""' for i in range(2500)]
    rows[7] = '7,,,,'          # missing values: NULL for typed columns, empty string for text
    p.write_text("A,B,C,D,E\r\n" + "\r\n".join(rows) + "\r\n")
    scan = CsvScan(hip, str(p), batch_size=1024)
    batches, got = read_all(scan)
    assert [b.num_rows for b in batches] == [1024, 1024, 452]          # csv.rs:105 batch_size
    assert scan.dtypes == [abi.INT64, abi.FLOAT64, abi.BOOLEAN, abi.UTF8, abi.UTF8]
    assert got.column(0).to_pylist() == list(range(2500))
    assert got.column(1).to_pylist()[:9] == [0.0, 0.5, 1.0, 1.5, 2.0, 2.5, 3.0, None, 4.0]
    assert got.column(2).to_pylist()[6:9] == [False, None, True]
    assert got.column(3).to_pylist()[6:9] == ["s,6", "", "s,8"] and got.column(4)[0].as_py() == 'he said "hi"'
    # bounds = (offset, limit) over the records, projection = column indices (csv.rs:207-224)
    _, part = read_all(CsvScan(hip, str(p), bounds=(1000, 30), projection=[3, 0]))
    assert part.column(1).to_pylist() == list(range(1000, 1030)) and part.column(0)[0].as_py() == "s,1000"
    # a file WITHOUT a header yields limit + 1 records: csv.rs:216-223 hands arrow-csv the line bounds
    # (offset, offset + limit + 1) and arrow-csv 28 starts its line counter one later only when there is a header
    q = tmp_path / "nohdr.csv"
    q.write_text("\n".join(f"{i},{i * 2}" for i in range(100)) + "\n")
    _, nh = read_all(CsvScan(hip, str(q), has_header=False, bounds=(10, 5)))
    assert nh.column(0).to_pylist() == list(range(10, 16))
    # straight into HBM
    dev = list(CsvScan(hip, str(p), out_mem=abi.MEM_DEVICE).execute())
    assert [d.num_rows for d in dev] == [1024, 1024, 452]
    assert hip.batch_to_string(dev[0]).splitlines()[7] == "7 NULL NULL (empty) (empty)"


def test_csv_malformed_input_follows_arrow_csv(hip, tmp_path):
    """a record with another number of fields than the header is an error (the csv crate's UnequalLengths, surfaced
    by arrow-csv as an ArrowError), not a row padded with NULLs; `1e+5` is text (arrow-csv 28's DECIMAL_RE allows
    only `[eE]-?\\d+`), `1e-5` and `2E5` are floats"""
    p = tmp_path / "ragged.csv"
    p.write_text("a,b,c\n1,2,3\n4,5\n7,8,9\n")
    with pytest.raises(Exception, match="found record with 2 fields"):
        read_all(CsvScan(hip, str(p)))
    q = tmp_path / "exp.csv"
    q.write_text("x,y\n1e-5,1e+5\n2E5,3\n")
    scan = CsvScan(hip, str(q))
    _, got = read_all(scan)
    assert scan.dtypes == [abi.FLOAT64, abi.UTF8]
    assert got.column(0).to_pylist() == [1e-5, 2e5] and got.column(1).to_pylist() == ["1e+5", "3"]


def test_c1_employee_group_by_state_end_to_end(hip, oracle):
    """BASELINE.json config C1 with the reference's golden (aggregation.slt:30-34 restricted to the three
    columns): CSV -> HBM -> HashAgg -> text, nothing but the file and the final string on the host"""
    path = os.path.join(CSV_DIR, "employee.csv")
    scan = CsvScan(hip, path, out_mem=abi.MEM_DEVICE)
    # employee: id, first_name, last_name, state(3), job_title, salary(5), department_id
    agg = HashAggExecutor(hip, [AggFunc("count", InputRef(3), abi.INT64), AggFunc("sum", InputRef(5), abi.INT64)],
                          [InputRef(3)], scan.execute(), out_mem=abi.MEM_DEVICE)
    (out,) = list(agg.execute())
    assert hip.batch_to_string(out) == "CA 1 12000\nCO 2 21500\n(empty) 1 NULL\n"
    # the same plan on the oracle, fed by the same scan through host batches
    host = list(CsvScan(hip, path).execute())
    (exp,) = list(HashAggExecutor(oracle, [AggFunc("count", InputRef(3), abi.INT64), AggFunc("sum", InputRef(5), abi.INT64)],
                                  [InputRef(3)], host).execute())
    assert oracle.batch_to_string(exp) == "CA 1 12000\nCO 2 21500\n(empty) 1 NULL\n"


def test_device_resident_plan_scan_filter_project_order_limit(hip, oracle):
    """select first_name, salary + 1 from employee where id > 1 order by salary desc limit 2 offset 1 — every
    operator device resident, compared with the oracle in text form"""
    path = os.path.join(CSV_DIR, "employee.csv")

    def plan(be, mem, child):
        f = FilterExecutor(be, InputRef(0) > Constant(1, abi.INT64), child, out_mem=mem)
        p = ProjectExecutor(be, [InputRef(1), InputRef(5) + Constant(1, abi.INT64)], f.execute(), out_mem=mem)
        o = OrderExecutor(be, [OrderBy(InputRef(1), asc=False)], p.execute(), out_mem=mem)
        return "".join(be.batch_to_string(b) for b in LimitExecutor(be, 2, 1, o.execute(), out_mem=mem).execute())
    got = plan(hip, abi.MEM_DEVICE, CsvScan(hip, path, out_mem=abi.MEM_DEVICE).execute())
    exp = plan(oracle, abi.MEM_HOST, list(CsvScan(hip, path).execute()))
    assert got == exp and got.count("\n") == 2
