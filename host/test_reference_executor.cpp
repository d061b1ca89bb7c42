// test_reference_executor.cpp — the reference's own operator unit tests, replayed against the
// HIP backend through the C++ host mirror (host/sqlrs_executor.hpp).
//
// Each test builds the same child streams (futures::stream::iter -> stream_iter), the same
// operator struct and asserts on the same pretty-printed table as the reference test it is
// named after:
//   src/executor/join/hash_join.rs:423-750   test_{inner,left,right,full}_join[_filter]_results
//   src/executor/aggregate/hash_agg.rs:182-222 test_hash_agg_with_multiple_chunks
//   src/executor/mod.rs:309-345                test_executor_hash_agg_works (operator level)
//   src/executor/mod.rs:264-285,368-395        filter / order, operator level
//   src/executor/limit.rs:97-117               limit (six cases); project / simple_agg at operator level
//   tests/slt/subquery.slt:50-57               scalar subquery = CrossJoinExecutor (cross_join.rs:8-58)
// The expected tables are the reference's golden vectors (data), not its code.
//
// Build:  g++ -std=c++17 -Iinclude host/test_reference_executor.cpp -Lsqlrs_amd/csrc -lsqlrs_hip
// Run on a machine with an MI355X (pytest -m gpu does both).
#include <cstdio>
#include <iostream>

#include "sqlrs_executor.hpp"

using namespace sqlrs;

static int failures = 0;
static HipCtxRef ctx;

static void expect_table(const char *name, const std::vector<RecordBatch> &output,
                         const std::vector<std::string> &expected) {
  std::string table = pretty_format_batches(output);
  std::string exp;
  for (size_t i = 0; i < expected.size(); i++) exp += expected[i] + (i + 1 < expected.size() ? "\n" : "");
  if (table == exp) {
    std::printf("ok   %s\n", name);
  } else {
    failures++;
    std::printf("FAIL %s\nActual result:\n%s\nExpected:\n%s\n", name, table.c_str(), exp.c_str());
  }
}

// build_table_i32 (hash_join.rs:341-361)
static RecordBatch build_table_i32(std::pair<const char *, std::vector<int32_t>> a,
                                   std::pair<const char *, std::vector<int32_t>> b,
                                   std::pair<const char *, std::vector<int32_t>> c) {
  auto schema = std::make_shared<Schema>(Schema{{a.first, DataType::Int32, false},
                                                {b.first, DataType::Int32, false},
                                                {c.first, DataType::Int32, false}});
  return RecordBatch::try_new(schema, {Int32Array(a.second), Int32Array(b.second), Int32Array(c.second)});
}

// build_table_schema (hash_join.rs:363-382)
static std::vector<ColumnCatalog> build_table_schema(const std::string &table_id, const RecordBatch &batch, bool nullable) {
  std::vector<ColumnCatalog> out;
  for (auto &f : *batch.schema) out.push_back(ColumnCatalog{table_id, f.name, nullable, ColumnDesc{f.name, f.data_type}});
  return out;
}

struct TestChild {
  BoxedExecutor left, right;
  std::vector<ColumnCatalog> schema;
};

static void force_nullable(JoinType jt, bool &l, bool &r) { // hash_join.rs:385-391
  l = jt == JoinType::Right || jt == JoinType::Full;
  r = jt == JoinType::Left || jt == JoinType::Full;
}

// build_test_child (hash_join.rs:384-421)
static TestChild build_test_child(JoinType jt) {
  bool ln, rn;
  force_nullable(jt, ln, rn);
  RecordBatch lb = build_table_i32({"a1", {0, 1, 2, 3, 4}}, {"b1", {0, 4, 5, 5, 8}}, {"c1", {10, 7, 8, 9, 10}});
  RecordBatch rb = build_table_i32({"a2", {10, 20, 30}}, {"b1", {4, 5, 6}}, {"c2", {70, 80, 90}});
  TestChild t;
  t.schema = build_table_schema("l", lb, ln);
  auto rs = build_table_schema("r", rb, rn);
  t.schema.insert(t.schema.end(), rs.begin(), rs.end());
  t.left = stream_iter({lb});
  t.right = stream_iter({rb});
  return t;
}

// build_test_join_filter_child (hash_join.rs:552-592)
static TestChild build_test_join_filter_child(JoinType jt) {
  bool ln, rn;
  force_nullable(jt, ln, rn);
  RecordBatch lb = build_table_i32({"a", {0, 1, 2, 2}}, {"b", {4, 5, 7, 8}}, {"c", {7, 8, 9, 1}});
  RecordBatch rb = build_table_i32({"a", {10, 20, 30, 40}}, {"b", {2, 2, 3, 4}}, {"c", {7, 5, 6, 6}});
  TestChild t;
  t.schema = build_table_schema("l", lb, ln);
  auto rs = build_table_schema("r", rb, rn);
  t.schema.insert(t.schema.end(), rs.begin(), rs.end());
  t.left = stream_iter({lb});
  t.right = stream_iter({rb});
  return t;
}

static void join_test(const char *name, JoinType jt, bool with_filter, const std::vector<std::string> &expected) {
  TestChild t = with_filter ? build_test_join_filter_child(jt) : build_test_child(jt);
  HashJoinExecutor ex;
  ex.ctx = ctx;
  ex.left_child = std::move(t.left);
  ex.right_child = std::move(t.right);
  ex.join_type = jt;
  if (with_filter) { // on t1.a = t2.b and t1.c > t2.c   (hash_join.rs:605-613)
    ex.join_condition.on = {{build_bound_input_ref(0), build_bound_input_ref(1)}};
    ex.join_condition.filter = BoundExpr::binary_op(BinaryOperator::Gt, build_bound_input_ref(2), build_bound_input_ref(5));
  } else {
    ex.join_condition.on = {{build_bound_input_ref(1), build_bound_input_ref(1)}};
  }
  ex.join_output_schema = t.schema;
  ex.num_left_columns = 3;
  expect_table(name, try_collect(ex.execute()), expected);
}

int main() {
  try {
    ctx = std::make_shared<HipCtx>(0);
  } catch (const ExecutorError &e) {
    std::printf("no device: %s\n", e.what());
    return 2;
  }
  // ---- hash_join.rs:423-550
  join_test("test_inner_join_results", JoinType::Inner, false,
            {"+------+------+------+------+------+------+", "| l.a1 | l.b1 | l.c1 | r.a2 | r.b1 | r.c2 |",
             "+------+------+------+------+------+------+", "| 1    | 4    | 7    | 10   | 4    | 70   |",
             "| 2    | 5    | 8    | 20   | 5    | 80   |", "| 3    | 5    | 9    | 20   | 5    | 80   |",
             "+------+------+------+------+------+------+"});
  join_test("test_left_join_results", JoinType::Left, false,
            {"+------+------+------+------+------+------+", "| l.a1 | l.b1 | l.c1 | r.a2 | r.b1 | r.c2 |",
             "+------+------+------+------+------+------+", "| 1    | 4    | 7    | 10   | 4    | 70   |",
             "| 2    | 5    | 8    | 20   | 5    | 80   |", "| 3    | 5    | 9    | 20   | 5    | 80   |",
             "| 0    | 0    | 10   |      |      |      |", "| 4    | 8    | 10   |      |      |      |",
             "+------+------+------+------+------+------+"});
  join_test("test_right_join_results", JoinType::Right, false,
            {"+------+------+------+------+------+------+", "| l.a1 | l.b1 | l.c1 | r.a2 | r.b1 | r.c2 |",
             "+------+------+------+------+------+------+", "| 1    | 4    | 7    | 10   | 4    | 70   |",
             "| 2    | 5    | 8    | 20   | 5    | 80   |", "| 3    | 5    | 9    | 20   | 5    | 80   |",
             "|      |      |      | 30   | 6    | 90   |", "+------+------+------+------+------+------+"});
  join_test("test_full_join_results", JoinType::Full, false,
            {"+------+------+------+------+------+------+", "| l.a1 | l.b1 | l.c1 | r.a2 | r.b1 | r.c2 |",
             "+------+------+------+------+------+------+", "| 1    | 4    | 7    | 10   | 4    | 70   |",
             "| 2    | 5    | 8    | 20   | 5    | 80   |", "| 3    | 5    | 9    | 20   | 5    | 80   |",
             "|      |      |      | 30   | 6    | 90   |", "| 0    | 0    | 10   |      |      |      |",
             "| 4    | 8    | 10   |      |      |      |", "+------+------+------+------+------+------+"});
  // ---- hash_join.rs:594-750
  join_test("test_inner_join_filter_results", JoinType::Inner, true,
            {"+-----+-----+-----+-----+-----+-----+", "| l.a | l.b | l.c | r.a | r.b | r.c |",
             "+-----+-----+-----+-----+-----+-----+", "| 2   | 7   | 9   | 10  | 2   | 7   |",
             "| 2   | 7   | 9   | 20  | 2   | 5   |", "+-----+-----+-----+-----+-----+-----+"});
  join_test("test_left_join_filter_results", JoinType::Left, true,
            {"+-----+-----+-----+-----+-----+-----+", "| l.a | l.b | l.c | r.a | r.b | r.c |",
             "+-----+-----+-----+-----+-----+-----+", "| 2   | 7   | 9   | 10  | 2   | 7   |",
             "| 2   | 7   | 9   | 20  | 2   | 5   |", "| 0   | 4   | 7   |     |     |     |",
             "| 1   | 5   | 8   |     |     |     |", "| 2   | 8   | 1   |     |     |     |",
             "+-----+-----+-----+-----+-----+-----+"});
  join_test("test_right_join_filter_results", JoinType::Right, true,
            {"+-----+-----+-----+-----+-----+-----+", "| l.a | l.b | l.c | r.a | r.b | r.c |",
             "+-----+-----+-----+-----+-----+-----+", "| 2   | 7   | 9   | 10  | 2   | 7   |",
             "| 2   | 7   | 9   | 20  | 2   | 5   |", "|     |     |     | 30  | 3   | 6   |",
             "|     |     |     | 40  | 4   | 6   |", "+-----+-----+-----+-----+-----+-----+"});
  join_test("test_full_join_filter_results", JoinType::Full, true,
            {"+-----+-----+-----+-----+-----+-----+", "| l.a | l.b | l.c | r.a | r.b | r.c |",
             "+-----+-----+-----+-----+-----+-----+", "| 2   | 7   | 9   | 10  | 2   | 7   |",
             "| 2   | 7   | 9   | 20  | 2   | 5   |", "|     |     |     | 30  | 3   | 6   |",
             "|     |     |     | 40  | 4   | 6   |", "| 0   | 4   | 7   |     |     |     |",
             "| 1   | 5   | 8   |     |     |     |", "| 2   | 8   | 1   |     |     |     |",
             "+-----+-----+-----+-----+-----+-----+"});

  // ---- hash_agg.rs:182-222  select a, sum(b) from t group by a  (two identical chunks)
  {
    auto schema = std::make_shared<Schema>(Schema{{"a", DataType::Int64, false}, {"b", DataType::Int64, false}});
    RecordBatch chunk = RecordBatch::try_new(schema, {Int64Array({1, 1, 2}), Int64Array({1, 1, 3})});
    HashAggExecutor ex;
    ex.ctx = ctx;
    ex.agg_funcs = {BoundAggFunc{AggFunc::Sum, {build_bound_input_ref(1)}, DataType::Int64, false}};
    ex.group_by = {build_bound_input_ref(0)};
    ex.child = stream_iter({chunk, chunk});
    ex.output_names = {"a", "Sum(b)"};
    expect_table("test_hash_agg_with_multiple_chunks", try_collect(ex.execute()),
                 {"+---+--------+", "| a | Sum(b) |", "+---+--------+", "| 1 | 4      |", "| 2 | 6      |", "+---+--------+"});
  }

  // ---- tests/slt/subquery.slt:50-57  select a, (select max(b) from t1) max_b from t1 — the cross join the binder makes of
  //      the scalar subquery (binder/table/subquery.rs:120-167), executed by CrossJoinExecutor (cross_join.rs:8-58):
  //      one output batch per (right batch, left row)
  {
    auto t1s = std::make_shared<Schema>(Schema{{"a", DataType::Int64, false}, {"b", DataType::Int64, false}, {"c", DataType::Int64, false}});
    RecordBatch t1 = RecordBatch::try_new(t1s, {Int64Array({0, 1, 2, 2}), Int64Array({4, 5, 7, 8}), Int64Array({7, 8, 9, 1})});
    SimpleAggExecutor sub;
    sub.ctx = ctx;
    sub.agg_funcs = {BoundAggFunc{AggFunc::Max, {build_bound_input_ref(1)}, DataType::Int64, false}};
    sub.child = stream_iter({t1});
    CrossJoinExecutor cj;
    cj.ctx = ctx;
    cj.left_child = stream_iter({t1.slice(0, 2), t1.slice(2, 2)}); // two left batches: concatenated first (cross_join.rs:36)
    cj.right_child = sub.execute();
    std::vector<RecordBatch> batches = try_collect(cj.execute());
    if (batches.size() != 4) {
      failures++;
      std::printf("FAIL cross_join: %zu output batches, expected one per left row (4)\n", batches.size());
    }
    ProjectExecutor pr;
    pr.ctx = ctx;
    pr.exprs = {build_bound_input_ref(0), build_bound_input_ref(3)};
    pr.child = stream_iter(batches);
    pr.output_names = {"a", "max_b"};
    expect_table("slt_scalar_subquery_alias (cross join)", try_collect(pr.execute()),
                 {"+---+-------+", "| a | max_b |", "+---+-------+", "| 0 | 8     |", "| 1 | 8     |", "| 2 | 8     |", "| 2 | 8     |", "+---+-------+"});
  }

  // ---- executor/mod.rs:221-241 table, operator-level plans of :309-345, :264-285, :368-395
  auto emp_schema = std::make_shared<Schema>(Schema{{"id", DataType::Int64, false}, {"first_name", DataType::Utf8, false},
                                                    {"last_name", DataType::Utf8, false}, {"salary", DataType::Int64, false}});
  RecordBatch employee = RecordBatch::try_new(
      emp_schema, {Int64Array({1, 2, 3, 4}), StringArray({"Bill", "Gregg", "John", "Von"}),
                   StringArray({"Hopkins", "Langford", "Travis", "Mill"}), Int64Array({100, 100, 200, 400})});
  { // select salary, count(id), sum(id), max(id), min(id) from employee group by salary
    HashAggExecutor ex;
    ex.ctx = ctx;
    ex.agg_funcs = {BoundAggFunc{AggFunc::Count, {build_bound_input_ref(0)}, DataType::Int64, false},
                    BoundAggFunc{AggFunc::Sum, {build_bound_input_ref(0)}, DataType::Int64, false},
                    BoundAggFunc{AggFunc::Max, {build_bound_input_ref(0)}, DataType::Int64, false},
                    BoundAggFunc{AggFunc::Min, {build_bound_input_ref(0)}, DataType::Int64, false}};
    ex.group_by = {build_bound_input_ref(3)};
    ex.child = stream_iter({employee});
    ex.output_names = {"salary", "Count(id)", "Sum(id)", "Max(id)", "Min(id)"};
    expect_table("test_executor_hash_agg_works", try_collect(ex.execute()),
                 {"+--------+-----------+---------+---------+---------+", "| salary | Count(id) | Sum(id) | Max(id) | Min(id) |",
                  "+--------+-----------+---------+---------+---------+", "| 100    | 2         | 3       | 2       | 1       |",
                  "| 200    | 1         | 3       | 3       | 3       |", "| 400    | 1         | 4       | 4       | 4       |",
                  "+--------+-----------+---------+---------+---------+"});
  }
  { // select * from employee where id = 1   (mod.rs:264-285 keeps first_name = "Bill")
    FilterExecutor ex;
    ex.ctx = ctx;
    ex.expr = BoundExpr::binary_op(BinaryOperator::Eq, build_bound_input_ref(0), BoundExpr::constant(ScalarValue::Int64(1)));
    ex.child = stream_iter({employee});
    expect_table("test_executor_works (filter id = 1)", try_collect(ex.execute()),
                 {"+----+------------+-----------+--------+", "| id | first_name | last_name | salary |",
                  "+----+------------+-----------+--------+", "| 1  | Bill       | Hopkins   | 100    |",
                  "+----+------------+-----------+--------+"});
  }
  { // select * from employee order by id desc   (mod.rs:368-395: offset 2 limit 1 -> id 2)
    OrderExecutor ex;
    ex.ctx = ctx;
    ex.order_by = {BoundOrderBy{build_bound_input_ref(0), false}};
    ex.child = stream_iter({employee});
    auto out = try_collect(ex.execute());
    expect_table("test_executor_order_works (order by id desc)", out,
                 {"+----+------------+-----------+--------+", "| id | first_name | last_name | salary |",
                  "+----+------------+-----------+--------+", "| 4  | Von        | Mill      | 400    |",
                  "| 3  | John       | Travis    | 200    |", "| 2  | Gregg      | Langford  | 100    |",
                  "| 1  | Bill       | Hopkins   | 100    |", "+----+------------+-----------+--------+"});
    if (out.size() != 1 || out[0].columns[0]->value_to_string(2) != "2") {
      failures++;
      std::printf("FAIL order offset 2 limit 1\n");
    }
  }
  // ---- `group` > 1: Filter / Project / the probe side of HashJoin hand several child batches to the *_push_many entry
  //      points at once and must yield the SAME stream of batches — one per input batch, in order (filter.rs:15-24,
  //      project.rs:15-27, hash_join.rs:284-291) — as the per-batch loop.  The reference's own tables, cut into 1-2 row batches.
  {
    auto sch = std::make_shared<Schema>(Schema{{"id", DataType::Int64, false}, {"salary", DataType::Int64, false}});
    std::vector<RecordBatch> parts;
    for (int64_t i = 0; i < 7; i++) parts.push_back(RecordBatch::try_new(sch, {Int64Array({i, i + 10}), Int64Array({100 * (i % 3), 100 * ((i + 1) % 3)})}));
    for (size_t group : {(size_t)0, (size_t)3, (size_t)16}) {
      FilterExecutor fe;
      fe.ctx = ctx;
      fe.expr = BoundExpr::binary_op(BinaryOperator::Gt, build_bound_input_ref(1), BoundExpr::constant(ScalarValue::Int64(50)));
      fe.child = stream_iter(parts);
      fe.group = group;
      ProjectExecutor pe;
      pe.ctx = ctx;
      pe.exprs = {build_bound_input_ref(1), BoundExpr::binary_op(BinaryOperator::Plus, build_bound_input_ref(0), BoundExpr::constant(ScalarValue::Int64(1)))};
      pe.child = fe.execute();
      pe.output_names = {"salary", "id1"};
      pe.group = group;
      std::vector<RecordBatch> out = try_collect(pe.execute());
      bool ok = out.size() == parts.size(); // one output batch per input batch, empty ones included
      std::string got;
      for (size_t b = 0; ok && b < out.size(); b++)
        for (int64_t r = 0; r < out[b].num_rows(); r++) got += out[b].columns[0]->value_to_string(r) + "," + out[b].columns[1]->value_to_string(r) + ";";
      const std::string exp = "100,11;100,2;200,12;200,3;100,14;100,5;200,15;200,6;100,17;";
      if (!ok || got != exp) {
        failures++;
        std::printf("FAIL filter+project with group %zu: %zu batches, rows %s\n", group, out.size(), got.c_str());
      } else {
        std::printf("ok   filter+project, %zu batches per call\n", group);
      }
    }
    for (size_t depth : {(size_t)1, (size_t)4}) { // the same streams through push_async / batch_wait, `depth` tickets in flight
      FilterExecutor fe;
      fe.ctx = ctx;
      fe.expr = BoundExpr::binary_op(BinaryOperator::Gt, build_bound_input_ref(1), BoundExpr::constant(ScalarValue::Int64(50)));
      fe.child = stream_iter(parts);
      fe.depth = depth;
      std::vector<RecordBatch> out = try_collect(fe.execute());
      bool ok = out.size() == parts.size();
      std::string got;
      for (size_t b = 0; ok && b < out.size(); b++)
        for (int64_t r = 0; r < out[b].num_rows(); r++) got += out[b].columns[0]->value_to_string(r) + "," + out[b].columns[1]->value_to_string(r) + ";";
      const std::string exp = "10,100;1,100;11,200;2,200;13,100;4,100;14,200;5,200;16,100;";
      if (!ok || got != exp) {
        failures++;
        std::printf("FAIL filter with %zu tickets in flight: %zu batches, rows %s\n", depth, out.size(), got.c_str());
      } else {
        std::printf("ok   filter, push_async with %zu tickets in flight\n", depth);
      }
      { // ... and Filter -> Project, both through their push_async (project.rs:15-27 over filter.rs:15-24)
        FilterExecutor f2;
        f2.ctx = ctx;
        f2.expr = BoundExpr::binary_op(BinaryOperator::Gt, build_bound_input_ref(1), BoundExpr::constant(ScalarValue::Int64(50)));
        f2.child = stream_iter(parts);
        f2.depth = depth;
        ProjectExecutor pe;
        pe.ctx = ctx;
        pe.exprs = {build_bound_input_ref(1), BoundExpr::binary_op(BinaryOperator::Plus, build_bound_input_ref(0), BoundExpr::constant(ScalarValue::Int64(1)))};
        pe.child = f2.execute();
        pe.output_names = {"salary", "id1"};
        pe.depth = depth;
        std::vector<RecordBatch> out2 = try_collect(pe.execute());
        std::string got2;
        for (size_t b = 0; b < out2.size(); b++)
          for (int64_t r = 0; r < out2[b].num_rows(); r++) got2 += out2[b].columns[0]->value_to_string(r) + "," + out2[b].columns[1]->value_to_string(r) + ";";
        if (out2.size() != parts.size() || got2 != "100,11;100,2;200,12;200,3;100,14;100,5;200,15;200,6;100,17;") {
          failures++;
          std::printf("FAIL filter+project with %zu tickets in flight: %zu batches, rows %s\n", depth, out2.size(), got2.c_str());
        } else {
          std::printf("ok   filter+project, push_async with %zu tickets in flight\n", depth);
        }
      }
      for (JoinType jt : {JoinType::Inner, JoinType::Left}) {
        TestChild t = build_test_child(jt);
        RecordBatch rb = build_table_i32({"a2", {10, 20, 30}}, {"b1", {4, 5, 6}}, {"c2", {70, 80, 90}});
        HashJoinExecutor ex;
        ex.ctx = ctx;
        ex.left_child = std::move(t.left);
        ex.right_child = stream_iter({rb.slice(0, 1), rb.slice(1, 1), rb.slice(2, 1)});
        ex.join_type = jt;
        ex.join_condition.on = {{build_bound_input_ref(1), build_bound_input_ref(1)}};
        ex.join_output_schema = t.schema;
        ex.num_left_columns = 3;
        ex.depth = depth;
        std::vector<std::string> exp2 = {"+------+------+------+------+------+------+", "| l.a1 | l.b1 | l.c1 | r.a2 | r.b1 | r.c2 |",
                                         "+------+------+------+------+------+------+", "| 1    | 4    | 7    | 10   | 4    | 70   |",
                                         "| 2    | 5    | 8    | 20   | 5    | 80   |", "| 3    | 5    | 9    | 20   | 5    | 80   |"};
        if (jt == JoinType::Left) {
          exp2.push_back("| 0    | 0    | 10   |      |      |      |");
          exp2.push_back("| 4    | 8    | 10   |      |      |      |");
        }
        exp2.push_back("+------+------+------+------+------+------+");
        expect_table(jt == JoinType::Inner ? "inner join, async probe batches" : "left join, async probe batches", try_collect(ex.execute()), exp2);
      }
    }
    for (size_t group : {(size_t)2, (size_t)8}) { // test_inner_join_results / test_left_join_results with the probe side in three batches
      for (JoinType jt : {JoinType::Inner, JoinType::Left}) {
        TestChild t = build_test_child(jt);
        RecordBatch rb = build_table_i32({"a2", {10, 20, 30}}, {"b1", {4, 5, 6}}, {"c2", {70, 80, 90}});
        HashJoinExecutor ex;
        ex.ctx = ctx;
        ex.left_child = std::move(t.left);
        ex.right_child = stream_iter({rb.slice(0, 1), rb.slice(1, 1), rb.slice(2, 1)});
        ex.join_type = jt;
        ex.join_condition.on = {{build_bound_input_ref(1), build_bound_input_ref(1)}};
        ex.join_output_schema = t.schema;
        ex.num_left_columns = 3;
        ex.group = group;
        std::vector<std::string> exp = {"+------+------+------+------+------+------+", "| l.a1 | l.b1 | l.c1 | r.a2 | r.b1 | r.c2 |",
                                        "+------+------+------+------+------+------+", "| 1    | 4    | 7    | 10   | 4    | 70   |",
                                        "| 2    | 5    | 8    | 20   | 5    | 80   |", "| 3    | 5    | 9    | 20   | 5    | 80   |"};
        if (jt == JoinType::Left) {
          exp.push_back("| 0    | 0    | 10   |      |      |      |");
          exp.push_back("| 4    | 8    | 10   |      |      |      |");
        }
        exp.push_back("+------+------+------+------+------+------+");
        expect_table(jt == JoinType::Inner ? "inner join, grouped probe batches" : "left join, grouped probe batches", try_collect(ex.execute()), exp);
      }
    }
  }
  { // error behaviour: HashAgg without input panics in the reference (hash_agg.rs:125) -> InternalError here
    HashAggExecutor ex;
    ex.ctx = ctx;
    ex.agg_funcs = {BoundAggFunc{AggFunc::Count, {build_bound_input_ref(0)}, DataType::Int64, false}};
    ex.group_by = {build_bound_input_ref(0)};
    ex.child = stream_iter({});
    try {
      try_collect(ex.execute());
      failures++;
      std::printf("FAIL empty hash agg did not fail\n");
    } catch (const ExecutorError &e) {
      if (e.kind != ExecutorError::InternalError) failures++;
      std::printf("ok   hash agg without input -> InternalError(%s)\n", e.what());
    }
  }
  // ---- limit.rs:97-117: the six (inputs, offset, limit, outputs) cases of `limit`, as data
  {
    struct LimitCase {
      std::vector<std::pair<int, int>> inputs; // half-open ranges, one chunk each (range_to_chunk, limit.rs:119-123)
      int64_t offset, limit;
      std::vector<std::pair<int, int>> outputs;
    };
    const std::vector<LimitCase> cases = {
        {{{0, 6}}, 1, 4, {{1, 5}}},
        {{{0, 6}}, 0, 10, {{0, 6}}},
        {{{0, 6}}, 10, 0, {}},
        {{{0, 2}, {2, 4}, {4, 6}}, 1, 4, {{1, 2}, {2, 4}, {4, 5}}},
        {{{0, 2}, {2, 4}, {4, 6}}, 1, 2, {{1, 2}, {2, 3}}},
        {{{0, 2}, {2, 4}, {4, 6}}, 3, 0, {}},
    };
    auto schema = std::make_shared<Schema>(Schema{{"a", DataType::Int32, false}});
    auto chunk = [&](std::pair<int, int> r) {
      std::vector<int32_t> v;
      for (int x = r.first; x < r.second; x++) v.push_back(x);
      return RecordBatch::try_new(schema, {Int32Array(v)});
    };
    int ci = 0;
    for (const LimitCase &c : cases) {
      std::vector<RecordBatch> in;
      for (auto r : c.inputs) in.push_back(chunk(r));
      LimitExecutor ex;
      ex.ctx = ctx;
      ex.offset = c.offset;
      ex.limit = c.limit;
      ex.child = stream_iter(in);
      auto out = try_collect(ex.execute());
      bool same = out.size() == c.outputs.size(); // the reference compares the batches one by one (assert_eq on Vec<RecordBatch>)
      for (size_t i = 0; same && i < out.size(); i++) {
        same = out[i].num_rows() == c.outputs[i].second - c.outputs[i].first;
        for (int64_t r = 0; same && r < out[i].num_rows(); r++)
          same = out[i].columns[0]->value_to_string(r) == std::to_string(c.outputs[i].first + (int)r);
      }
      if (same) std::printf("ok   limit case %d\n", ci);
      else { failures++; std::printf("FAIL limit case %d\n", ci); }
      ci++;
    }
  }
  { // select id + 1, salary from employee   (ProjectExecutor, project.rs:13-28: one output batch per input batch)
    ProjectExecutor ex;
    ex.ctx = ctx;
    ex.exprs = {BoundExpr::binary_op(BinaryOperator::Plus, build_bound_input_ref(0), BoundExpr::constant(ScalarValue::Int64(1))),
                build_bound_input_ref(3)};
    ex.child = stream_iter({employee, employee});
    ex.output_names = {"id + 1", "salary"};
    auto out = try_collect(ex.execute());
    if (out.size() != 2) { failures++; std::printf("FAIL project: %zu batches for 2 input batches\n", out.size()); }
    expect_table("project id + 1, salary (two chunks)", out,
                 {"+--------+--------+", "| id + 1 | salary |", "+--------+--------+", "| 2      | 100    |", "| 3      | 100    |",
                  "| 4      | 200    |", "| 5      | 400    |", "| 2      | 100    |", "| 3      | 100    |", "| 4      | 200    |",
                  "| 5      | 400    |", "+--------+--------+"});
  }
  { // select count(id), sum(salary), max(salary), min(id) from employee   (SimpleAggExecutor over two chunks: one row)
    SimpleAggExecutor ex;
    ex.ctx = ctx;
    ex.agg_funcs = {BoundAggFunc{AggFunc::Count, {build_bound_input_ref(0)}, DataType::Int64, false},
                    BoundAggFunc{AggFunc::Sum, {build_bound_input_ref(3)}, DataType::Int64, false},
                    BoundAggFunc{AggFunc::Max, {build_bound_input_ref(3)}, DataType::Int64, false},
                    BoundAggFunc{AggFunc::Min, {build_bound_input_ref(0)}, DataType::Int64, false}};
    ex.child = stream_iter({employee, employee});
    ex.output_names = {"Count(id)", "Sum(salary)", "Max(salary)", "Min(id)"};
    expect_table("simple agg over two chunks", try_collect(ex.execute()),
                 {"+-----------+-------------+-------------+---------+", "| Count(id) | Sum(salary) | Max(salary) | Min(id) |",
                  "+-----------+-------------+-------------+---------+", "| 8         | 1600        | 400         | 1       |",
                  "+-----------+-------------+-------------+---------+"});
  }
  // ---- ExecutorBuilder over plan trees (executor/mod.rs:36-56, 87-200): one executor per node, and the
  //      PhysicalHashAgg(PhysicalHashJoin[Inner](l, PhysicalFilter?(r))) peephole that reaches sqlrs_join_agg_*
  { // mod.rs:368-395  select * from employee order by id desc offset 2 limit 1
    ExecutorBuilder eb{ctx};
    PlanRef plan = PlanNode::limit_node(1, 2, PlanNode::order({BoundOrderBy{build_bound_input_ref(0), false}}, PlanNode::table_scan({employee})));
    expect_table("builder: order by id desc offset 2 limit 1", try_collect(eb.build(plan)),
                 {"+----+------------+-----------+--------+", "| id | first_name | last_name | salary |",
                  "+----+------------+-----------+--------+", "| 2  | Gregg      | Langford  | 100    |",
                  "+----+------------+-----------+--------+"});
  }
  { // the headline plan shape: select d.k, count(f.v), sum(f.v) from f join d on f.k = d.k where f.v > 5 group by d.k
    auto dsch = std::make_shared<Schema>(Schema{{"k", DataType::Int64, false}});
    auto fsch = std::make_shared<Schema>(Schema{{"k", DataType::Int64, false}, {"v", DataType::Int64, false}});
    auto make_plan = [&](const RecordBatch &dim, const std::vector<RecordBatch> &fact) {
      std::vector<ColumnCatalog> out = build_table_schema("d", dim, false);
      for (auto &c : build_table_schema("f", fact[0], false)) out.push_back(c);
      JoinCondition cond;
      cond.on = {{build_bound_input_ref(0), build_bound_input_ref(0)}};
      PlanRef probe = PlanNode::filter(BoundExpr::binary_op(BinaryOperator::Gt, build_bound_input_ref(1), BoundExpr::constant(ScalarValue::Int64(5))),
                                       PlanNode::table_scan(fact));
      PlanRef join = PlanNode::hash_join(JoinType::Inner, cond, out, 1, PlanNode::table_scan({dim}), probe);
      return PlanNode::hash_agg({BoundAggFunc{AggFunc::Count, {build_bound_input_ref(2)}, DataType::Int64, false},
                                 BoundAggFunc{AggFunc::Sum, {build_bound_input_ref(2)}, DataType::Int64, false}},
                                {build_bound_input_ref(0)}, join, {"d.k", "Count(f.v)", "Sum(f.v)"});
    };
    RecordBatch dim = RecordBatch::try_new(dsch, {Int64Array({3, 1, 2, 9})});
    RecordBatch f1 = RecordBatch::try_new(fsch, {Int64Array({2, 1, 2, 7, 3}), Int64Array({10, 6, 4, 8, 9})});
    RecordBatch f2 = RecordBatch::try_new(fsch, {Int64Array({1, 2, 3, 3}), Int64Array({5, 7, 6, 20})});
    const std::vector<std::string> expected = {"+-----+------------+----------+", "| d.k | Count(f.v) | Sum(f.v) |", "+-----+------------+----------+",
                                               "| 2   | 2          | 17       |", "| 1   | 1          | 6        |",
                                               "| 3   | 3          | 35       |", "+-----+------------+----------+"};
    ExecutorBuilder plain{ctx};
    plain.fuse_join_agg = false; // Filter, HashJoin, HashAgg: three executors, as executor/mod.rs builds them
    expect_table("builder: HashAgg(HashJoin(dim, Filter(fact))) as three operators", try_collect(plain.build(make_plan(dim, {f1, f2}))), expected);
    ExecutorBuilder fused{ctx};
    expect_table("builder: the same plan through the join_agg peephole", try_collect(fused.build(make_plan(dim, {f1, f2}))), expected);
    if (plain.rewrites != 0 || fused.rewrites != 1) { failures++; std::printf("FAIL peephole: rewrites %d / %d\n", plain.rewrites, fused.rewrites); }
    // at a size where the library's fused route applies (unique build keys, group key = join key, probe-side
    // arguments): both shapes must agree row for row and the rewritten plan must report fused batches
    const int64_t nd = 3000, nf = 400000;
    std::vector<int64_t> dk(nd), fk(nf), fv(nf);
    uint64_t x = 88172645463325252ull;
    auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (int64_t i = 0; i < nd; i++) dk[i] = (i * 7919) % nd; // a permutation of 0..nd-1 (7919 is coprime to 3000)
    for (int64_t i = 0; i < nf; i++) { fk[i] = (int64_t)(rnd() % (uint64_t)(nd + 500)); fv[i] = (int64_t)(rnd() % 11); }
    RecordBatch bigd = RecordBatch::try_new(dsch, {Int64Array(dk)});
    RecordBatch bigf = RecordBatch::try_new(fsch, {Int64Array(fk), Int64Array(fv)});
    ExecutorBuilder p2{ctx}, f2b{ctx};
    p2.fuse_join_agg = false;
    auto a = try_collect(p2.build(make_plan(bigd, {bigf})));
    auto b = try_collect(f2b.build(make_plan(bigd, {bigf})));
    bool same = a.size() == 1 && b.size() == 1 && a[0].num_rows() == b[0].num_rows() && a[0].num_rows() > 0;
    for (size_t c = 0; same && c < 3; c++)
      for (int64_t r = 0; same && r < a[0].num_rows(); r++) same = a[0].columns[c]->value_to_string(r) == b[0].columns[c]->value_to_string(r);
    if (same && f2b.last_fused_batches >= 1) std::printf("ok   builder: 4e5-row plan, three operators == join_agg peephole (%lld groups, fused batches %lld)\n",
                                                         (long long)a[0].num_rows(), (long long)f2b.last_fused_batches);
    else { failures++; std::printf("FAIL builder: large plan (same %d, fused batches %lld)\n", (int)same, (long long)f2b.last_fused_batches); }
  }
  { // PhysicalHashAgg(PhysicalFilter(scan)): select k, count(v), sum(v) from f where v > 5 group by k
    auto fsch = std::make_shared<Schema>(Schema{{"k", DataType::Int64, false}, {"v", DataType::Int64, false}});
    auto make_plan = [&](const std::vector<RecordBatch> &fact) {
      PlanRef flt = PlanNode::filter(BoundExpr::binary_op(BinaryOperator::Gt, build_bound_input_ref(1), BoundExpr::constant(ScalarValue::Int64(5))),
                                     PlanNode::table_scan(fact));
      return PlanNode::hash_agg({BoundAggFunc{AggFunc::Count, {build_bound_input_ref(1)}, DataType::Int64, false},
                                 BoundAggFunc{AggFunc::Sum, {build_bound_input_ref(1)}, DataType::Int64, false}},
                                {build_bound_input_ref(0)}, flt, {"k", "Count(v)", "Sum(v)"});
    };
    RecordBatch f1 = RecordBatch::try_new(fsch, {Int64Array({2, 1, 2, 7, 3}), Int64Array({10, 6, 4, 8, 9})});
    RecordBatch f2 = RecordBatch::try_new(fsch, {Int64Array({1, 2, 3, 3}), Int64Array({5, 7, 6, 20})});
    const std::vector<std::string> expected = {"+---+----------+--------+", "| k | Count(v) | Sum(v) |", "+---+----------+--------+",
                                               "| 2 | 2        | 17     |", "| 1 | 1        | 6      |", "| 7 | 1        | 8      |",
                                               "| 3 | 3        | 35     |", "+---+----------+--------+"};
    ExecutorBuilder plain{ctx}, fused{ctx};
    plain.fuse_join_agg = false;
    expect_table("builder: HashAgg(Filter(scan)) as two operators", try_collect(plain.build(make_plan({f1, f2}))), expected);
    expect_table("builder: the same plan with the filter handed to the aggregate", try_collect(fused.build(make_plan({f1, f2}))), expected);
    if (plain.rewrites != 0 || fused.rewrites != 1) { failures++; std::printf("FAIL agg-filter peephole: rewrites %d / %d\n", plain.rewrites, fused.rewrites); }
  }
  std::printf("%s (%d failure%s)\n", failures ? "FAILED" : "PASSED", failures, failures == 1 ? "" : "s");
  return failures ? 1 : 0;
}
