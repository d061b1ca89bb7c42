#!/bin/bash
# round 3, second GPU call: optimistic key statistics + 8192-row tiles (agg tests, fuzz, full sizes), host mirror, bench
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_probe_filter.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py tests/test_gpu_bench_sizes.py -m gpu -x -q \
  --durations=12 > gpurun_out/r03b_pytest.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r03b_pytest.txt
tail -25 gpurun_out/r03b_pytest.txt
./host/test_reference_executor 2>&1 | tail -8
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r03b_bench.json 2> gpurun_out/r03b_bench.err
echo "bench rc=$?"
grep -E "check|ms/step|C4|variant|kernel classes" gpurun_out/r03b_bench.err | cut -c1-400
