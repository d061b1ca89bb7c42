#!/bin/bash
# round 3, first GPU call: new exchange tests, composite probe microbenchmark, default bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_probe_filter.py tests/test_gpu_bench_multirank.py -m gpu -x -q \
  -k "hash_partition or guards or bench_" --durations=15 > gpurun_out/r03a_pytest.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r03a_pytest.txt
tail -30 gpurun_out/r03a_pytest.txt
timeout 300 ./tools/ubench p > gpurun_out/r03_probe_composite.txt 2>&1
tail -60 gpurun_out/r03_probe_composite.txt
timeout 600 python bench.py > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err
echo "bench rc=$?"
tail -45 gpurun_out/r03a_bench.err
