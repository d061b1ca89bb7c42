#!/bin/bash
# round 3, re-entry: the whole GPU suite, the default bench line and the round profile on the current tree
TAG=${1:-r03g}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1100 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/${TAG}_pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.txt
tail -22 gpurun_out/${TAG}_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench.err
echo "bench rc=$?"
grep -E "check|ms/step|C2|C3|C4|Order|variant|kernel classes" gpurun_out/${TAG}_bench.err | cut -c1-300
bash tools/profile_round.sh $TAG 2>&1 | tail -40
