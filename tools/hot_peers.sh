#!/bin/bash
# hot-key threshold of the LDS bucket passes (compile-time): C4 uniform / Zipf per value
cd $GRAFT_REPO_ROOT
for t in 8 6 4 3; do
  touch sqlrs_amd/csrc/agg_partition.hip
  SQLRS_EXTRA_CFLAGS="-DHOT_MIN_PEERS_N=$t" python -m sqlrs_amd.build > /dev/null 2>&1 || { echo "build failed $t"; continue; }
  echo "== HOT_MIN_PEERS=$t"
  VAR=SQLRS_DENSE_CHUNK_DIV VALUES=2 timeout 300 python tools/c4_zipf.py 2>&1 | tail -2
done
touch sqlrs_amd/csrc/agg_partition.hip
python -m sqlrs_amd.build > /dev/null 2>&1
