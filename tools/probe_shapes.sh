#!/bin/bash
# dense join probe block shapes (compile-time: rows per lane, blocks per CU), C3 all-hit / half-hit timing per shape
cd $GRAFT_REPO_ROOT
for cfg in "32 2" "16 3" "16 4" "24 2" "8 4" "32 1"; do
  set -- $cfg
  touch sqlrs_amd/csrc/join.hip
  SQLRS_EXTRA_CFLAGS="-DJD_ITEMS_N=$1 -DJD_OCC=$2" python -m sqlrs_amd.build > /dev/null 2>&1 || { echo "build failed $cfg"; continue; }
  echo "== JD_ITEMS=$1 JD_OCC=$2"
  PROBE_SWEEP_ONLY=1000000 timeout 200 python tools/probe_sweep.py 2>&1 | grep "^build"
done
touch sqlrs_amd/csrc/join.hip
python -m sqlrs_amd.build > /dev/null 2>&1
