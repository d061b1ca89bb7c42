#!/bin/bash
# slivers per wave trip of the LDS probe kernel (compile-time), C3 sparse keys per value
cd $GRAFT_REPO_ROOT
for q in 4 8 2; do
  touch sqlrs_amd/csrc/join.hip
  SQLRS_EXTRA_CFLAGS="-DLJ_Q_N=$q" python -m sqlrs_amd.build > /dev/null 2>&1 || { echo "build failed $q"; continue; }
  echo "== LJ_Q=$q"
  VAR=SQLRS_LJ_RPI VALUES=1024,256 REPS=1 python tools/c3_sparse.py 2>&1 | tail -2
done
touch sqlrs_amd/csrc/join.hip
python -m sqlrs_amd.build > /dev/null 2>&1
