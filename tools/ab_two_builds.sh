#!/bin/bash
# In-process A/B of two BUILDS of the library (compile-time variants): build B with EXTRA_B flags for file FILE_B,
# keep both .so files, run tools/ab_two_builds.py (C5 steps alternating between the two on the same inputs).
#   FILE_B=radix_part.hip EXTRA_B=-DRP_PREFETCH_LATE bash tools/ab_two_builds.sh      (FILE_B: one or more files)
cd $GRAFT_REPO_ROOT
FILE_B=${FILE_B:-radix_part.hip}
cp sqlrs_amd/csrc/libsqlrs_hip.so /tmp/libA.so
for f in $FILE_B; do touch sqlrs_amd/csrc/$f; done
SQLRS_EXTRA_CFLAGS="$EXTRA_B" python -m sqlrs_amd.build > /dev/null 2>&1 || { echo "build B failed"; exit 1; }
cp sqlrs_amd/csrc/libsqlrs_hip.so /tmp/libB.so
for f in $FILE_B; do touch sqlrs_amd/csrc/$f; done
python -m sqlrs_amd.build > /dev/null 2>&1
LIB_A=/tmp/libA.so LIB_B=/tmp/libB.so python tools/ab_two_builds.py
