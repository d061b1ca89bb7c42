#!/bin/bash
# Build a compile-time VARIANT of the library next to the default one, HERE (hipcc cross-compiles without a GPU), so that
# a gpurun call can compare the two in one process without paying for the build on the GPU box:
#   FILES="radix_part.hip agg_partition.hip" FLAGS=-DSLIM_AOS NAME=aos bash tools/build_variant.sh
#   -> tools/_bin/lib_aos.so ;  on the GPU box: LIB_A=sqlrs_amd/csrc/libsqlrs_hip.so LIB_B=tools/_bin/lib_aos.so python tools/ab_two_builds.py
cd "$(dirname "$0")/.."
mkdir -p tools/_bin
for f in $FILES; do touch sqlrs_amd/csrc/$f; done
SQLRS_EXTRA_CFLAGS="$FLAGS" python -m sqlrs_amd.build > /dev/null 2>&1 || { echo "variant build failed"; exit 1; }
cp sqlrs_amd/csrc/libsqlrs_hip.so tools/_bin/lib_${NAME}.so
for f in $FILES; do touch sqlrs_amd/csrc/$f; done
python -m sqlrs_amd.build > /dev/null 2>&1 || { echo "default rebuild failed"; exit 1; }
ls -la tools/_bin/lib_${NAME}.so sqlrs_amd/csrc/libsqlrs_hip.so
