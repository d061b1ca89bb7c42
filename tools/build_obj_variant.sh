#!/bin/bash
# One translation unit rebuilt with extra flags and linked against the other objects of the default build:
#   FILE=agg_partition FLAGS="-DSLIM_DBG=1" NAME=dbg1 bash tools/build_obj_variant.sh   -> tools/_bin/lib_dbg1.so
# (several of these can run at once; tools/build_variant.sh rebuilds in place and cannot)
cd "$(dirname "$0")/.."
mkdir -p tools/_bin
C=sqlrs_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-result -I include -I $C $FLAGS -c $C/$FILE.hip -o tools/_bin/${FILE}_${NAME}.o || exit 1
OBJS=$(ls $C/build/*.o | grep -v "/$FILE.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS tools/_bin/${FILE}_${NAME}.o -o tools/_bin/lib_${NAME}.so && ls -la tools/_bin/lib_${NAME}.so
