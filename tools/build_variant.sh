#!/bin/bash
# usage: tools/build_variant.sh <name> [-DFLAG ...]   builds jsnoop_kernels.hip with extra flags and links it with the objects of the
# regular build into gpurun_out/variants/lib_<name>.so (for A/B experiments on the GPU box: tools/exp_bench.sh <so> <tag>)
set -e
NAME=$1; shift
HERE=$(cd $(dirname $0)/.. && pwd)
SRC=$HERE/jpegsnoop_amd/csrc
OUT=$HERE/gpurun_variants; mkdir -p $OUT /tmp/var_$NAME
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -Wno-unused-const-variable"
make -s -j8 -C $SRC
/opt/rocm/bin/hipcc $FLAGS "$@" -c ${KSRC:-$SRC/jsnoop_kernels.hip} -I$SRC -o /tmp/var_$NAME/k.o
OBJS=$(ls $SRC/build/*.o | grep -v jsnoop_kernels.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_$NAME.so /tmp/var_$NAME/k.o $OBJS -L/opt/rocm/lib -lrocprofiler-sdk-roctx -Wl,-rpath,/opt/rocm/lib
echo built $OUT/lib_$NAME.so
