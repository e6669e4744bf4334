#!/bin/bash
# usage: tools/build_variant.sh <name> [-p patchfile ...] [-DFLAG ...]
#   Builds a VARIANT of the kernels for A/B experiments on the GPU box: a private copy of jsnoop_kernels.hip (or $KSRC) with the given
#   patches of tools/variants/ applied (patch -p1) and the given -D switches set, linked with the objects of the regular build into
#   gpurun_variants/lib_<name>.so (run them with tools/ab_round.sh).  The product source holds no experiment switches: ablations
#   (results wrong, timing valid) and candidate rewrites live in the patch files, tests/test_capi_symbols.py greps for that.
set -e
NAME=$1; shift
HERE=$(cd $(dirname $0)/.. && pwd)
SRC=$HERE/jpegsnoop_amd/csrc
OUT=$HERE/gpurun_variants; W=/tmp/var_$NAME; rm -rf $W; mkdir -p $OUT $W
cp ${KSRC:-$SRC/jsnoop_kernels.hip} $W/jsnoop_kernels.hip
DEFS=()
while [ $# -gt 0 ]; do
  case "$1" in
    -p) (cd $W && patch -s -p1 < $(cd $HERE && realpath $2)); shift 2;;
    *)  DEFS+=("$1"); shift;;
  esac
done
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -Wno-unused-const-variable"
make -s -j8 -C $SRC
/opt/rocm/bin/hipcc $FLAGS "${DEFS[@]}" -c $W/jsnoop_kernels.hip -I$SRC -o $W/k.o
OBJS=$(ls $SRC/build/*.o | grep -v jsnoop_kernels.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_$NAME.so $W/k.o $OBJS -L/opt/rocm/lib -lrocprofiler-sdk-roctx -Wl,-rpath,/opt/rocm/lib
echo built $OUT/lib_$NAME.so
