#!/bin/bash
# usage: tools/build_full_variant.sh <name> [-p patchfile ...] [-DFLAG ...]
#   Like build_variant.sh, but for experiments that touch the host side too: a private copy of ALL of jpegsnoop_amd/csrc with the given patches of
#   tools/variants/ applied (patch -p1, paths relative to csrc/) is built whole into gpurun_variants/lib_<name>.so.
set -e
NAME=$1; shift
HERE=$(cd $(dirname $0)/.. && pwd)
W=/tmp/fullvar_$NAME; rm -rf $W; mkdir -p $W/jpegsnoop_amd $W/include $HERE/gpurun_variants
cp -r $HERE/jpegsnoop_amd/csrc $W/jpegsnoop_amd/csrc; cp $HERE/include/*.h $W/include/; rm -rf $W/jpegsnoop_amd/csrc/build
DEFS=""
while [ $# -gt 0 ]; do
  case "$1" in
    -p) (cd $W/jpegsnoop_amd/csrc && patch -s -p1 < $(cd $HERE && realpath $2)); shift 2;;
    *)  DEFS="$DEFS $1"; shift;;
  esac
done
make -s -j16 -C $W/jpegsnoop_amd/csrc EXTRA="$DEFS"
cp $W/jpegsnoop_amd/libjsnoop_gpu.so $HERE/gpurun_variants/lib_$NAME.so
echo built $HERE/gpurun_variants/lib_$NAME.so
