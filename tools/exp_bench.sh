#!/bin/bash
# usage: tools/exp_bench.sh <variant.so> <tag>  -- quick bench with an experimental build of the library swapped in (stage split only)
SO=$1; TAG=$2
cp jpegsnoop_amd/libjsnoop_gpu.so /tmp/libjsnoop_gpu.orig.so
cp $SO jpegsnoop_amd/libjsnoop_gpu.so
bash tools/quick_bench.sh $TAG --no-extras
cp /tmp/libjsnoop_gpu.orig.so jpegsnoop_amd/libjsnoop_gpu.so
