#!/bin/bash
# usage: tools/exp_run.sh <variant.so> <command...>  -- runs a command with an experimental build of the library swapped in
SO=$1; shift
cp jpegsnoop_amd/libjsnoop_gpu.so /tmp/libjsnoop_gpu.orig.so
cp $SO jpegsnoop_amd/libjsnoop_gpu.so
"$@"
cp /tmp/libjsnoop_gpu.orig.so jpegsnoop_amd/libjsnoop_gpu.so
