#!/bin/bash
# usage (on the GPU box): tools/ab_kernel_stats.sh <tag> [variant.so ...]  -- rocprofv3 per-kernel averages of the one-stream bench for the tree's library and for
# every variant build, same box, same call; results in gpurun_out/<tag>/<name>_kernel_stats.csv (+ the bench line of each run)
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
run() {   # name
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$1 -o p -- python $GRAFT_REPO_ROOT/bench.py --steps ${AB_STEPS:-10} --warmup 2 --cpu-seconds 0 --no-extras --no-split > $OUT/bench_$1.json 2>$OUT/$1.err)
  F=$(find $OUT/prof_$1 -name '*kernel_stats.csv' | head -1)
  [ -n "$F" ] && cp $F $OUT/$1_kernel_stats.csv
  rm -rf $OUT/prof_$1
  echo "== $1"; tail -1 $OUT/bench_$1.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['bit_exact'], d['roofline']['stages_ms'])"
  cut -c1-60 $OUT/$1_kernel_stats.csv | paste -d, - <(python - <<PY
import csv
for r in csv.reader(open("$OUT/$1_kernel_stats.csv")): print(r[1], r[3], r[4])
PY
) | head -12
}
run tree
for SO in "$@"; do
  N=$(basename $SO .so)
  cp jpegsnoop_amd/libjsnoop_gpu.so /tmp/libjsnoop_gpu.orig.so; cp $SO jpegsnoop_amd/libjsnoop_gpu.so
  run $N
  cp /tmp/libjsnoop_gpu.orig.so jpegsnoop_amd/libjsnoop_gpu.so
done
