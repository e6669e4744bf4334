#!/bin/bash
# (experiment 16 of profiles/r06_experiments.txt) the first k_sync launch three times in a row (variant r06_sync3x): what a warm cache is worth to the walks
OUT=$PWD/gpurun_out/sync3x; mkdir -p $OUT; export TMPDIR=/tmp
cp jpegsnoop_amd/libjsnoop_gpu.so /tmp/orig.so; cp gpurun_variants/lib_r06_sync3x.so jpegsnoop_amd/libjsnoop_gpu.so
for N in 32 64 256 1024; do
  (cd /tmp && JSNOOP_CAND=0 JSNOOP_SUB_WL=7 rocprofv3 --kernel-trace --output-format csv -d $OUT/p$N -o p -- python $GRAFT_REPO_ROOT/bench.py --images $N --distinct 16 --steps 3 --warmup 1 --cpu-seconds 0 --no-extras --no-split > $OUT/b$N.log 2>&1)
  F=$(find $OUT/p$N -name '*kernel_trace.csv' | head -1)
  python - "$F" $N <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if r['Kernel_Name'].startswith('void k_sync')]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
print(sys.argv[2], "images: k_sync launches (us), last decode:", [round(x,1) for x in d[-4:]])
PY
  rm -rf $OUT/p$N
done
cp /tmp/orig.so jpegsnoop_amd/libjsnoop_gpu.so
