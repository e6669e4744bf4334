#!/bin/bash
# usage (GPU box): tools/trace_step.sh <tag> [bench args]  -- kernel trace of the one-stream bench; prints the launches of the LAST decode with their durations
TAG=$1; shift; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/p -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-extras --no-split "$@" > $OUT/bench.log 2>&1)
F=$(find $OUT/p -name '*kernel_trace.csv' | head -1)
python - "$F" <<'PY' | tee $OUT/last_decode.txt
import csv, sys
rows=sorted(csv.DictReader(open(sys.argv[1])), key=lambda r:int(r['Start_Timestamp']))
# last decode = from the last k_clear3 on
idx=[k for k,r in enumerate(rows) if r['Kernel_Name'].startswith('k_clear3')]
last=rows[idx[-1]:] if idx else rows[-30:]
t0=int(last[0]['Start_Timestamp'])
for r in last:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    print("%9.1f us  +%8.1f us  %s  grid %s" % ((s-t0)/1e3, (e-s)/1e3, r['Kernel_Name'][:44], r.get('Grid_Size', r.get('Grid_Size_X',''))))
PY
rm -rf $OUT/p
