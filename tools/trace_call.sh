#!/bin/bash
# usage (GPU box): tools/trace_call.sh  -- kernel + memory-copy trace of single-image calls (tools/dbg_call_timing.py); prints the device activity of the last 1080p call
OUT=$PWD/gpurun_out/trace_call; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/p -o p -- python $GRAFT_REPO_ROOT/tools/dbg_call_timing.py > $OUT/run.log 2>&1)
K=$(find $OUT/p -name '*kernel_trace.csv' | head -1); M=$(find $OUT/p -name '*memory_copy_trace.csv' | head -1)
python - "$K" "$M" <<'PY' | tee $OUT/last_call.txt
import csv, sys
ev=[]
for r in csv.DictReader(open(sys.argv[1])): ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60]))
try:
    for r in csv.DictReader(open(sys.argv[2])): ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), "COPY %s %s B" % (r.get('Direction', r.get('Name','')), r.get('Size', r.get('Bytes','')))))
except Exception as e: print("no copy trace", e)
ev.sort()
# calls are separated by k_clear3; take the 6th-from-last clear (a 1080p call: the script runs 3+3 1080p then 3+3 2160p)
idx=[k for k,e in enumerate(ev) if e[2].startswith('k_clear3')]
a=idx[-8]; b=idx[-7]
t0=ev[a][0]
for s,e,n in ev[a:b]: print("%8.1f us +%7.1f  %s" % ((s-t0)/1e3, (e-s)/1e3, n))
PY
rm -rf $OUT/p
