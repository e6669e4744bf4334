#!/bin/bash
# usage (GPU box): tools/trace_damaged_call.sh  -- kernel + copy trace of tools/dbg_damaged_timing.py; prints the device activity of its LAST call (1080p + RSTn, an inserted RSTn: chunked side pass)
OUT=$PWD/gpurun_out/trace_damaged; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/p -o p -- python $GRAFT_REPO_ROOT/tools/dbg_damaged_timing.py > $OUT/run.log 2>&1)
K=$(find $OUT/p -name '*kernel_trace.csv' | head -1); M=$(find $OUT/p -name '*memory_copy_trace.csv' | head -1)
python - "$K" "$M" <<'PY' | tee $OUT/last_call.txt
import csv, sys
ev=[]
for r in csv.DictReader(open(sys.argv[1])): ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:70]))
try:
    for r in csv.DictReader(open(sys.argv[2])): ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), "COPY %s" % (r.get('Direction', r.get('Name','')))))
except Exception as e: print("no copy trace", e)
ev.sort()
idx=[k for k,e in enumerate(ev) if e[2].startswith('k_clear3')]
a=idx[-1]; t0=ev[a][0]
for s,e,n in ev[a:]: print("%8.1f us +%7.1f  %s" % ((s-t0)/1e3, (e-s)/1e3, n))
PY
rm -rf $OUT/p
