OUT=$PWD/gpurun_out/trace_batch; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/p -o p -- python $GRAFT_REPO_ROOT/tools/batch_report_timing.py 128 > $OUT/run.log 2>&1)
K=$(find $OUT/p -name '*kernel_trace.csv' | head -1)
python - "$K" <<'PY'
import csv, sys, collections
rows=sorted(csv.DictReader(open(sys.argv[1])), key=lambda r:int(r['Start_Timestamp']))
sm=[r for r in rows if r['Kernel_Name'].startswith('k_side_maps')]
wr=[r for r in rows if 'k_write<' in r['Kernel_Name'] and 'true>' in r['Kernel_Name'] and 'k_write2' not in r['Kernel_Name']]
print("k_side_maps launches", len(sm), "k_write<.,true>", len(wr))
if sm:
    t0=int(sm[1]['Start_Timestamp']); t1=int(sm[-1]['End_Timestamp'])
    print("span of side maps 2..last: %.2f ms" % ((t1-t0)/1e6))
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in sm]; print("k_side_maps us: median %.1f max %.1f" % (sorted(d)[len(d)//2], max(d)))
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in wr]; print("side walk us: median %.1f max %.1f" % (sorted(d)[len(d)//2], max(d)))
    print("queues", collections.Counter(r.get('Queue_Id','?') for r in sm))
    # concurrency: max overlapping side-walk kernels
    ev=[]
    for r in wr+sm: ev.append((int(r['Start_Timestamp']),1)); ev.append((int(r['End_Timestamp']),-1))
    ev.sort(); c=m=0
    for t,dlt in ev: c+=dlt; m=max(m,c)
    print("max concurrent side kernels", m)
PY
tail -1 $OUT/run.log
rm -rf $OUT/p
