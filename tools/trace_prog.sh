cd /tmp && export TMPDIR=/tmp
JSNOOP_PG_LANES=64 rocprofv3 --kernel-trace --output-format csv -d /tmp/tp -o p -- python $GRAFT_REPO_ROOT/tools/prog_batch_run.py 64 2 > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/tp/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_prog_scan' in r['Kernel_Name']]
for r in rows[-6:]:
    print(r['Kernel_Name'][:20], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, 'ms grid', r['Grid_Size_X'])
PY
