mkdir -p gpurun_out/r05_m
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05_m/prof -o nodri -- python $GRAFT_REPO_ROOT/tools/prog_nodri_trace.py > $GRAFT_REPO_ROOT/gpurun_out/r05_m/run.txt 2>&1
cd $GRAFT_REPO_ROOT
tail -5 gpurun_out/r05_m/run.txt
find gpurun_out/r05_m/prof -name "*kernel_trace.csv" | head -2
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r05_m/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
out = open("gpurun_out/r05_m/launches.txt", "w")
for r in rows:
    line = "%10.3f ms  +%9.3f ms  %s  grid %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r["Kernel_Name"][:40], r.get("Grid_Size_X", r.get("Grid_Size", "")))
    out.write(line + "\n")
out.close()
print(open("gpurun_out/r05_m/launches.txt").read()[-4000:])
PY
rm -rf gpurun_out/r05_m/prof
