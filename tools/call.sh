# scratch: the command list of the next gpurun call (not part of the tools)
mkdir -p gpurun_out/r05_p
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 > gpurun_out/r05_p/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_p/pytest.log; tail -5 gpurun_out/r05_p/pytest.log
bash tools/ab_env.sh r05_p "chain1|-||" "two1|-|JSNOOP_SYNC_LAUNCHES=2|" "chain2|-||" "two2|-|JSNOOP_SYNC_LAUNCHES=2|" "one1|-|JSNOOP_SYNC_LAUNCHES=1|"
for E in "" "JSNOOP_SYNC_LAUNCHES=2"; do
  echo "== $E" >> gpurun_out/r05_p/sweep.txt
  env $E python tools/small_jobs.py 64 96 128 192 256 2>/dev/null | tail -1 >> gpurun_out/r05_p/sweep.txt
done
python - <<'PY'
import json
lines = open("gpurun_out/r05_p/sweep.txt").read().strip().split("\n")
for i in range(0, len(lines), 2):
    d = json.loads(lines[i + 1])["jobs"]
    print("%-30s" % lines[i], "  ".join("%s: %.3f (s %.2f)" % (k, v["ms"], v["sync"]) for k, v in d.items()))
PY
