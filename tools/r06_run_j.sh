mkdir -p gpurun_out/r06_j
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r06_j/pytest.log 2>&1; tail -3 gpurun_out/r06_j/pytest.log
timeout 900 python tools/fuzz_damaged_log.py 600 103 1 > gpurun_out/r06_j/fuzz_big.txt 2>&1; grep -A3 "MISMATCH\|LOG\|exception" gpurun_out/r06_j/fuzz_big.txt | head; tail -1 gpurun_out/r06_j/fuzz_big.txt
timeout 900 python tools/fuzz_damaged_log.py 1500 107 > gpurun_out/r06_j/fuzz_small.txt 2>&1; grep -A3 "MISMATCH\|LOG\|exception" gpurun_out/r06_j/fuzz_small.txt | head; tail -1 gpurun_out/r06_j/fuzz_small.txt
python bench.py > gpurun_out/r06_j/bench.json 2>/dev/null; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06_j/bench.json").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], {k:(v.get("ms"),v.get("call_ms"),v.get("side_mode"),v.get("side_outputs_equal_oracle")) for k,v in d["damaged_files"].items()})
PY
