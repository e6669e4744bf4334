#!/bin/bash
# usage: tools/quick_bench.sh <tag> [extra bench args]  -- prints value, bit_exact, ms/step and the stage split
TAG=${1:-q}; shift || true
python bench.py --steps 10 --warmup 2 --cpu-seconds 0 "$@" 2>&1 | tail -1 > gpurun_out/bench_$TAG.json
python - <<PY
import json
d = json.load(open("gpurun_out/bench_$TAG.json"))
print("$TAG", d["value"], d["bit_exact"], d["ms_per_step"], d["roofline"]["stages_ms"])
PY
