#!/bin/bash
# usage: tools/exp_env.sh <tag> VAR=val [VAR=val ...] -- [bench args]   quick bench under environment settings (stage split + wall ms per step)
TAG=$1; shift
ENVS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done
[ "$1" == "--" ] && shift
env "${ENVS[@]}" python bench.py --steps 10 --warmup 2 --cpu-seconds 0 --no-extras "$@" 2>&1 | tail -1 > gpurun_out/bench_$TAG.json
python - <<PY
import json
d = json.load(open("gpurun_out/bench_$TAG.json"))
print("$TAG", d["value"], d["bit_exact"], d["ms_per_step"], d["roofline"]["stages_ms"])
PY
