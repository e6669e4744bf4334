#!/bin/bash
# usage (on the GPU box): tools/ab_env.sh <tag> "<label>|<variant.so or ->|<ENV=V ENV2=V ...>|<extra bench args>" ...
#   One quick bench per entry on the same box (boxes differ by a few %): the tree's library ("-") or a variant build, with environment presets of the
#   tuning defaults (tools/README.md) and extra bench arguments; results in gpurun_out/<tag>/bench_<label>.json and a one-line summary each.
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
cp jpegsnoop_amd/libjsnoop_gpu.so /tmp/libjsnoop_gpu.orig.so
for E in "$@"; do
  IFS='|' read -r LABEL SO ENVS ARGS <<< "$E"
  [ "$SO" != "-" ] && cp $SO jpegsnoop_amd/libjsnoop_gpu.so
  env $ENVS python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-extras $ARGS 2>$OUT/$LABEL.err | tail -1 > $OUT/bench_$LABEL.json
  cp /tmp/libjsnoop_gpu.orig.so jpegsnoop_amd/libjsnoop_gpu.so
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/bench_*.json"), key=os.path.getmtime):
    try:
        d = json.load(open(f)); s = d.get("one_stream", {})
        print("%-22s" % f.split("bench_")[1][:-5], d["value"], d["bit_exact"], d["ms_per_step"], "one-stream:", s.get("ms_per_step"), s.get("bit_exact"), d["roofline"]["stages_ms"])
    except Exception as e: print(f, "unreadable", e)
PY
