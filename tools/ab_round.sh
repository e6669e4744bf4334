#!/bin/bash
# usage (on the GPU box): tools/ab_round.sh <tag> [variant.so ...]  -- the GPU suite on the tree's library, then the quick bench of the tree's
# library and of every variant build (same box, same call: boxes differ by a few %), results in gpurun_out/<tag>/
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
# env: AB_SKIP_TESTS=1 (bench only), AB_ARGS="..." (extra bench arguments for the variants, e.g. --no-split)
if [ -z "$AB_SKIP_TESTS" ]; then timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log; fi
python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-extras 2>$OUT/tree.err | tail -1 > $OUT/bench_tree.json
for SO in "$@"; do
  N=$(basename $SO .so)
  cp jpegsnoop_amd/libjsnoop_gpu.so /tmp/libjsnoop_gpu.orig.so; cp $SO jpegsnoop_amd/libjsnoop_gpu.so
  python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-extras $AB_ARGS 2>$OUT/$N.err | tail -1 > $OUT/bench_$N.json
  cp /tmp/libjsnoop_gpu.orig.so jpegsnoop_amd/libjsnoop_gpu.so
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d = json.load(open(f)); s = d.get("one_stream", {})
        print(f.split("bench_")[1][:-5], d["value"], d["bit_exact"], d["ms_per_step"], "one-stream:", s.get("ms_per_step"), s.get("bit_exact"), d["roofline"]["stages_ms"])
    except Exception as e: print(f, "unreadable", e)
PY
