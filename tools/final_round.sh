#!/bin/bash
# usage (on the GPU box): tools/final_round.sh <tag>  -- the evidence of a round in one call: kernel stats + PMC passes (one-stream launches),
# the default bench line, damaged-file timing, small jobs, call latency, a 5000-case parity fuzz, PMC passes of a progressive batch; everything under gpurun_out/<tag>/
TAG=${1:-rXX_final}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh $TAG > $OUT/profile_round.log 2>&1
python bench.py 2> $OUT/bench_default.err | tail -1 > $OUT/bench_default.json
timeout 400 python tools/fuzz_1080p_timing.py 1000 17 > $OUT/fuzz_1080p_timing.log 2>&1; cp gpurun_out/fuzz_1080p_timing.json $OUT/ 2>/dev/null
python tools/small_jobs.py 1 2 4 8 16 32 48 64 96 128 256 2>/dev/null | tail -1 > $OUT/small_jobs.json
python tools/call_latency.py 2>/dev/null | tail -1 > $OUT/call_latency.json
timeout 420 python tools/fuzz_gpu.py ${FUZZ_CASES:-5000} 4242 > $OUT/fuzz_gpu.log 2>&1; tail -2 $OUT/fuzz_gpu.log
bash tools/pmc_progressive.sh $TAG/prog 64 > $OUT/pmc_progressive.log 2>&1; python tools/prog_batch.py 64 3 2>/dev/null | tail -1 > $OUT/prog_batch64.json
# round 6: damaged files through the drop-in call with a log callback -- log text against the compiled reference, side outputs against the oracle, who produced them, call time
timeout 600 python tools/fuzz_damaged_log.py ${LOG_FUZZ_SMALL:-4000} 101 > $OUT/fuzz_damaged_log_small.txt 2>&1; tail -1 $OUT/fuzz_damaged_log_small.txt
timeout 600 python tools/fuzz_damaged_log.py ${LOG_FUZZ_BIG:-600} 103 1 > $OUT/fuzz_damaged_log_1080p.txt 2>&1; tail -1 $OUT/fuzz_damaged_log_1080p.txt
# ... and the phase stamps of the back end (variant builds of tools/variants/r06_stamps.patch, when they travelled)
for V in r06_stamps r06_stamps_tile; do
  if [ -f gpurun_variants/lib_$V.so ]; then
    cp jpegsnoop_amd/libjsnoop_gpu.so /tmp/orig.so; cp gpurun_variants/lib_$V.so jpegsnoop_amd/libjsnoop_gpu.so
    python tools/backend_stamps.py 256 16 > $OUT/$V.json 2>/dev/null; cat $OUT/$V.json
    cp /tmp/orig.so jpegsnoop_amd/libjsnoop_gpu.so
  fi
done
bash tools/pmc_kernel.sh $TAG/pmc_idct k_idct_color > $OUT/pmc_idct.log 2>&1; grep -A40 "k_idct_color" $OUT/pmc_idct/tree/summary.txt | head -45 > $OUT/pmc_idct_color.txt
python - <<PY
import json
d = json.load(open("$OUT/bench_default.json"))
print("default line:", d["value"], d["unit"], d["ms_per_step"], "ms/step", d["bit_exact"], d["config"]["decode_form"], d["roofline"]["frac"], d["roofline"]["stages_ms"])
PY
tail -12 $OUT/fuzz_1080p_timing.log
head -12 $OUT/kernel_stats.csv | cut -c1-200
