#!/bin/bash
# usage (GPU box): tools/final_evidence.sh <part>   the round's profiles in two gpurun calls
#   a: config 3 kernel stats + PMC (profile_round.sh r03_b)      b: config 2 / small jobs / progressive kernel stats, default bench line
PART=${1:-a}
if [ "$PART" == "a" ]; then
  bash tools/profile_round.sh r03_b > gpurun_out/final_a.log 2>&1
  tail -5 gpurun_out/final_a.log
else
  bash tools/config2_profile.sh r03_config2_final > gpurun_out/final_b.log 2>&1
  bash tools/exp_prof_cmd.sh tools/prog_batch_run.py 512 3 >> gpurun_out/final_b.log 2>&1
  cp gpurun_out/exp_prof_cmd/kernel_stats.csv gpurun_out/r03_prog_batch512_kernel_stats.csv; cp gpurun_out/exp_prof_cmd/stdout.log gpurun_out/r03_prog_batch512_run.txt
  bash tools/exp_prof_cmd.sh tools/prog_timing.py >> gpurun_out/final_b.log 2>&1
  cp gpurun_out/exp_prof_cmd/kernel_stats.csv gpurun_out/r03_config5_single_kernel_stats.csv
  python tools/small_jobs.py 1 2 4 8 16 32 48 > gpurun_out/r03_small_jobs.json 2>> gpurun_out/final_b.log
  JSNOOP_CAND=0 python tools/small_jobs.py 1 2 4 8 16 32 48 64 >> gpurun_out/r03_small_jobs.json 2>> gpurun_out/final_b.log
  python tools/call_latency.py > gpurun_out/r03_call_latency.json 2>> gpurun_out/final_b.log
  python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
  tail -c 600 gpurun_out/bench_final.json
fi
