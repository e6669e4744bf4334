#!/bin/bash
# usage (on the GPU box): tools/fuzz_campaign.sh <tag> [seeds...]  -- the parity sweeps of tools/ with several seeds in one call: fuzz_gpu.py (every output of the single-image API
# against the oracle), fuzz_damaged_log.py (log text against the compiled reference), fuzz_batch.py, fuzz_progressive.py -> gpurun_out/<tag>/campaign.txt
TAG=${1:-campaign}; shift
SEEDS=${@:-201 202 203 204}
OUT=gpurun_out/$TAG; mkdir -p $OUT
: > $OUT/campaign.txt
for S in $SEEDS; do
  timeout 600 python tools/fuzz_gpu.py 5000 $S > $OUT/fuzz_gpu_$S.log 2>&1; echo "fuzz_gpu seed $S: $(tail -1 $OUT/fuzz_gpu_$S.log)" >> $OUT/campaign.txt; grep MISMATCH $OUT/fuzz_gpu_$S.log | head -5 >> $OUT/campaign.txt
  timeout 600 python tools/fuzz_damaged_log.py 3000 $S > $OUT/fuzz_log_$S.log 2>&1; echo "fuzz_damaged_log seed $S: $(tail -1 $OUT/fuzz_log_$S.log)" >> $OUT/campaign.txt; grep -A2 "MISMATCH\|LOG differs\|exception" $OUT/fuzz_log_$S.log | head -12 >> $OUT/campaign.txt
  timeout 600 python tools/fuzz_damaged_log.py 300 $S 1 > $OUT/fuzz_logbig_$S.log 2>&1; echo "fuzz_damaged_log 1080p seed $S: $(tail -1 $OUT/fuzz_logbig_$S.log)" >> $OUT/campaign.txt; grep -A2 "MISMATCH\|LOG differs\|exception" $OUT/fuzz_logbig_$S.log | head -12 >> $OUT/campaign.txt
done
timeout 900 python tools/fuzz_batch.py 301 > $OUT/fuzz_batch.log 2>&1; echo "fuzz_batch: $(grep "^batch" $OUT/fuzz_batch.log | tr "\n" " ")" >> $OUT/campaign.txt; grep "^side round\|^log round" $OUT/fuzz_batch.log | head -6 >> $OUT/campaign.txt
timeout 600 python tools/fuzz_progressive.py > $OUT/fuzz_progressive.log 2>&1; echo "fuzz_progressive: $(tail -1 $OUT/fuzz_progressive.log)" >> $OUT/campaign.txt
cat $OUT/campaign.txt
