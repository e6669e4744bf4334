#!/bin/bash
# usage (on the GPU box, via gpurun): tools/profile_round.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of the default bench command (short run) -> gpurun_out/<tag>/stats/*, bench line
#   2. the PMC passes of tools/pmc_collect.sh on the 256-image workload      -> gpurun_out/<tag>/pmc/summary.txt
set -u
TAG=${1:-rXX}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT/stats
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- python $ROOT/bench.py --steps 10 --warmup 2 --cpu-seconds 0 --no-extras --no-split > $OUT/bench_line.json 2> $OUT/bench_stderr.log
tail -1 $OUT/bench_line.json > $OUT/bench.json
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
cd $ROOT && bash tools/pmc_collect.sh $TAG/pmc > /dev/null 2>&1
cp $OUT/pmc/summary.txt $OUT/pmc_summary_256img.txt
head -20 $OUT/kernel_stats.csv
