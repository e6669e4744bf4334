#!/bin/bash
# usage (GPU box): tools/config2_profile.sh <tag>   kernel stats of BASELINE config 2 (one 3840x2160 4:2:0 image, 20 decodes) -> gpurun_out/<tag>/
TAG=${1:-r03_config2}; ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- python $ROOT/tools/config2_probe.py > $OUT/probe.json 2> $OUT/stderr.log
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -rf $OUT/stats
tail -1 $OUT/probe.json; head -14 $OUT/kernel_stats.csv
