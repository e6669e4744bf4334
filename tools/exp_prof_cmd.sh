#!/bin/bash
# usage (GPU box): tools/exp_prof_cmd.sh <python script and args...>   kernel stats (rocprofv3 --kernel-trace --stats) of a command, averages > 3 us
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/exp_prof_cmd; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( cd $ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- python "$@" > $OUT/stdout.log 2> $OUT/stderr.log )
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
rm -rf $OUT/stats
tail -2 $OUT/stdout.log
python - <<PY
import csv
for r in csv.DictReader(open('$OUT/kernel_stats.csv')):
    if float(r['AverageNs']) > 3000: print(' ', r['Name'].split('(')[0][:40].ljust(42), r['Calls'].rjust(4), '%9.1f us avg  min %8.1f max %8.1f' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
