#!/bin/bash
# usage (GPU box): tools/prog_profile.sh <tag>   kernel stats + per-dispatch trace + one PMC pass of the config-5 progressive batch
TAG=${1:-r03_prog}; ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/tools/prog_batch_run.py 64 5 > $OUT/run.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- python $ROOT/tools/prog_batch_run.py 64 5 > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT/stats -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_trace.csv
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/pmc1 -o p -- python $ROOT/tools/prog_batch_run.py 64 1 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/pmc2 -o p -- python $ROOT/tools/prog_batch_run.py 64 1 > $OUT/pmc2.log 2>&1
find $OUT/pmc1 -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $OUT/pmc1.csv
find $OUT/pmc2 -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $OUT/pmc2.csv
rm -rf $OUT/stats $OUT/pmc1 $OUT/pmc2
cat $OUT/run.txt; head -12 $OUT/kernel_stats.csv
