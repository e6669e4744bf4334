#!/bin/bash
# usage: tools/pc_sample.sh <tag> [method: host_trap|stochastic]  -- rocprofv3 PC sampling of the bench workload (256 images), raw csv under gpurun_out/<tag>
TAG=${1:-pcs}; METHOD=${2:-host_trap}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$METHOD" = stochastic ]; then UNIT=cycles; INT=1048576; else UNIT=time; INT=100; fi
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $UNIT --pc-sampling-method $METHOD --pc-sampling-interval $INT --kernel-trace --output-format csv -d $OUT -o pcs \
   -- python $GRAFT_REPO_ROOT/bench.py --images 256 --distinct 16 --steps 3 --warmup 1 --cpu-seconds 0 --no-extras > $OUT/run.log 2>&1
echo rc=$? >> $OUT/run.log
ls -la $OUT | head -20
tail -5 $OUT/run.log
