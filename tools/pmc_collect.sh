#!/bin/bash
# Collects rocprofv3 PMC passes for the bench workload (one counter group per run, as the gfx950 guide prescribes).
# usage: tools/pmc_collect.sh <tag> [bench args...]
set -u
TAG=${1:-pmc}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS=${@:---images 256 --distinct 16 --steps 2 --warmup 1 --cpu-seconds 0 --no-extras --no-split}
GROUPS_MAX=${PMC_GROUPS:-6}
i=0
for GROUP in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_PERF_SEL_TOTAL_HIT_LRU_READ TCP_PERF_SEL_TOTAL_MISS_LRU_READ TCP_PERF_SEL_TOTAL_READ"; do
  i=$((i+1)); [ $i -gt $GROUPS_MAX ] && break
  rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/p$i.log 2>&1
done
python $GRAFT_REPO_ROOT/tools/pmc_summarize.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
