#!/bin/bash
# usage (on the GPU box): tools/pmc_progressive.sh <tag> [n_files]  -- kernel stats and two PMC passes of a batch of progressive files (tools/prog_batch.py)
TAG=${1:-prog_pmc}; N=${2:-64}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o ks -- python $GRAFT_REPO_ROOT/tools/prog_batch.py $N 3 > $OUT/ks.log 2>&1
i=0
for GROUP in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- python $GRAFT_REPO_ROOT/tools/prog_batch.py $N 1 > $OUT/p$i.log 2>&1
done
python $GRAFT_REPO_ROOT/tools/pmc_summarize.py $OUT > $OUT/summary.txt 2>&1
find $OUT/ks -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
grep -A22 "k_prog" $OUT/summary.txt | head -80
