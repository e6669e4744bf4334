#!/bin/bash
# usage (on the GPU box): tools/pmc_kernel.sh <tag> <kernel regex> [variant.so ...]
#   Issue / stall counters of ONE kernel (256-image workload) for the tree's library and every variant: separate rocprofv3 --pmc passes, one counter
#   group per run (the gfx950 guide's rule), plus the kernel's durations without counters -> gpurun_out/<tag>/<lib>/summary.txt
TAG=$1; RE=$2; shift 2
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--images 256 --distinct 16 --steps 2 --warmup 1 --cpu-seconds 0 --no-extras --no-split"
run_lib() {
  local NAME=$1 i=0
  mkdir -p $OUT/$NAME
  for GROUP in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
               "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
               "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH SQ_LDS_BANK_CONFLICT" \
               "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_IFETCH_LEVEL SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"; do
    i=$((i+1))
    timeout 240 rocprofv3 --pmc $GROUP --kernel-trace --kernel-include-regex "$RE" --output-format csv -d $OUT/$NAME/p$i -o p$i -- python $ROOT/bench.py $ARGS > $OUT/$NAME/p$i.log 2>&1
  done
  python $ROOT/tools/pmc_summarize.py $OUT/$NAME > $OUT/$NAME/summary.txt 2>&1
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$NAME/st -o st -- python $ROOT/bench.py $ARGS > $OUT/$NAME/st.log 2>&1
  find $OUT/$NAME/st -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/$NAME/kernel_stats.csv
  rm -rf $OUT/$NAME/st $OUT/$NAME/p?/*/*_agent_info.csv
}
run_lib tree
for SO in "$@"; do
  N=$(basename $SO .so)
  cp $ROOT/jpegsnoop_amd/libjsnoop_gpu.so /tmp/libjsnoop_gpu.orig.so; cp $ROOT/$SO $ROOT/jpegsnoop_amd/libjsnoop_gpu.so
  run_lib $N
  cp /tmp/libjsnoop_gpu.orig.so $ROOT/jpegsnoop_amd/libjsnoop_gpu.so
done
for d in $OUT/*/; do echo "== $d"; grep -A40 "$RE" $d/summary.txt | head -45; grep "$RE" $d/kernel_stats.csv | cut -c1-40,200-330; done
