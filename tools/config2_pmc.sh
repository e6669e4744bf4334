#!/bin/bash
# usage (GPU box): tools/config2_pmc.sh <tag>   two PMC passes of the config-2 probe (one 3840x2160 image, 20 decodes) -> gpurun_out/<tag>/summary.txt
TAG=${1:-r03_config2_pmc}; ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/p1 -o p1 -- python $ROOT/tools/config2_probe.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/p2 -o p2 -- python $ROOT/tools/config2_probe.py > $OUT/p2.log 2>&1
python $ROOT/tools/pmc_summarize.py $OUT > $OUT/summary.txt 2>&1
grep -A17 "^k_cand_chain\|^k_cand_spec\|^k_cand_walk\|^k_write2<4>" $OUT/summary.txt | head -90
