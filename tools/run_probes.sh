#!/bin/bash
# builds and runs the hardware probes under tools/probes on the GPU box; output under gpurun_out/
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for p in pk_f32_rate idct_terms2; do
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/$p tools/probes/$p.hip 2> gpurun_out/$p.build.log || { echo "build of $p failed"; tail -5 gpurun_out/$p.build.log; continue; }
done
timeout 300 /tmp/pk_f32_rate > gpurun_out/pk_f32_rate.txt 2>&1; echo "pk_f32_rate rc=$?"; cat gpurun_out/pk_f32_rate.txt
( timeout 60 /tmp/idct_terms2 16 0; timeout 60 /tmp/idct_terms2 16 8 ) > gpurun_out/idct_terms2.txt 2>&1; echo "idct_terms2 rc=$?"; cat gpurun_out/idct_terms2.txt
