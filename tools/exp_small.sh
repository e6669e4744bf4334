#!/bin/bash
# usage: tools/exp_small.sh "VAR=a VAR2=b" "VAR=c" ... : tools/small_jobs.py under each environment (quote a group of settings)
mkdir -p gpurun_out; rm -f gpurun_out/exp_small.log
for e in "$@"; do env $e timeout 600 python tools/small_jobs.py $NS >> gpurun_out/exp_small.log 2>&1; done
cat gpurun_out/exp_small.log
