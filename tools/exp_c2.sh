#!/bin/bash
# usage: tools/exp_c2.sh VAR=a VAR=b ... : config-2 probe (one 3840x2160 image) under each environment setting, one line each
mkdir -p gpurun_out; rm -f gpurun_out/exp_c2.log
for e in "$@"; do echo "== $e" >> gpurun_out/exp_c2.log; env $e timeout 300 python tools/config2_probe.py >> gpurun_out/exp_c2.log 2>&1; done
cat gpurun_out/exp_c2.log
