#!/bin/bash
# usage: tools/exp_c2prof.sh VAR=a ... : kernel stats of the config-2 probe under each environment setting (the k_cand_* / k_sync lines)
mkdir -p gpurun_out; rm -f gpurun_out/exp_c2prof.log
for e in "$@"; do
  echo "== $e" >> gpurun_out/exp_c2prof.log
  env $e bash tools/config2_profile.sh exp_prof > /dev/null 2>&1
  python - >> gpurun_out/exp_c2prof.log <<'PY'
import csv, json
print(open('gpurun_out/exp_prof/probe.json').read().strip().splitlines()[-1])
for r in csv.DictReader(open('gpurun_out/exp_prof/kernel_stats.csv')):
    if float(r['AverageNs']) > 3000: print(' ', r['Name'].split('(')[0][:40].ljust(42), r['Calls'].rjust(4), '%9.1f us avg' % (float(r['AverageNs'])/1e3))
PY
done
cat gpurun_out/exp_c2prof.log
