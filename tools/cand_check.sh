#!/bin/bash
# candidate synchronisation (k_cand_*): config-2 probe with diagnostics, against the classic rounds (JSNOOP_CAND=0), then the parity tests
mkdir -p gpurun_out; rm -f gpurun_out/cand.log
for c in ${CANDS:-2 0}; do
  echo "== JSNOOP_CAND=$c" >> gpurun_out/cand.log
  JSNOOP_CAND=$c JSNOOP_DEBUG_CAND=1 timeout 300 python tools/config2_probe.py >> gpurun_out/cand.log 2>&1
done
bash tools/config2_profile.sh r03_config2_cand > gpurun_out/c2p.log 2>&1
python - >> gpurun_out/cand.log <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r03_config2_cand/kernel_stats.csv')):
    print(r['Name'].split('(')[0][:40].ljust(42), r['Calls'].rjust(4), '%9.1f us avg' % (float(r['AverageNs'])/1e3), 'min %8.1f max %8.1f' % (float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
if [ -z "$NOTESTS" ]; then timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_entropy_variants.py tests/test_large_golden.py -m gpu -x -q 2>&1 | tail -15 >> gpurun_out/cand.log; fi
tail -45 gpurun_out/cand.log
