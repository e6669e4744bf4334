#!/bin/bash
# usage (GPU box): tools/ab_env_sync.sh <tag>  -- GPU suite, then the quick bench with the list-round synchronisation (default) and with two plain k_sync launches
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT
if [ -z "$AB_SKIP_TESTS" ]; then timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log; fi
for V in default classic default classic; do
  if [ $V = classic ]; then export JSNOOP_SYNC_LAUNCHES=2; else unset JSNOOP_SYNC_LAUNCHES; fi
  python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-extras $AB_ARGS 2>>$OUT/$V.err | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); s = d.get('one_stream', {})
print('$V', d['value'], d['bit_exact'], d['ms_per_step'], 'one-stream:', s.get('ms_per_step'), s.get('bit_exact'), d['roofline']['stages_ms'])"
done
