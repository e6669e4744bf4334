#!/bin/bash
# per-image stage times of the bench workload against the batch size (does the per-image cost depend on how much is resident / how long the chip is loaded?)
for n in 128 256 512 1024; do
  python bench.py --images $n --distinct 64 --steps 20 --warmup 3 --cpu-seconds 0 --no-extras --no-split 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); n=$n; st=d['roofline']['stages_ms']
print(n, 'ms/step', d['ms_per_step'], 'us/img', round(d['ms_per_step']*1e3/n,2), {k: round(v*1e3/n,2) for k,v in st.items() if v*1e3/n > 0.05})"
done
python bench.py --images 1024 --steps 2 --warmup 0 --cpu-seconds 0 --no-extras --no-split 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('1024 cold, 2 steps: ms/step', d['ms_per_step'])"
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -4
