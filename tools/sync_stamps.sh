cp jpegsnoop_amd/libjsnoop_gpu.so /tmp/orig.so; cp gpurun_variants/lib_r06_syncstamp.so jpegsnoop_amd/libjsnoop_gpu.so
python bench.py --images 128 --distinct 16 --steps 1 --warmup 0 --cpu-seconds 0 --no-extras --no-split > gpurun_out/syncstamp.log 2>&1
cp /tmp/orig.so jpegsnoop_amd/libjsnoop_gpu.so
grep -c STAMP gpurun_out/syncstamp.log
python - <<'PY'
import re
T=V=L=N=0; n=0
for l in open('gpurun_out/syncstamp.log'):
    if l.startswith('STAMP'):
        a=l.split(); T+=int(a[2]); V+=int(a[3]); L+=int(a[4]); N+=int(a[5]); n+=1
print("walks", n, "cycles/walk", T/max(n,1), "vm wait frac", V/max(T,1), "lds wait frac", L/max(T,1), "refill events per walk", N/max(n,1), "vm wait per event", V/max(N,1))
PY
