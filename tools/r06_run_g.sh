cp jpegsnoop_amd/libjsnoop_gpu.so /tmp/orig.so; cp gpurun_variants/lib_r06_syncstat.so jpegsnoop_amd/libjsnoop_gpu.so
python bench.py --images 128 --distinct 16 --steps 1 --warmup 0 --cpu-seconds 0 --no-extras --no-split 2>/dev/null | grep SYNCSTAT > gpurun_out/r06_syncstat.txt
cp /tmp/orig.so jpegsnoop_amd/libjsnoop_gpu.so
wc -l gpurun_out/r06_syncstat.txt; python - <<'PY'
import numpy as np
rows=[l.split() for l in open('gpurun_out/r06_syncstat.txt')]
a=np.array([[int(x) for x in r[1:5]]+[int(x) for x in r[6:14]] for r in rows if len(r)>=14])
print("workgroups", len(a), "rounds mean %.2f max %d" % (a[:,1].mean(), a[:,1].max()), "lane-walks per wg %.1f" % a[:,2].mean(), "wave-rounds per wg %.2f" % a[:,3].mean())
print("active lanes by round (mean):", np.round(a[:,4:].mean(axis=0),1).tolist())
import collections
print("rounds histogram:", sorted(collections.Counter(a[:,1].tolist()).items()))
PY
