"""Damaged 1080p files: how long does decode + sync of a resident batch of one take, case by case?  Mutations of tests/fuzz_util.py (scan
bytes flipped / truncated / markers and garbage inserted / bytes deleted) on 1080p 4:2:0 files with and without restart markers; every
flagged case and every twentieth clean one is compared with the oracle (DIB).   usage: python tools/fuzz_1080p_timing.py [cases] [seed]"""
import collections, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import harness as H
import fuzz_util as F
import jpegsnoop_amd as J
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bases = [H.synth_jpeg(width=1920, height=1080, seed=61), H.synth_jpeg(width=1920, height=1080, seed=62, restart_interval=120),
         H.synth_jpeg(width=1920, height=1080, seed=63, restart_interval=8), H.synth_jpeg(width=1920, height=1080, seed=64, noise_sigma=3)]
orc = H.oracle_backend()
b = J.JpegBatch()
times = []; by = collections.defaultdict(list); bad = 0; checked = 0; worst = (0, None)
for k in range(n):
    base = bases[k % len(bases)]
    while True:
        data, q, mode = F.mutate(H, rng, base)
        if mode <= 5: break                                     # scan damage only (header mutations change the geometry, not the path)
    b.clear()
    try:
        b.add_jpeg(data)
    except RuntimeError:
        continue
    b.upload()
    t = time.perf_counter(); b.decode(); b.sync(); ms = (time.perf_counter() - t) * 1e3
    inf = b.info(0); key = (inf["path"], "0x%04x" % inf["flags"])
    times.append(ms); by[key].append(ms)
    if ms > worst[0]: worst = (ms, (k, mode, key))
    if inf["flags"] or k % 20 == 0:
        H.drive(orc, data); checked += 1
        if int(b.dib_checksums()[0]) != J.dib_checksum_numpy(orc.dib()):
            bad += 1; print("MISMATCH case", k, "mode", mode, key)
b.close()
times = np.array(times)
out = {"cases": len(times), "oracle_checked": checked, "mismatches": bad, "ms_median": round(float(np.median(times)), 3), "ms_p99": round(float(np.percentile(times, 99)), 3),
       "ms_max": round(float(times.max()), 3), "worst": str(worst[1]), "over_50ms": int((times > 50).sum()),
       "classes": {"path %d flags %s" % k2: {"n": len(v), "median_ms": round(float(np.median(v)), 3), "max_ms": round(float(np.max(v)), 3)} for k2, v in sorted(by.items(), key=lambda kv: -len(kv[1]))}}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "fuzz_1080p_timing.json"), "w"), indent=1)
print({k: v for k, v in out.items() if k != "classes"})
for k2, v in sorted(out["classes"].items(), key=lambda kv: -kv[1]["max_ms"]):
    print("   %-26s n %4d  median %9.3f ms  max %9.3f ms" % (k2, v["n"], v["median_ms"], v["max_ms"]))
