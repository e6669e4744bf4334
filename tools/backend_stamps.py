"""Reads back the phase stamps of a back-end STAMP build (tools/variants/r06_stamps.patch): N x 1080p 4:2:0 q85 (the bench pictures) through the
batch path with the stamp library copied over jpegsnoop_amd/libjsnoop_gpu.so (tools/ab_round.sh style); every wave of k_idct_color leaves the sums of
its s_memtime deltas in the first sixteen pixels of the bottom row of its first MCU.
usage: python tools/backend_stamps.py [N=256] [distinct=16]   -> one JSON line (cycles per MCU and wave, averaged over all waves)"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpegsnoop_amd as J
from oracle import harness as H
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
distinct = int(sys.argv[2]) if len(sys.argv) > 2 else 16
files = [H.synth_jpeg(width=1920, height=1080, hs=2, vs=2, quality=85, seed=100 + i) for i in range(distinct)]
b = J.JpegBatch()
for i in range(n):
    b.add_jpeg(files[i % distinct])
b.set_tuning(split=1)
b.upload(); b.decode(); b.sync()
ms, st = b.decode_timed(3)
total_mcus = n * 120 * 68
mpw = min(64, max(1, (total_mcus + 8191) // 8192))
nmcu, xmax = 120 * 68, 120
wgs = max(1, (nmcu + 8 * mpw - 1) // (8 * mpw)); per = (nmcu + wgs - 1) // wgs
rec = []
for i in range(0, n, max(1, n // 32)):
    d = b.dib(i).view(np.uint32).reshape(1088, 1920)
    for wg in range(wgs):
        for w in range(8):
            m = wg * per + w
            if m >= min(wg * per + per, nmcu): continue
            mx, my = m % xmax, m // xmax
            rec.append(d[1088 - (my + 1) * 16, mx * 16: mx * 16 + 16].astype(np.int64))
rec = np.array(rec)
mcus = rec[:, 12].sum()
names = ["p0_build", "p0_terms", "p0_tile", "p1_build", "p1_terms", "p1_tile", "p2_build", "p2_terms", "p2_tile", "colour_store"]
out = {"images": n, "idct_color_ms": round(st["idct_color"], 4), "waves_read": int(len(rec)), "mcus_per_wave": float(rec[:, 12].mean()),
       "cycles_per_mcu": {k: round(float(rec[:, j].sum() / mcus), 1) for j, k in enumerate(names)},
       "steps_per_mcu": round(float(rec[:, 10].sum() / mcus), 2), "rounds_per_mcu": round(float(rec[:, 11].sum() / mcus), 2),
       "loop_cycles_per_mcu": round(float(rec[:, 13].sum() / mcus), 1)}
out["sum_of_phases"] = round(sum(out["cycles_per_mcu"].values()), 1)
print(json.dumps(out))
