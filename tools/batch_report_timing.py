"""What the per-image results of a BATCH cost (jsnoop_batch_side_outputs / jsnoop_batch_log for every image behind one decode): python tools/batch_report_timing.py [N]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpegsnoop_amd as J
from oracle import harness as H
H.build(["oracle", "synth"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
files = [H.synth_jpeg(width=1920, height=1080, hs=2, vs=2, quality=85, seed=100 + i) for i in range(16)]
b = J.JpegBatch(want_planes=True)
if hasattr(b, 'enable_log'): b.enable_log()
for f in files: b.add_jpeg(f)
b.tile(n); b.upload(); b.decode(); b.sync()
t = time.perf_counter(); b.decode(); b.sync(); dec = (time.perf_counter() - t) * 1e3
t = time.perf_counter()
for i in range(n): b.side_outputs(i)
so = (time.perf_counter() - t) * 1e3
t = time.perf_counter()
for i in range(n): b.side_outputs(i)
so2 = (time.perf_counter() - t) * 1e3
t = time.perf_counter()
for i in range(n): b.side_outputs(i, bright=False)
so3 = (time.perf_counter() - t) * 1e3
out = {"images": n, "decode_ms": round(dec, 2), "side_outputs_ms_total": round(so, 1), "side_outputs_ms_per_image": round(so / n, 3), "again_ms_per_image": round(so2 / n, 3), "again_without_brightest_pixel": round(so3 / n, 3)}
if hasattr(b, "log_lines"):
    t = time.perf_counter()
    for i in range(n): b.log_lines(i)
    lg = (time.perf_counter() - t) * 1e3
    out["log_ms_per_image"] = round(lg / n, 3)
print(json.dumps(out))
