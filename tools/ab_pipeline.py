"""A/B of the staging pipeline's overlap (tests/test_gpu_batch_large.py::test_staging_pipeline_overlaps_and_stays_exact) between two builds
of the library.  usage: python tools/ab_pipeline.py [path of the .so]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpegsnoop_amd.capi as capi
if len(sys.argv) > 1:
    capi.LIB_PATH = os.path.abspath(sys.argv[1])
import jpegsnoop_amd as J
from oracle import harness as H
files = [H.synth_jpeg(width=1280, height=720, seed=300 + i) for i in range(4)]
pipe = J.JpegPipeline(2)
for b in pipe.slots:
    for f in files:
        b.add_jpeg(f)
    b.tile(128)
out = []
for rep in range(4):
    r2 = pipe.run(6, d2h=False); r3 = pipe.run(3, d2h=True)
    out.append({"T2": round(r2["ms_per_batch"], 3), "h2d": round(r2["h2d_ms"], 3), "dec": round(r2["decode_ms"], 3), "T3": round(r3["ms_per_batch"], 3), "d2h": round(r3["d2h_ms"], 3),
                "T3_over_sum": round(r3["ms_per_batch"] / (r3["h2d_ms"] + r3["decode_ms"] + r3["d2h_ms"]), 3)})
pipe.close()
print(json.dumps({"lib": capi.LIB_PATH.split("/")[-1], "runs": out}))
