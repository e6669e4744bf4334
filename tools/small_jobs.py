"""Small jobs through the batch path: N distinct 1920x1080 4:2:0 images resident in HBM, ms per decode -- `ms`: 20 back-to-back decodes, wall clock around a device
synchronise (the headline's way of timing); `ms_with_stage_events` and the stage times: decode_timed, a hipEvent behind every stage (5-10 % on jobs this small).  usage: python tools/small_jobs.py [N ...]   (env: JSNOOP_CAND, JSNOOP_SUB_WL)"""
import os, sys, json, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpegsnoop_amd as J
from oracle import harness as H
ns = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 32, 64]
files = [H.synth_jpeg(width=1920, height=1080, hs=2, vs=2, quality=85, seed=100 + i) for i in range(max(ns))]
out = {}
for n in ns:
    b = J.JpegBatch()
    for f in files[:n]:
        b.add_jpeg(f)
    b.upload(); b.decode(); b.sync()
    mst, st = b.decode_timed(10)
    b.decode(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        b.decode()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3 / 20
    out[n] = {"ms": round(ms, 4), "ms_per_image": round(ms / n, 4), "ms_with_stage_events": round(mst, 4), "sync": round(st["sync"], 4), "write": round(st["write"], 4), "idct_color": round(st["idct_color"], 4)}
    b.close()
print(json.dumps({"env": {k: os.environ[k] for k in ("JSNOOP_CAND", "JSNOOP_SUB_WL") if k in os.environ}, "jobs": out}))
