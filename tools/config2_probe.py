"""One 3840x2160 4:2:0 image through the batch path (BASELINE config 2): ms per decode and stage split for the sub-sequence
length given by JSNOOP_SUB_WL (default: the library's own choice).  usage: python tools/config2_probe.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpegsnoop_amd as J
from oracle import harness as H
one = J.JpegBatch()
one.add_jpeg(H.synth_jpeg(width=3840, height=2160, hs=2, vs=2, quality=85, seed=77)); one.upload(); one.decode(); one.sync()
ms, st = one.decode_timed(20)
print(json.dumps({"wl": os.environ.get("JSNOOP_SUB_WL", "auto"), "ms": round(ms, 4), "stages_ms": {k: round(v, 4) for k, v in st.items()}}))
one.close()
