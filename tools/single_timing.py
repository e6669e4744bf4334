import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import harness as H
import jpegsnoop_amd as J
H.build(["oracle", "synth"])
f4k = H.synth_jpeg(width=3840, height=2160, hs=2, vs=2, quality=85, seed=77)
one = J.JpegBatch(); one.add_jpeg(f4k); one.upload(); one.decode(); one.sync()
ms, st = one.decode_timed(10)
print(os.environ.get("JSNOOP_SUB_WL"), round(ms,3), {k: round(v,3) for k,v in st.items()}, one.info(0)["path"], one.info(0)["flags"])
