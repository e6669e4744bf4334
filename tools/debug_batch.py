import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["JSNOOP_SUB_WL"] = sys.argv[1] if len(sys.argv) > 1 else "5"
from oracle import harness as H
import jpegsnoop_amd as J
H.build(["oracle", "synth"])
kws = [dict(width=320, height=240), dict(width=333, height=217, hs=1, vs=1), dict(width=160, height=120, gray=1),
       dict(width=640, height=360, hs=2, vs=1, restart_interval=40), dict(width=1280, height=720, quality=92)]
files = [H.synth_jpeg(seed=20 + i, **kw) for i, kw in enumerate(kws)]
b = J.JpegBatch(want_planes=True)
for f in files: b.add_jpeg(f)
b.upload(); b.decode(); b.sync()
for i in range(5): print(i, kws[i], "path", b.info(i)["path"], "flags", hex(b.info(i)["flags"]))
