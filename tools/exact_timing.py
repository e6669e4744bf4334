"""Times the sequential exact-mirror kernel (the path malformed streams take) on well-formed images, forced."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import harness as H
import jpegsnoop_amd as J
H.build(["oracle", "synth"])
for n, kw in ((1, dict(width=1920, height=1080, seed=1)), (64, dict(width=1920, height=1080, seed=2)), (1, dict(width=3840, height=2160, seed=3))):
    b = J.JpegBatch(force_exact=True)
    for i in range(n):
        b.add_jpeg(H.synth_jpeg(**{**kw, "seed": kw["seed"] + i}))
    b.upload(); b.decode(); b.sync()
    t = time.perf_counter(); b.decode(); b.sync(); dt = time.perf_counter() - t
    print(n, "x", kw["width"], "x", kw["height"], "exact path: %.1f ms" % (dt * 1e3))
    b.close()
