"""Probe: a single-image progressive decode before and after GetBitmapPtr on the same decoder (a read-back into pageable memory used to
make every later call of the process slower: 2.8 -> 5.1 ms), beside other steps of bench.extras_single_gpu."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import harness as H
import jpegsnoop_amd as J
H.build(["oracle", "synth"])
kw5 = dict(width=1920, height=1080, hs=2, vs=1, restart_interval=120, quality=85, seed=55)
base5, prog5 = H.synth_jpeg(progressive=0, **kw5), H.synth_jpeg(progressive=2, **kw5)
def loop(tag, dec=None):
    own = dec is None
    if own: dec = J.CimgDecode(); dec.DecodeProgressive(prog5)
    t = time.perf_counter()
    for _ in range(10): dec.DecodeProgressive(prog5)
    print(tag, "ms per call %.3f" % ((time.perf_counter() - t) * 100))
    if own: dec.close()
loop("fresh decoder")
orc = H.oracle_backend()
loop("after creating the oracle backend")
H.drive(orc, base5)
loop("after an oracle decode")
dec = J.CimgDecode(); dec.DecodeProgressive(prog5)
ok = np.array_equal(dec.GetBitmapPtr(), orc.dib())
loop("same decoder after GetBitmapPtr (%s)" % ok, dec)
dec.close()
one = J.JpegBatch(); one.add_jpeg(H.synth_jpeg(width=3840, height=2160, hs=2, vs=2, quality=85, seed=77)); one.upload(); one.decode(); one.sync(); one.decode_timed(10); one.close()
loop("after a 4K batch object")
