"""One large image through the single-image API vs the oracle (DIB + side outputs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import harness as H
import jpegsnoop_amd
H.build(["oracle", "synth"])
orc = H.oracle_backend(); gpu = H.Backend(jpegsnoop_amd.load(), "jsnoop_", "hip")
for kw in (dict(width=8192, height=8192, seed=1), dict(width=10001, height=3001, hs=2, vs=1, restart_interval=626, seed=2), dict(width=4096, height=4096, restart_interval=1, seed=3)):
    t = time.time(); data = H.synth_jpeg(**kw); t1 = time.time()
    H.drive(orc, data); t2 = time.time(); H.drive(gpu, data); t3 = time.time()
    ok = np.array_equal(orc.dib(), gpu.dib()) and np.array_equal(orc.mcu_map(), gpu.mcu_map()) and orc.status() == gpu.status() and np.array_equal(orc.dht_histo(), gpu.dht_histo())
    print(kw, "bytes", len(data), "synth %.1fs oracle %.1fs gpu %.2fs" % (t1 - t, t2 - t1, t3 - t2), "path", gpu.lib.jsnoop_last_path(gpu.h), "identical", ok)
