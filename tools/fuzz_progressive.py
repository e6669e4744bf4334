"""Randomised transitive parity of the progressive path: random size / sampling / quality / restart interval / scan script;
the progressive file must decode to the pixels of the baseline file carrying the same coefficients (oracle).
usage: python tools/fuzz_progressive.py [n_cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import harness as H
import jpegsnoop_amd
H.build(["oracle", "synth"])
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
orc = H.oracle_backend(); gpu = H.Backend(jpegsnoop_amd.load(), "jsnoop_", "hip")
bad = 0
for k in range(n_cases):
    gray = int(rng.integers(6) == 0)
    hs, vs = (1, 1) if gray else [(1, 1), (2, 1), (1, 2), (2, 2)][int(rng.integers(4))]
    kw = dict(width=int(rng.integers(8, 700)), height=int(rng.integers(8, 500)), hs=hs, vs=vs, gray=gray,
              quality=int(rng.choice([10, 30, 50, 75, 85, 95, 100])), restart_interval=int(rng.choice([0, 0, 1, 2, 7, 33, 200])),
              seed=int(rng.integers(1 << 30)), optimize_huffman=int(rng.integers(2)))
    mode = int(rng.integers(1, 3))
    base = H.synth_jpeg(progressive=0, **kw); prog = H.synth_jpeg(progressive=mode, **kw)
    H.drive(orc, base)
    ns = gpu.decode_progressive(prog)
    if ns <= 0 or gpu.lib.jsnoop_last_flags(gpu.h):
        bad += 1; print("case", k, kw, "mode", mode, "decode failed", ns, gpu.lib.jsnoop_last_error()); continue
    a, b = orc.dib(), gpu.dib()
    Hh, W = kw["height"], kw["width"]
    ok = a.shape == b.shape and np.array_equal(a[a.shape[0] - Hh:, :W], b[b.shape[0] - Hh:, :W])
    if ok:
        for pa, pb in zip(orc.planes(), gpu.planes()):
            if pa is not None and not np.array_equal(pa[:Hh, :W], pb[:Hh, :W]): ok = False
    if not ok: bad += 1; print("case", k, kw, "mode", mode, "MISMATCH")
print("progressive cases", n_cases, "mismatches", bad)
