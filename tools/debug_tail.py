"""Debug helper: golden cases + fuzz cases through the single-image API; on a DIB mismatch prints flags, first anomalous block and the
first MCU whose pixels differ.   usage: python tools/debug_tail.py [n_fuzz] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import harness as H
import fuzz_util as F
from golden_util import load_case, manifest
import jpegsnoop_amd as J
H.build(["oracle", "synth"])
orc = H.oracle_backend(); gpu = H.Backend(J.load(), "jsnoop_", "hip")
def check(tag, data, q=None):
    H.drive(orc, data, q); H.drive(gpu, data, q)
    a, b = orc.dib(), gpu.dib()
    if a is None or b is None or a.shape != b.shape: 
        if (a is None) != (b is None): print(tag, "preview differs")
        return
    if np.array_equal(a, b): return
    g = gpu.geometry(); mw, mh, mxm = g[0], g[1], g[2]
    d = np.any(a != b, axis=2)[::-1]            # top-down
    ys, xs = np.nonzero(d)
    mcus = sorted(set((y // mh) * mxm + x // mw for y, x in zip(ys, xs)))
    print(tag, "path", gpu.lib.jsnoop_last_path(gpu.h), "flags 0x%04x" % gpu.lib.jsnoop_last_flags(gpu.h), "geom", g[:4], "first bad MCUs", mcus[:6], "n bad", len(mcus), "of", g[2] * g[3])
M = manifest()
for name in sorted(M["cases"]):
    check("golden " + name, load_case(name))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 31)
B = F.bases(H)
for k in range(n):
    data, q, mode = F.mutate(H, rng, B[int(rng.integers(len(B)))])
    histo = int(rng.integers(2)); ac = int(rng.integers(4) != 0); em = int(rng.choice([20, 20, 3, 1]))
    for b in (orc, gpu): b.set_options(histo_en=histo, decode_ac=ac, err_max=em)
    try: check("fuzz %d mode %d ac %d" % (k, mode, ac), data, q)
    except Exception as ex: print("case", k, "exception", ex)
