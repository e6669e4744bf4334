import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, itertools
from oracle import harness as H
import jpegsnoop_amd
H.build(["oracle", "synth"])
orc = H.oracle_backend(); gpu = H.Backend(jpegsnoop_amd.load(), "jsnoop_", "hip")
data = H.synth_jpeg(width=160, height=96, seed=4)
p = H.parse_jpeg(data)
bad = []
rng = np.random.default_rng(3)
combos = [tuple(int(x) for x in rng.integers(1, 5, 6)) for _ in range(150)]
for cb in combos:
    samp = [(cb[0], cb[1]), (cb[2], cb[3]), (cb[4], cb[5])]
    q = H.parse_jpeg(data); q.comps = [(c[0], h, v, c[3]) for c, (h, v) in zip(p.comps, samp)]
    H.drive(orc, data, q); H.drive(gpu, data, q)
    a, b = orc.dib(), gpu.dib()
    if (a is None) != (b is None) or (a is not None and not np.array_equal(a, b)):
        pl = [np.array_equal(x, y) for x, y in zip(orc.planes(), gpu.planes()) if x is not None]
        n = int((a != b).any(axis=2).sum()) if a is not None and b is not None and a.shape == b.shape else -1
        bad.append((samp, pl, n, gpu.lib.jsnoop_last_path(gpu.h)))
print(len(bad), "of", len(combos))
for x in bad[:25]: print(x)
