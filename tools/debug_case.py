import sys, os
sys.path.insert(0, '.')
import numpy as np
from oracle import harness as H
import jpegsnoop_amd as J
H.build(["oracle", "synth"])
o = H.oracle_backend()
g = H.Backend(J.load(), "jsnoop_", "hip")
data = H.synth_jpeg(width=160, height=96, seed=4)
p = H.parse_jpeg(data)
for samp in ([(4, 1), (1, 1), (1, 1)], [(1, 4), (1, 1), (1, 1)], [(4, 2), (2, 1), (1, 2)], [(2, 2), (2, 1), (1, 1)],
             [(1, 1), (2, 2), (2, 2)], [(3, 1), (1, 1), (1, 1)], [(2, 2), (2, 2), (2, 2)], [(4, 4), (1, 1), (2, 2)]):
    q = H.parse_jpeg(data)
    q.comps = [(c[0], h, v, c[3]) for c, (h, v) in zip(p.comps, samp)]
    H.drive(o, data, q); H.drive(g, data, q)
    path, fl = g.lib.jsnoop_last_path(g.h), g.lib.jsnoop_last_flags(g.h)
    do, dg = o.dib(), g.dib()
    po, pg = o.planes(), g.planes()
    pe = [bool(np.array_equal(a, b)) for a, b in zip(po, pg)]
    bad = np.argwhere((do != dg).any(axis=2))
    print(samp, 'path', path, 'flags %#x' % fl, 'dib eq', np.array_equal(do, dg), 'planes eq', pe, 'geom', g.geometry(), 'status', o.status() == g.status(),
          'first bad px', bad[:3].tolist(), 'nbad', len(bad))
    if not pe[0]:
        bp = np.argwhere(po[0] != pg[0]); print('   Y plane bad', bp[:5].tolist(), len(bp), po[0][tuple(bp[0])], pg[0][tuple(bp[0])])
