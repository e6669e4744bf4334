import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import harness as H
import fuzz_util as F
import jpegsnoop_amd
H.build(["oracle", "synth"])
n_cases, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
orc = H.oracle_backend(); gpu = H.Backend(jpegsnoop_amd.load(), "jsnoop_", "hip")
B = F.bases(H); shown = 0
for k in range(n_cases):
    bi = int(rng.integers(len(B)))
    data, q, mode = F.mutate(H, rng, B[bi])
    histo = int(rng.integers(2))
    for b in (orc, gpu): b.set_options(histo_en=histo)
    H.drive(orc, data, q); H.drive(gpu, data, q)
    r = F.differs(orc, gpu)
    if r and shown < 4:
        shown += 1
        a, b = orc.mcu_map().ravel(), gpu.mcu_map().ravel()
        d = np.nonzero(a != b)[0]
        p0 = H.parse_jpeg(B[bi])
        chg = [(key, p0.dht[key], q.dht[key]) for key in q.dht if q.dht[key] != p0.dht[key]]
        print("case", k, "base", bi, "path", gpu.lib.jsnoop_last_path(gpu.h), "flags", hex(gpu.lib.jsnoop_last_flags(gpu.h)), "diffs", len(d), "of", a.size, "first", d[:5])
        for i in d[:4]: print("   m", i, "orc", a[i] >> 4, a[i] & 15, "gpu", b[i] >> 4, b[i] & 15, "rst_int", q.rst_interval)
        for key, old, new in chg:
            oc, ov = old; nc, nv = new
            print("   table", key, "counts changed" if list(oc) != list(nc) else "", [(i, x, y) for i, (x, y) in enumerate(zip(ov, nv)) if x != y][:3], "len", len(ov), len(nv))
