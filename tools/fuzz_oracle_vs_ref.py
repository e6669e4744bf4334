"""CPU only (needs /root/reference: the compiled reference in oracle/_ref): oracle vs compiled reference on mutated scans and
headers with random options -- the long version of tests/test_oracle_vs_ref.py.   usage: python tools/fuzz_oracle_vs_ref.py [n] [seed]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle import harness as H
import fuzz_util as F
H.build(["oracle", "synth", "ref"])
orc = H.oracle_backend(); ref = H.ref_backend()
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B = F.bases(H); bad = 0; t = time.time()
for k in range(n):
    data, q, mode = F.mutate(H, rng, B[int(rng.integers(len(B)))])
    histo = int(rng.integers(2)); ac = int(rng.integers(4) != 0); em = int(rng.choice([20, 20, 3, 1]))
    for b in (orc, ref): b.set_options(histo_en=histo, decode_ac=ac, err_max=em)
    try:
        H.drive(ref, data, q); H.drive(orc, data, q)
    except Exception as ex:
        print("case", k, mode, "exception", ex); bad += 1; continue
    r = F.differs(ref, orc, stats=bool(histo))
    if r: bad += 1; print("case", k, "mode", mode, "MISMATCH", r)
print("cases", n, "mismatches", bad, "%.0f s" % (time.time() - t))
