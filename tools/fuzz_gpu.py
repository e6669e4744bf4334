"""Randomised parity sweep on the GPU box: corrupted scans and mutated headers (tests/fuzz_util.py), HIP path vs oracle
(DIB, planes, side outputs, status words, colour statistics).   usage: python tools/fuzz_gpu.py [n_cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import collections
import numpy as np
from oracle import harness as H
import fuzz_util as F
import jpegsnoop_amd
H.build(["oracle", "synth"])
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
orc = H.oracle_backend(); gpu = H.Backend(jpegsnoop_amd.load(), "jsnoop_", "hip")
B = F.bases(H)
bad = 0; paths = collections.Counter()
for k in range(n_cases):
    data, q, mode = F.mutate(H, rng, B[int(rng.integers(len(B)))])
    histo = int(rng.integers(2)); ac = int(rng.integers(4) != 0); em = int(rng.choice([20, 20, 3, 1]))
    for b in (orc, gpu):
        b.set_options(histo_en=histo, decode_ac=ac, err_max=em)
    try:
        H.drive(orc, data, q); H.drive(gpu, data, q)
    except Exception as ex:
        print("case", k, "mode", mode, "exception", ex); bad += 1; continue
    paths[gpu.lib.jsnoop_last_path(gpu.h)] += 1
    r = F.differs(orc, gpu, stats=bool(histo))
    if r: bad += 1; print("case", k, "mode", mode, "MISMATCH:", r, "path", gpu.lib.jsnoop_last_path(gpu.h), "flags 0x%04x" % gpu.lib.jsnoop_last_flags(gpu.h), "opts", histo, ac, em)
print("cases", n_cases, "mismatches", bad, "paths", dict(paths))
