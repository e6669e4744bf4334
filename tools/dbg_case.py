import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import harness as H
import fuzz_util as F
import jpegsnoop_amd
H.build(["oracle", "synth"])
orc = H.oracle_backend(); gpu = H.Backend(jpegsnoop_amd.load(), "jsnoop_", "hip")
B = F.bases(H)
rng = np.random.default_rng(4242)
want = [int(a) for a in sys.argv[1:]] or [2]
for k in range(max(want) + 1):
    data, q, mode = F.mutate(H, rng, B[int(rng.integers(len(B)))])
    histo = int(rng.integers(2))
    if k not in want: continue
    for b in (orc, gpu): b.set_options(histo_en=histo)
    H.drive(orc, data, q); H.drive(gpu, data, q)
    r = F.differs(orc, gpu, stats=bool(histo))
    a, b_ = np.asarray(orc.mcu_map()).ravel(), np.asarray(gpu.mcu_map()).ravel()
    w = np.nonzero(a != b_)[0]
    print("case", k, "mode", mode, "differs", r, "flags 0x%04x" % gpu.lib.jsnoop_last_flags(gpu.h), "side", gpu.lib.jsnoop_last_side_mode(gpu.h), "nmcu", len(a), "geom", orc.geometry())
    print("  status orc", orc.status()); print("  status gpu", gpu.status())
    print("  mcu_map differing", len(w), "first", w[:12].tolist(), "orc", [hex(int(a[i])) for i in w[:12]], "gpu", [hex(int(b_[i])) for i in w[:12]])
    gpu.set_options(histo_en=histo)
