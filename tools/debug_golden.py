import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import harness as H
from golden_util import load_case, manifest, record
import jpegsnoop_amd
H.build(["oracle", "synth"])
M = manifest()
orc = H.oracle_backend(); gpu = H.Backend(jpegsnoop_amd.load(), "jsnoop_", "hip")
for name in sorted(M["cases"]):
    if not name.startswith(sys.argv[1]): continue
    data = load_case(name)
    for mode, ac in (("full_idct", 1), ("dc_only", 0)):
        for b in (orc, gpu): b.set_options(decode_ac=ac)
        H.drive(orc, data); H.drive(gpu, data)
        r = record(H, gpu); w = M["cases"][name][mode]
        diff = [k for k in w if r.get(k) != w[k]]
        print(name, mode, "path", gpu.lib.jsnoop_last_path(gpu.h), "flags", hex(gpu.lib.jsnoop_last_flags(gpu.h)), "diff", diff)
        if "planes" in diff:
            a, b = orc.planes()[0], gpu.planes()[0]
            d = np.argwhere(a != b); print("   Y plane diffs", len(d), "first", d[:3].tolist(), "orc", a[tuple(d[0])], "gpu", b[tuple(d[0])], "bbox", d.min(0).tolist(), d.max(0).tolist())
