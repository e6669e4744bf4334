"""Replays case K of tools/fuzz_gpu.py (seed S) alone: python tools/dbg_fuzz_gpu_case.py S K"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import harness as H
import fuzz_util as F
import jpegsnoop_amd
H.build(["oracle", "synth"])
seed, K = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
orc = H.oracle_backend(); gpu = H.Backend(jpegsnoop_amd.load(), "jsnoop_", "hip")
B = F.bases(H)
for k in range(K + 1):
    data, q, mode = F.mutate(H, rng, B[int(rng.integers(len(B)))])
    histo = int(rng.integers(2)); ac = int(rng.integers(4) != 0); em = int(rng.choice([20, 20, 3, 1]))
    if k != K: continue
    for b in (orc, gpu): b.set_options(histo_en=histo, decode_ac=ac, err_max=em)
    H.drive(orc, data, q); H.drive(gpu, data, q)
    r = F.differs(orc, gpu, stats=bool(histo))
    a, b_ = orc.dib(), gpu.dib()
    w = np.argwhere((a != b_).any(axis=2)) if a is not None and b_ is not None else []
    print("case", k, "mode", mode, "differs", r, "path", gpu.lib.jsnoop_last_path(gpu.h), "flags 0x%04x" % gpu.lib.jsnoop_last_flags(gpu.h), "geom", orc.geometry(), "pixels differing", len(w), "first", w[:3].tolist() if len(w) else [], "status", orc.status())
    ma, mb = np.asarray(orc.mcu_map()).ravel(), np.asarray(gpu.mcu_map()).ravel()
    wm = np.nonzero(ma != mb)[0]
    print("  mcu_map differing", len(wm), "of", len(ma), "first", wm[:8].tolist(), "orc", [hex(int(ma[i])) for i in wm[:8]], "gpu", [hex(int(mb[i])) for i in wm[:8]], "status orc", orc.status(), "gpu", gpu.status(),
          "side_mode", gpu.lib.jsnoop_last_side_mode(gpu.h) if hasattr(gpu.lib, "jsnoop_last_side_mode") else "?", "rst", getattr(q, "rst_en", None), getattr(q, "rst_interval", None), "scan", q.scan_start)
    os.makedirs("gpurun_out", exist_ok=True)
    open("gpurun_out/dbg_case_%d_%d.jpg" % (seed, K), "wb").write(data)
