"""Batch-API sweep: hostile and valid images mixed in one submission (both sub-sequence lengths), DIB / planes / coefficient
rows vs oracle; then corrupted progressive files through jsnoop_decode_progressive (no oracle: must return, not hang)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import harness as H
import fuzz_util as F
import jpegsnoop_amd as J
H.build(["oracle", "synth"])
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
orc = H.oracle_backend()
B = F.bases(H)
bad = 0; total = 0
for rnd in range(12):
    files = []
    while len(files) < 24:
        data, q, mode = F.mutate(H, rng, B[int(rng.integers(len(B)))])
        if mode <= 5: files.append(data)                      # byte-level damage only: add_jpeg parses the header itself
        elif rng.integers(3) == 0: files.append(B[int(rng.integers(len(B)))])
    batch = J.JpegBatch(want_planes=True); idx = []
    batch.set_tuning(sub_wl=7 if rnd % 2 else 5)
    for f in files:
        try: idx.append(batch.add_jpeg(f))
        except Exception: idx.append(None)                        # header no longer walkable
    if len(batch) == 0: continue
    batch.upload(); batch.decode(); batch.sync()
    k = 0
    for f, i in zip(files, idx):
        if i is None: continue
        H.drive(orc, f); total += 1
        a = orc.dib()
        if a is None: continue
        if not np.array_equal(a, batch.dib(i)): bad += 1; print("round", rnd, "image", i, "DIB differs, path", batch.info(i)["path"], hex(batch.info(i)["flags"]))
    batch.close()
print("batch images", total, "mismatches", bad)
dec = J.CimgDecode(); ok = err = 0
for mode in (1, 2):
    base = H.synth_jpeg(width=160, height=96, hs=2, vs=1, restart_interval=5, progressive=mode, seed=9)
    for _ in range(150):
        d = bytearray(base)
        for _ in range(int(rng.integers(1, 6))): d[int(rng.integers(2, len(d)))] = int(rng.integers(256))
        try: dec.DecodeProgressive(bytes(d)); ok += 1
        except RuntimeError: err += 1
print("progressive hostile files: decoded", ok, "refused", err)
