"""Randomised parity sweep on the GPU box: corrupted and re-headed scans, HIP path vs oracle (DIB, planes, side outputs, status).
usage: python tools/fuzz_gpu.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import harness as H
import jpegsnoop_amd
H.build(["oracle", "synth"])
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
orc = H.oracle_backend(); gpu = H.Backend(jpegsnoop_amd.load(), "jsnoop_", "hip")
bases = [H.synth_jpeg(width=w, height=h, seed=s, **kw) for s, (w, h, kw) in enumerate([
    (160, 96, {}), (141, 93, dict(hs=2, vs=1, restart_interval=3)), (128, 64, dict(hs=1, vs=1, restart_interval=1)), (97, 61, dict(gray=1)),
    (200, 120, dict(quality=25, restart_interval=7)), (96, 96, dict(quality=97, optimize_huffman=1)), (64, 48, dict(hs=1, vs=2))])]
def same(a, b):
    da, db = a.dib(), b.dib()
    if (da is None) != (db is None): return "preview"
    if da is None: return None
    if not np.array_equal(da, db): return "dib"
    for x, y in zip(a.planes(), b.planes()):
        if x is not None and not np.array_equal(x, y): return "planes"
    if not np.array_equal(a.mcu_map(), b.mcu_map()): return "mcu_map"
    for x, y in zip(a.blk_dc(), b.blk_dc()):
        if x is not None and not np.array_equal(x, y): return "blk_dc"
    if not np.array_equal(a.dht_histo(), b.dht_histo()): return "histo"
    if a.status() != b.status(): return "status %s %s" % (a.status(), b.status())
    if a.bright_avg() != b.bright_avg(): return "bright"
    return None
bad = 0; paths = {1: 0, 2: 0, 0: 0}
for k in range(n_cases):
    base = bases[int(rng.integers(len(bases)))]
    p = H.parse_jpeg(base); d = bytearray(base)
    mode = int(rng.integers(7))
    s, e = p.scan_start, p.scan_end
    if mode == 0:
        for _ in range(int(rng.integers(1, 5))): d[int(rng.integers(s, e))] ^= 1 << int(rng.integers(8))
    elif mode == 1: d = d[: int(rng.integers(s + 1, len(d)))]
    elif mode == 2: i = int(rng.integers(s, e)); d[i:i] = bytes([0xFF, int(rng.integers(0xC0, 0xFF))])
    elif mode == 3: i = int(rng.integers(s, e)); d[i:i] = b"\xff" * int(rng.integers(2, 5))
    elif mode == 4: i = int(rng.integers(s, e - 8)); del d[i:i + int(rng.integers(1, 6))]
    elif mode == 5: i = int(rng.integers(s, e)); d[i:i] = bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
    data = bytes(d); q = H.parse_jpeg(data) if mode != 1 else p
    if mode == 6:
        q.comps = [(c[0], int(rng.integers(1, 5)), int(rng.integers(1, 5)), c[3]) for c in p.comps]
    try:
        H.drive(orc, data, q); H.drive(gpu, data, q)
    except Exception as ex:
        print("case", k, "mode", mode, "exception", ex); bad += 1; continue
    paths[gpu.lib.jsnoop_last_path(gpu.h)] += 1
    r = same(orc, gpu)
    if r: bad += 1; print("case", k, "mode", mode, "MISMATCH:", r)
print("cases", n_cases, "mismatches", bad, "paths", paths)
