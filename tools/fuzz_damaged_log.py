"""Damaged files through the drop-in call WITH a log callback (what a CjfifDecode does, source/JfifDecode.cpp:5299): the log text line for line
against the compiled reference, side outputs and status words against the oracle, and who produced them (jsnoop_last_side_mode: 3 = the chunked
side pass of round 6, 2 = the sequential mirror).   usage: python tools/fuzz_damaged_log.py [n_cases] [seed] [big=0/1]"""
import os, sys, time, collections, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import harness as H
import fuzz_util as F
import jpegsnoop_amd
H.build(["oracle", "synth"])
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
big = len(sys.argv) > 3 and sys.argv[3] == "1"
orc = H.oracle_backend(); gpu = H.Backend(jpegsnoop_amd.load(), "jsnoop_", "hip")
ref = H.ref_backend() if H.have_ref() else None
if big:
    B = [H.synth_jpeg(width=1920, height=1080, seed=61), H.synth_jpeg(width=1920, height=1080, restart_interval=120, seed=62), H.synth_jpeg(width=1280, height=720, hs=2, vs=1, restart_interval=17, seed=63)]
else:
    B = F.bases(H)
def damage(base):
    p = H.parse_jpeg(base); d = bytearray(base); n = p.scan_end - p.scan_start
    kind = int(rng.integers(8))
    i = p.scan_start + int(rng.integers(max(1, n - 64)))
    if kind == 0:
        for _ in range(int(rng.integers(1, 4))): d[p.scan_start + int(rng.integers(n - 2))] ^= 1 << int(rng.integers(8))
    elif kind == 1: d[i:i + int(rng.integers(8, 200))] = rng.integers(0, 255, int(rng.integers(8, 200))).astype(np.uint8).tobytes()     # garbage (no FF)
    elif kind == 2: d[i:i + 4] = b"\\xff\\x00\\xff\\x00"
    elif kind == 3: del d[i:i + int(rng.integers(1, 6))]
    elif kind == 4: d[i:i] = bytes([0xFF, 0xD0 + int(rng.integers(8))])
    elif kind == 5: d[i:i + int(rng.integers(4, 60))] = bytes(int(rng.integers(4, 60)))
    elif kind == 6: d[i] = int(rng.integers(255))
    else: d = d[:i]                                                                       # truncated (a carved file): the rest reads as zero bytes
    return bytes(d), kind
modes = collections.Counter(); bad = 0; times = collections.defaultdict(list); checked = 0
for k in range(n_cases):
    em = int(rng.choice([20, 20, 20, 3, 1]))
    if not big and k % 3 == 2:                                      # hostile headers too (tests/fuzz_util.py: sampling factors, table selectors, restart interval announced != used, ...)
        try: data, q, kind = F.mutate(H, rng, B[int(rng.integers(len(B)))]); kind += 100
        except Exception: continue
    else:
        data, kind = damage(B[int(rng.integers(len(B)))])
        try: q = H.parse_jpeg(data)
        except Exception: continue
    for b in (orc, gpu) + ((ref,) if ref else ()): b.set_options(decode_ac=1, err_max=em)
    try:
        t0 = time.perf_counter(); H.drive(gpu, data, q, quiet=0); ms = (time.perf_counter() - t0) * 1e3
    except Exception as ex:
        print("case", k, "kind", kind, "exception", ex); continue
    fl = gpu.lib.jsnoop_last_flags(gpu.h); path = gpu.lib.jsnoop_last_path(gpu.h); sm = gpu.lib.jsnoop_last_side_mode(gpu.h)
    if fl == 0: continue
    checked += 1; modes[(path, sm)] += 1; times[sm].append(ms)
    got = gpu.log_lines()
    H.drive(orc, data, q)
    r = F.differs(orc, gpu)
    if r:
        bad += 1; print("case", k, "kind", kind, "MISMATCH", r, "flags 0x%04x path %d side %d em %d" % (fl, path, sm, em))
        if r == "blk_dc":
            for ci, (x, y) in enumerate(zip(orc.blk_dc(), gpu.blk_dc())):
                if x is None: continue
                x = np.asarray(x); y = np.asarray(y); w = np.argwhere(x != y)
                if len(w): print("   comp", ci, "shape", x.shape, "cells differing", len(w), "first", w[:4].tolist(), "oracle", [int(x[tuple(t)]) for t in w[:4]], "gpu", [int(y[tuple(t)]) for t in w[:4]], "status", orc.status())
    if ref is not None:
        H.drive(ref, data, q, quiet=0); want = ref.log_lines()
        # (the compression figures divide by pos0 - first in the reference's `unsigned long`: 64 bits in the reference compiled here, 32 in the program as
        #  shipped (MSVC) and in this library -- they differ when the reader never got past the scan's first byte; every other line must agree)
        if orc.status()["pos0"] < orc.status()["first"]:
            strip = lambda L: [x for x in L if "Bits per pixel" not in x and "Compression Ratio" not in x]
            got, want = strip(got), strip(want)
        if got != want:
            bad += 1
            j = next((i for i, (a, b) in enumerate(zip(got + [None], want + [None])) if a != b), -1)
            print("case", k, "kind", kind, "LOG differs at line", j, "| got:", got[j] if j < len(got) else None, "| want:", want[j] if j < len(want) else None, "| flags 0x%04x path %d side %d em %d lens %d %d" % (fl, path, sm, em, len(got), len(want)))
print(json.dumps({"cases": n_cases, "flagged": checked, "mismatches": bad, "reference_log_compared": ref is not None, "path_side_mode": {str(k): v for k, v in modes.items()},
                  "call_ms_by_side_mode": {str(k): {"n": len(v), "median": round(float(np.median(v)), 2), "max": round(float(max(v)), 2)} for k, v in times.items()}}))
