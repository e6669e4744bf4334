#!/usr/bin/env python3
"""Golden results for LARGE pictures, written by the COMPILED REFERENCE (oracle/_ref, needs /root/reference):
tests/golden/large_manifest.json holds, per case, the generator parameters of oracle/jpeg_synth.c (the file itself is
reproduced from them: deterministic C code, its SHA-256 is recorded), and SHA-256 digests of what the reference's
DecodeScanImg leaves behind -- the DIB, the three int16 planes, the MCU file map, the status words.  The committed .jpg
golden files are all <= 160x120; these cases pin 640x480 ... 3840x2160 (and a long restart-marker stream) without carrying
megabytes of pixels, on the GPU box as well, where the reference does not travel."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import harness as H          # noqa: E402

CASES = {
    "c1_640x480_444":        dict(width=640, height=480, hs=1, vs=1, quality=85, seed=9101),
    "c3_1920x1080_420":      dict(width=1920, height=1080, hs=2, vs=2, quality=85, seed=9102),
    "c2_3840x2160_420":      dict(width=3840, height=2160, hs=2, vs=2, quality=85, seed=9103),
    "c5_1920x1080_422_rst":  dict(width=1920, height=1080, hs=2, vs=1, restart_interval=120, quality=85, seed=9104),
    "odd_2001x1333_420_q97": dict(width=2001, height=1333, hs=2, vs=2, quality=97, seed=9105),
    "gray_2048x2048_q40":    dict(width=2048, height=2048, gray=1, quality=40, seed=9106),
    "wide_4096x512_440_rst": dict(width=4096, height=512, hs=1, vs=2, restart_interval=7, quality=75, seed=9107),
}


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def record(backend, data):
    H.drive(backend, data)
    rec = {"dib": digest(backend.dib()), "planes": [digest(p) if p is not None else None for p in backend.planes()],
           "mcu_map": digest(backend.mcu_map()), "blk_dc": [digest(p) if p is not None else None for p in backend.blk_dc()],
           "status": {k: int(v) for k, v in backend.status().items()}, "bright_avg": [int(v) for v in backend.bright_avg()]}
    return rec


def main():
    H.build(["ref", "synth"])
    ref = H.ref_backend()
    out = {}
    for name, kw in CASES.items():
        data = H.synth_jpeg(**kw)
        out[name] = {"params": kw, "jpeg_sha256": hashlib.sha256(data).hexdigest(), "jpeg_bytes": len(data), **record(ref, data)}
        print(name, len(data), out[name]["dib"][:16])
    json.dump(out, open(os.path.join(HERE, "large_manifest.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
