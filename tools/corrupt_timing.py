"""Damaged files: wall time of decode + sync (batch of one, bytes resident) for a 1080p / 4K 4:2:0 file with ONE flipped scan bit,
with and without restart markers, beside the undamaged file.   usage: python tools/corrupt_timing.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import harness as H
import jpegsnoop_amd as J
H.build(["oracle", "synth"])
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
orc = H.oracle_backend()
for label, kw in (("1080p", dict(width=1920, height=1080)), ("1080p rst", dict(width=1920, height=1080, restart_interval=120)),
                  ("4K", dict(width=3840, height=2160)), ("4K rst", dict(width=3840, height=2160, restart_interval=240))):
    base = H.synth_jpeg(seed=9, **kw)
    p = H.parse_jpeg(base)
    for where in (None, 0.5, 0.05, 0.95, "cut 0.5", "cut 0.9", "marker 0.3", "delete 0.7"):
        d = bytearray(base)
        if isinstance(where, str):
            kind, frac = where.split(); i = p.scan_start + int((p.scan_end - p.scan_start) * float(frac))
            if kind == "cut": d = d[:i]                                   # a truncated (carved) file
            elif kind == "marker": d[i:i + 2] = b"\xff\xe3"              # two bytes overwritten by a stray marker
            else: del d[i:i + 3]                                          # three bytes lost
        elif where is not None:
            i = p.scan_start + int((p.scan_end - p.scan_start) * where)
            while d[i] == 0xFF or d[i - 1] == 0xFF or (d[i] ^ 0x10) == 0xFF: i += 1     # keep the damage a plain data byte
            d[i] ^= 0x10
        d = bytes(d)
        b = J.JpegBatch(); b.add_jpeg(d); b.upload(); b.decode(); b.sync()
        t = time.perf_counter()
        for _ in range(reps): b.decode(); b.sync()
        ms = (time.perf_counter() - t) / reps * 1e3
        H.drive(orc, d)
        ok = int(b.dib_checksums()[0]) == J.dib_checksum_numpy(orc.dib())
        inf = b.info(0)
        print(f"{label:10s} flip@{where}: {ms:9.2f} ms  path {inf['path']} flags 0x{inf['flags']:04x} exact={ok}")
        b.close()
