"""Where the wall time of ONE drop-in call goes (jsnoop_decode_scan_img with and without a log callback / report): python tools/call_breakdown.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["JSNOOP_DEBUG_TIMING"] = "1"
import numpy as np
from oracle import harness as H
import jpegsnoop_amd
H.build(["oracle", "synth"])
gpu = H.Backend(jpegsnoop_amd.load(), "jsnoop_", "hip")
base = H.synth_jpeg(width=1920, height=1080, seed=9)
p = H.parse_jpeg(base)
d = bytearray(base); i = p.scan_start + int((p.scan_end - p.scan_start) * 0.95)
while d[i] == 0xFF or d[i - 1] == 0xFF or (d[i] ^ 0x10) == 0xFF: i += 1
d[i] ^= 0x10
for name, data in (("clean", base), ("flip", bytes(d))):
    for display, quiet in ((1, 0), (1, 1), (0, 1)):
        H.drive(gpu, data, display=display, quiet=quiet)
        t = time.perf_counter(); H.drive(gpu, data, display=display, quiet=quiet); ms = (time.perf_counter() - t) * 1e3
        print(name, "display", display, "quiet", quiet, "ms %.2f" % ms, "flags 0x%04x side %d lines %d" % (gpu.lib.jsnoop_last_flags(gpu.h), gpu.lib.jsnoop_last_side_mode(gpu.h), len(gpu.log_lines())), flush=True)
