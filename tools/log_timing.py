"""Wall time of DecodeScanImg WITH its log text (side outputs, messages, report) for a clean and an overflow-damaged 1080p file."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import harness as H
import jpegsnoop_amd as J
H.build(["oracle", "synth"])
gpu = H.Backend(J.load(), "jsnoop_", "hip")
base = H.synth_jpeg(width=1920, height=1080, seed=9)
p = H.parse_jpeg(base)
d = bytearray(base); i = p.scan_start + int((p.scan_end - p.scan_start) * 0.95)
while d[i] == 0xFF or d[i - 1] == 0xFF or (d[i] ^ 0x10) == 0xFF: i += 1
d[i] ^= 0x10
for label, data in (("clean", base), ("flip (overflow)", bytes(d))):
    H.drive(gpu, data, quiet=0)
    t = time.perf_counter()
    for _ in range(3): H.drive(gpu, data, quiet=0)
    ms = (time.perf_counter() - t) / 3 * 1e3
    print("%-16s decode + log text: %8.2f ms   flags 0x%04x   %d log lines" % (label, ms, gpu.lib.jsnoop_last_flags(gpu.h), len(gpu.log_lines())))
