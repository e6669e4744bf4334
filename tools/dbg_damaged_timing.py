"""Where the wall time of the drop-in call goes for damaged 1080p files (JSNOOP_DEBUG_TIMING lines on stderr): python tools/dbg_damaged_timing.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["JSNOOP_DEBUG_TIMING"] = "1"
from oracle import harness as H
import jpegsnoop_amd
H.build(["oracle", "synth"])
gpu = H.Backend(jpegsnoop_amd.load(), "jsnoop_", "hip")
def damage(base, kind, frac):
    p = H.parse_jpeg(base); d = bytearray(base); i = p.scan_start + int((p.scan_end - p.scan_start) * frac)
    if kind == "cut": d = d[:i]
    elif kind == "marker": d[i:i + 2] = b"\xff\xe3"
    elif kind == "zeros": d[i:i + 40] = bytes(40)
    elif kind == "delete": del d[i:i + 3]
    elif kind == "rst": d[i:i] = b"\xff\xd5"
    return bytes(d)
for rst in ([int(x) for x in os.environ.get("DBG_RST", "0,120").split(",")]):
    base = H.synth_jpeg(width=1920, height=1080, seed=9, restart_interval=rst)
    for kind in (os.environ.get("DBG_KINDS", "cut,marker,zeros,delete,rst").split(",")):
        data = damage(base, kind, 0.5)
        for _ in range(2): H.drive(gpu, data, quiet=0)
        sys.stderr.write("==== rst %d %s: " % (rst, kind)); sys.stderr.flush()
        t = time.perf_counter(); H.drive(gpu, data, quiet=0); ms = (time.perf_counter() - t) * 1e3
        sys.stderr.write("     -> %.2f ms flags 0x%04x path %d side %d lines %d\n" % (ms, gpu.lib.jsnoop_last_flags(gpu.h), gpu.lib.jsnoop_last_path(gpu.h), gpu.lib.jsnoop_last_side_mode(gpu.h), len(gpu.log_lines())))
