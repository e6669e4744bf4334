"""The libjpeg-turbo progressive 1920x1080 4:2:2 file WITHOUT restart markers (tests/golden/pillow), decoded three times: run under
`rocprofv3 --kernel-trace --stats` for the per-launch durations of the scan levels.   usage: python tools/prog_nodri_trace.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import jpegsnoop_amd as J
data = open(os.path.join(ROOT, "tests", "golden", "pillow", "p422_nodri_1920x1080_prog.jpg"), "rb").read()
dec = J.CimgDecode()
for i in range(3):
    t = time.perf_counter(); n = dec.DecodeProgressive(data); print("decode", i, "scans", n, "ms %.2f" % ((time.perf_counter() - t) * 1e3))
dec.close()
