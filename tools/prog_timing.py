"""Times jsnoop_decode_progressive on BASELINE config 5 (1920x1080 4:2:2, RSTn every MCU row, successive approximation)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import harness as H
import jpegsnoop_amd as J
H.build(["oracle", "synth"])
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
prog = H.synth_jpeg(width=1920, height=1080, hs=2, vs=1, restart_interval=120, quality=85, seed=55, progressive=mode)
dec = J.CimgDecode()
dec.DecodeProgressive(prog)
t = time.perf_counter()
for _ in range(10):
    dec.DecodeProgressive(prog)
print("mode", mode, "bytes", len(prog), "ms per decode", (time.perf_counter() - t) * 100)
