"""BASELINE config 5 as a batch: N progressive 1920x1080 4:2:2 files with RSTn every MCU row (10 scans, successive approximation), decoded
`reps` times -- the command rocprofv3 is pointed at for profiles/r03_prog_*.   usage: python tools/prog_batch_run.py [N] [reps] [restart_interval]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import harness as H
import jpegsnoop_amd as J
H.build(["oracle", "synth"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ri = int(sys.argv[3]) if len(sys.argv) > 3 else 120
prog = H.synth_jpeg(width=1920, height=1080, hs=2, vs=1, restart_interval=ri, quality=85, seed=55, progressive=2)
b = J.JpegBatch(); b.add_jpeg(prog); b.tile(n); b.upload(); b.decode(); b.sync()
t = time.perf_counter()
for _ in range(reps):
    b.decode()
b.sync()
ms = (time.perf_counter() - t) / reps * 1e3
msb, st = b.decode_timed(3)
print("progressive batch", n, "x 1080p 4:2:2 ri", ri, ": %.3f ms per batch, %.4f ms per image, %.1f Mpix/s" % (ms, ms / n, n * 1920 * 1080 / ms / 1e3), {k: round(v, 3) for k, v in st.items() if v > 0.001})
b.close()
