"""N copies of BASELINE config 5's progressive file (1920x1080 4:2:2, RSTn every MCU row, successive approximation) as one batch: ms per
decode and the stage split.   usage: python tools/prog_batch.py [n_files = 64] [reps = 3]   (tools/pmc_progressive.sh profiles this)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import harness as H
import jpegsnoop_amd as J
H.build(["oracle", "synth"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
prog = H.synth_jpeg(width=1920, height=1080, hs=2, vs=1, restart_interval=120, quality=85, seed=55, progressive=2)
b = J.JpegBatch(); b.add_jpeg(prog); b.tile(n); b.upload(); b.decode(); b.sync()
ms, stages = b.decode_timed(reps)
print(json.dumps({"files": n, "ms_per_batch": round(ms, 3), "gpix_per_s": round(n * 1920 * 1080 / ms / 1e6, 2), "stages_ms": {k: round(v, 3) for k, v in stages.items()}}))
b.close()
