"""Clean batches WITH restart markers (walk_slow resets (c, k) at a marker only for speculative walks since round 4 -- does a well-formed
DRI stream pay for it?): N x 1920x1080 4:2:0 with a restart marker per MCU row through the batch path, ms per decode, stage split, flags, every image
against the oracle.   usage: python tools/dri_batch.py [N ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import harness as H
import jpegsnoop_amd as J
H.build(["oracle", "synth"])
ns = [int(a) for a in sys.argv[1:]] or [1, 16, 64, 1024]
files = [H.synth_jpeg(width=1920, height=1080, hs=2, vs=2, quality=85, restart_interval=120, seed=300 + i) for i in range(16)]
orc = H.oracle_backend(); want = []
for f in files:
    H.drive(orc, f); want.append(J.dib_checksum_numpy(orc.dib()))
out = {}
for n in ns:
    b = J.JpegBatch()
    for f in files[:min(n, 16)]:
        b.add_jpeg(f)
    if n > 16:
        b.tile(n)
    b.upload(); b.decode(); b.sync()
    ms, st = b.decode_timed(5)
    ok = all(int(s) == want[i % 16] for i, s in enumerate(b.dib_checksums())) and not any(b.info(i)["flags"] for i in range(n))
    out[str(n)] = {"ms": round(ms, 4), "sync_ms": round(st["sync"], 4), "write_ms": round(st["write"], 4), "bit_exact_no_flags": bool(ok)}
    b.close()
print(json.dumps(out))
