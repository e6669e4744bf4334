"""How the parallel path classifies damaged files: histogram of JSNOOP_FLAG_* over random single-byte damage (flip / overwrite / delete /
insert) of a 1080p 4:2:0 file, with and without restart markers, and the decode time per class.
usage: python tools/damage_survey.py [cases] [seed]"""
import collections, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import harness as H
import jpegsnoop_amd as J
H.build(["oracle", "synth"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
W, Hh = (int(os.environ.get("SURVEY_W", 1920)), int(os.environ.get("SURVEY_H", 1080)))
for label, kw in (("no RST", {}), ("RST/row", dict(restart_interval=W // 16))):
    base = H.synth_jpeg(width=W, height=Hh, seed=9, **kw)
    p = H.parse_jpeg(base)
    hist = collections.Counter(); tms = collections.defaultdict(list)
    b = J.JpegBatch()
    for k in range(n):
        d = bytearray(base)
        i = int(rng.integers(p.scan_start, p.scan_end - 2))
        kind = int(rng.integers(4))
        if kind == 0: d[i] ^= 1 << int(rng.integers(8))
        elif kind == 1: d[i] = int(rng.integers(256))
        elif kind == 2: del d[i]
        else: d[i:i] = bytes([int(rng.integers(256))])
        b.clear(); b.add_jpeg(bytes(d)); b.upload()
        t = time.perf_counter(); b.decode(); b.sync(); ms = (time.perf_counter() - t) * 1e3
        inf = b.info(0)
        key = (inf["path"], "0x%04x" % inf["flags"])
        hist[key] += 1; tms[key].append(ms)
    b.close()
    print(label, "cases", n)
    for key, c in sorted(hist.items(), key=lambda kv: -kv[1]):
        print("   path %d flags %s : %4d   median %.2f ms" % (key[0], key[1], c, float(np.median(tms[key]))))
