import sys, time, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '.')
import numpy as np
import jpegsnoop_amd as J
from oracle import harness as H
orc = H.oracle_backend()
out = {}
for label, kwd in (("1080p", dict(width=1920, height=1080)), ("1080p_rst", dict(width=1920, height=1080, restart_interval=120))):
    based = H.synth_jpeg(seed=9, **kwd); pd = H.parse_jpeg(based)
    for kind in ("garbage", "garbage_noff", "rst_in_block", "badcode"):
        d = bytearray(based); i = pd.scan_start + int((pd.scan_end - pd.scan_start) * 0.3)
        if kind == "garbage": d[i:i+256] = np.random.RandomState(5).randint(0, 256, 256).astype(np.uint8).tobytes()
        elif kind == "garbage_noff": d[i:i+256] = np.random.RandomState(5).randint(0, 255, 256).astype(np.uint8).tobytes()
        elif kind == "badcode": d[i:i+4] = b"\xff\x00\xff\x00"
        else:
            if label == "1080p": continue
            j = bytes(d).index(b"\xff\xd1", pd.scan_start); del d[j-2:j]
        d = bytes(d)
        b = J.JpegBatch(); b.add_jpeg(d); b.upload(); b.decode(); b.sync()
        t0 = time.perf_counter(); b.decode(); b.sync(); ms = (time.perf_counter() - t0) * 1e3
        H.drive(orc, d); inf = b.info(0)
        out[label + "_" + kind] = dict(ms=round(ms, 2), path=int(inf["path"]), flags="0x%04x" % inf["flags"], exact=bool(int(b.dib_checksums()[0]) == J.dib_checksum_numpy(orc.dib())))
        b.close()
print(json.dumps(out, indent=1))
