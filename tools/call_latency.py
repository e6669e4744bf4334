"""What one call costs a caller of the single-image API: wall time of jsnoop_decode_scan_img (staging copy + H2D + decode + wait + flags)
per call, tables already set (CjfifDecode's setter calls precede it once), repeated on the same decoder object; and of
jsnoop_decode_progressive for the progressive form of config 5.  usage: python tools/call_latency.py"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpegsnoop_amd as J
from oracle import harness as H
H.build(["oracle", "synth"])
gpu = H.Backend(J.load(), "jsnoop_", "hip")
out = {}
for name, kw in (("640x480_444", dict(width=640, height=480, hs=1, vs=1, seed=3)), ("1080p_420", dict(width=1920, height=1080, hs=2, vs=2, seed=100)), ("2160p_420", dict(width=3840, height=2160, hs=2, vs=2, seed=77)),
                 ("1080p_422_rst", dict(width=1920, height=1080, hs=2, vs=1, restart_interval=120, seed=55))):
    f = H.synth_jpeg(quality=85, **kw)
    p = H.drive(gpu, f)
    buf = (C.c_uint8 * len(f)).from_buffer_copy(f)
    for _ in range(3):
        gpu.decode_scan_img(C.cast(buf, C.c_void_p), len(f), p.scan_start, 1, 1)
    n = 30; t0 = time.perf_counter()
    for _ in range(n):
        gpu.decode_scan_img(C.cast(buf, C.c_void_p), len(f), p.scan_start, 1, 1)
    out[name] = {"ms_per_call": round((time.perf_counter() - t0) / n * 1e3, 3), "file_bytes": len(f)}
prog = H.synth_jpeg(width=1920, height=1080, hs=2, vs=1, restart_interval=120, quality=85, seed=55, progressive=2)
dec = J.CimgDecode()
for _ in range(3):
    dec.DecodeProgressive(prog)
t0 = time.perf_counter()
for _ in range(10):
    dec.DecodeProgressive(prog)
out["1080p_422_rst_progressive"] = {"ms_per_call": round((time.perf_counter() - t0) / 10 * 1e3, 3), "file_bytes": len(prog)}
print(json.dumps(out))
