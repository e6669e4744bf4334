import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["JSNOOP_DEBUG_TIMING"] = "1"
import jpegsnoop_amd as J
from oracle import harness as H
H.build(["oracle", "synth"])
gpu = H.Backend(J.load(), "jsnoop_", "hip")
for name, kw in (("1080p_420", dict(width=1920, height=1080, hs=2, vs=2, seed=100)), ("2160p_420", dict(width=3840, height=2160, hs=2, vs=2, seed=77))):
    f = H.synth_jpeg(quality=85, **kw)
    p = H.drive(gpu, f)
    buf = (C.c_uint8 * len(f)).from_buffer_copy(f)
    for _ in range(3):
        gpu.decode_scan_img(C.cast(buf, C.c_void_p), len(f), p.scan_start, 1, 1)
    sys.stderr.write("==== %s\n" % name); sys.stderr.flush()
    for _ in range(3):
        gpu.decode_scan_img(C.cast(buf, C.c_void_p), len(f), p.scan_start, 1, 1)
