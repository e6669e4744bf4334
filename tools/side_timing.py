"""Times the on-demand side-output pass (parallel vs the sequential exact-mirror kernel) and checks that both agree."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import harness as H
import ctypes as C
import jpegsnoop_amd
from jpegsnoop_amd import capi
H.build(["oracle", "synth"])
gpu = H.Backend(jpegsnoop_amd.load(), "jsnoop_", "hip")
def side_exact(on):
    t = capi.Tuning(); gpu.lib.jsnoop_tuning_defaults(C.byref(t)); t.cross_checks = capi.XC_SIDE_EXACT if on else 0
    assert gpu.lib.jsnoop_set_tuning(gpu.h, C.byref(t)) == 0
for kw in (dict(width=1920, height=1080, seed=1), dict(width=1920, height=1080, hs=2, vs=1, restart_interval=120, seed=2), dict(width=3840, height=2160, seed=3),
           dict(width=1920, height=1080, restart_interval=1, seed=4)):
    data = H.synth_jpeg(**kw)
    res = {}
    for mode in ("parallel", "exact"):
        side_exact(mode == "exact")
        H.drive(gpu, data)
        t0 = time.perf_counter(); mm = gpu.mcu_map(); dt = time.perf_counter() - t0
        res[mode] = (dt, mm, gpu.blk_dc(), gpu.dht_histo(), gpu.status())
    a, b = res["parallel"], res["exact"]
    same = np.array_equal(a[1], b[1]) and all(np.array_equal(x, y) for x, y in zip(a[2], b[2]) if x is not None) and np.array_equal(a[3], b[3]) and a[4] == b[4]
    print(kw, "parallel %.1f ms, exact %.1f ms, identical %s" % (a[0] * 1e3, b[0] * 1e3, same))
