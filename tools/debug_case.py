import sys, os, ctypes as C, subprocess
sys.path.insert(0, '.')
import numpy as np
from oracle import harness as H
import jpegsnoop_amd as J
H.build(["oracle", "synth"])
subprocess.check_call("gcc -O2 -Ioracle tools/syncdist.c oracle/jpeg_synth.c -lm -o /tmp/syncdist && /tmp/syncdist 2 2 1024 0 /tmp/truth.bin", shell=True)
truth = np.fromfile('/tmp/truth.bin', np.uint32).reshape(-1, 3)
lib = J.load()
lib.jsnoop_batch_debug_copy.restype = C.c_uint64
lib.jsnoop_batch_debug_copy.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64]
data = H.synth_jpeg(seed=11, width=1920, height=1080)
for nl in (1, 2, 3):
    os.environ['JSNOOP_SYNC_LAUNCHES'] = str(nl)
    b = J.JpegBatch(want_planes=False)
    b.add_jpeg(data); b.upload(); b.decode()
    buf = np.zeros(1 << 22, np.uint32)
    n = lib.jsnoop_batch_debug_copy(b._h, 0, buf.ctypes.data, buf.nbytes) // 4
    c = buf[:n].reshape(6, -1).copy()
    ns = len(truth)
    gp, gs = c[0][:ns], c[1][:ns]
    gc, gk = (gs >> 8) & 255, gs & 255
    ok = (gp == truth[:, 0]) & (gc == truth[:, 1]) & (gk == truth[:, 2])
    bad = np.nonzero(~ok)[0]
    print('launches', nl, 'nsub', ns, 'mismatching out states', len(bad), 'first', bad[:12], 'per WG', np.bincount(bad // 256, minlength=18) if len(bad) else '')
    if len(bad):
        i = bad[0]
        for j in range(max(0, i - 2), i + 3):
            print('  i', j, 'gpu out', gp[j], gc[j], gk[j], 'truth', truth[j], 'gpu in', c[2][j], (c[3][j] >> 8) & 255, c[3][j] & 255, 'nblk', c[4][j])
    b.sync(); b.close()
