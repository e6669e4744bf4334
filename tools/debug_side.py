"""Debug aid: compare the side outputs of the HIP path with the oracle on one synthetic case and print the differences."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import harness as H
import jpegsnoop_amd
kw = json.loads(sys.argv[1]) if len(sys.argv) > 1 else dict(width=161, height=97, gray=1, restart_interval=2, seed=43)
H.build(["oracle", "synth"])
data = H.synth_jpeg(**kw)
orc = H.oracle_backend(); gpu = H.Backend(jpegsnoop_amd.load(), "jsnoop_", "hip")
H.drive(orc, data); H.drive(gpu, data)
a, b = orc.mcu_map().ravel(), gpu.mcu_map().ravel()
d = np.nonzero(a != b)[0]
print("mcu_map diffs:", len(d), "of", a.size)
for i in d[:12]:
    print(i, "orc byte", a[i] >> 4, "bit", a[i] & 15, " gpu byte", b[i] >> 4, "bit", b[i] & 15, " prev orc", a[i - 1] >> 4 if i else None)
    o = int(a[i] >> 4); print("   raw bytes around orc pos:", data[o - 4:o + 6].hex())
print("status", orc.status(), gpu.status())
print("histo equal", np.array_equal(orc.dht_histo(), gpu.dht_histo()))
for x, y in zip(orc.blk_dc(), gpu.blk_dc()):
    if x is not None: print("blk_dc equal", np.array_equal(x, y))
