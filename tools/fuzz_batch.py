"""Batch-API sweep: hostile and valid images mixed in one submission (both sub-sequence lengths), DIB vs oracle; the per-image results of a batch
(jsnoop_batch_side_outputs / jsnoop_batch_log) of hostile files against the oracle's side outputs and the compiled reference's log; then corrupted
progressive files through jsnoop_decode_progressive (no oracle: must return, not hang)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import harness as H
import fuzz_util as F
import jpegsnoop_amd as J
H.build(["oracle", "synth"])
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
orc = H.oracle_backend()
B = F.bases(H)
bad = 0; total = 0
for rnd in range(12):
    files = []
    while len(files) < 24:
        data, q, mode = F.mutate(H, rng, B[int(rng.integers(len(B)))])
        if mode <= 5: files.append(data)                      # byte-level damage only: add_jpeg parses the header itself
        elif rng.integers(3) == 0: files.append(B[int(rng.integers(len(B)))])
    batch = J.JpegBatch(want_planes=True); idx = []
    batch.set_tuning(sub_wl=7 if rnd % 2 else 5, split=2 if rnd % 2 else 0, **(dict(cand_rounds=-1) if rnd % 4 >= 2 else {}))     # (cand_rounds = -1: synchronisation by rounds -- the list rounds of larger jobs; split = 2: two halves on two streams, the second a stage behind)
    for f in files:
        try: idx.append(batch.add_jpeg(f))
        except Exception: idx.append(None)                        # header no longer walkable
    if len(batch) == 0: continue
    batch.upload(); batch.decode(); batch.sync()
    k = 0
    for f, i in zip(files, idx):
        if i is None: continue
        H.drive(orc, f); total += 1
        a = orc.dib()
        if a is None: continue
        if not np.array_equal(a, batch.dib(i)): bad += 1; print("round", rnd, "image", i, "DIB differs, path", batch.info(i)["path"], hex(batch.info(i)["flags"]))
    batch.close()
print("batch images", total, "mismatches", bad)
# ---- per-image results of a batch: side outputs and log text of hostile files (what DoBatchFileProcess leaves per file, source/JPEGsnoopCore.cpp:765-845)
ref = H.ref_backend() if H.have_ref() else None
gpu = H.Backend(J.load(), "jsnoop_", "hip")                       # the table-state object of jsnoop_batch_add
PREFIX = ("", "W:", "E:")
sbad = lbad = stot = 0
for rnd in range(10):
    files = []
    while len(files) < 16:
        data, q, mode = F.mutate(H, rng, B[int(rng.integers(len(B)))])
        if mode <= 5 and mode != 1:
            try: files.append((data, H.parse_jpeg(data)))
            except Exception: pass
    batch = J.JpegBatch(want_planes=True); batch.enable_log(True)
    for data, q in files:
        H.push_tables(gpu, q); batch.add(gpu.h, data, q.scan_start)
    batch.upload(); batch.decode(); batch.sync()
    for i, (data, q) in enumerate(files):
        H.drive(orc, data, q)
        if orc.dib() is None: continue
        stot += 1
        so = batch.side_outputs(i)
        st = orc.status(); keys = ("scan_bad", "scan_end", "restart_read", "num_pixels", "pos0", "align", "warn_bad", "first")
        same = np.array_equal(so["mcu_map"], orc.mcu_map()) and np.array_equal(so["dht_histo"], orc.dht_histo()) and [int(v) for v in so["status"].values()] == [int(st[k]) for k in keys] \
               and all(x is None or np.array_equal(x, y) for x, y in zip(orc.blk_dc(), so["blk_dc"]))
        if not same: sbad += 1; print("side round", rnd, "image", i, "differs, path", batch.info(i)["path"], hex(batch.info(i)["flags"]))
        if ref is not None:
            H.drive(ref, data, q, quiet=0)
            got = [PREFIX[min(max(l, 0), 2)] + t for l, t in batch.log_lines(i)]
            want = ref.log_lines()
            if orc.status()["pos0"] < orc.status()["first"]:
                strip = lambda L: [x for x in L if "Bits per pixel" not in x and "Compression Ratio" not in x]; got, want = strip(got), strip(want)
            if got != want:
                lbad += 1; j = next((k for k, (a, b_) in enumerate(zip(got + [None], want + [None])) if a != b_), -1)
                print("log round", rnd, "image", i, "line", j, "| got:", got[j] if j < len(got) else None, "| want:", want[j] if j < len(want) else None, "| flags", hex(batch.info(i)["flags"]))
    batch.close()
print("batch side outputs / logs of", stot, "hostile images: side mismatches", sbad, "log mismatches", lbad, "(reference log compared:", ref is not None, ")")
dec = J.CimgDecode(); ok = err = 0
for mode in (1, 2):
    base = H.synth_jpeg(width=160, height=96, hs=2, vs=1, restart_interval=5, progressive=mode, seed=9)
    for _ in range(150):
        d = bytearray(base)
        for _ in range(int(rng.integers(1, 6))): d[int(rng.integers(2, len(d)))] = int(rng.integers(256))
        try: dec.DecodeProgressive(bytes(d)); ok += 1
        except RuntimeError: err += 1
print("progressive hostile files: decoded", ok, "refused", err)
