"""Generates tests/golden/: small JPEG inputs + the outputs of the UNMODIFIED reference on them.

Run in the build container (needs /root/reference):  python tests/golden/make_golden.py
The reference is compiled in place by `make -C oracle ref` (oracle/_ref/libjsnoop_ref.so); the
fixtures committed here are what travels to machines without /root/reference (the GPU box).

For every case the manifest stores sha256 of: DIB bytes, the three int16 planes, the MCU file map,
the block-DC maps, the Huffman code-length histogram, plus the status words, brightest-pixel /
average-Y record and image size, all as produced by reference CimgDecode::DecodeScanImg in
Full-IDCT mode (bDecodeScanImgAc=true, bHistoEn=false) -- a second record for DC-only mode -- and a third
for the bHistoEn colour path (statistics after the decode and after two preview re-renders).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import harness as H  # noqa: E402


def record(b):
    dib = b.dib()
    if dib is None:
        return {"preview": False, "size": list(b.image_size())}
    r = {"preview": bool(b.is_preview_ready()), "size": list(b.image_size()), "dib": H.hash_bytes(dib)}
    r["planes"] = [H.hash_bytes(p) if p is not None else None for p in b.planes()]
    r["mcu_map"] = H.hash_bytes(b.mcu_map())
    r["blk_dc"] = [H.hash_bytes(p) if p is not None else None for p in b.blk_dc()]
    r["histo"] = H.hash_bytes(b.dht_histo())
    r["status"] = {k: int(v) for k, v in b.status().items()}
    r["bright_avg"] = [int(v) for v in b.bright_avg()]
    return r


def stats_record(b):
    """bHistoEn statistics (ConvertYCCtoRGB / CapYccRange / CapRgbRange, reference :4229-4601)."""
    st = b.color_stats()
    words = np.concatenate([st["histo"].view(np.uint32), np.array([st["count"]], np.uint32), st["clip"], st["rgb"].ravel(), st["yfull"]])
    return {"sha256": H.hash_bytes(words), "count": st["count"], "clip": [int(v) for v in st["clip"]], "histo": [int(v) for v in st["histo"]]}


def histo_record(b, data):
    """Decode with bHistoEn, then two re-renders (the reference keeps accumulating, it only clears in DecodeScanImg)."""
    b.set_options(decode_ac=1, histo_en=1)
    H.drive(b, data)
    if b.dib() is None:
        b.set_options(decode_ac=1)
        return {"preview": False}
    r = {"preview": True, "dib": H.hash_bytes(b.dib()), "stats": stats_record(b)}
    b.set_preview_mode(6)
    b.set_preview_ycc_offset(1, 1, 500, -200, 100)
    r["dib_rerender"] = H.hash_bytes(b.dib())
    r["stats_rerender"] = stats_record(b)
    b.set_preview_ycc_offset(0, 0, 0, 0, 0)
    b.set_preview_mode(1)
    b.set_options(decode_ac=1)
    return r


def log_record(b, data):
    """The text DecodeScanImg(nStart, true, false) writes to CDocLog (shim: 'W:' / 'E:' mark AddLineWarn / AddLineErr),
    with the histogram path off and on (the second adds the colour-statistics lines and the YCC clip warnings)."""
    out = {}
    for key, histo in (("plain", 0), ("histo", 1)):
        b.set_options(decode_ac=1, histo_en=histo)
        H.drive(b, data, quiet=0)
        out[key] = b.log_lines()
    b.set_options(decode_ac=0)
    H.drive(b, data, quiet=0)
    out["dc_only"] = b.log_lines()
    b.set_options(decode_ac=1)
    H.drive(b, data, quiet=1)
    out["quiet"] = b.log_lines()
    return out


def corrupt(data, mode, rng):
    d = bytearray(data)
    p = H.parse_jpeg(data)
    s, e = p.scan_start, p.scan_end
    if mode == "flip":
        for _ in range(3):
            d[int(rng.integers(s, e))] = int(rng.integers(0, 256))
    elif mode == "trunc":
        d = d[: int(rng.integers(s + 1, e))]
    elif mode == "marker":
        i = int(rng.integers(s, e)); d[i:i] = bytes([0xFF, 0xC4])
    elif mode == "ffff":
        i = int(rng.integers(s, e)); d[i:i] = bytes([0xFF, 0xFF, 0xFF])
    elif mode == "delete":
        i = int(rng.integers(s, e - 4)); del d[i:i + 2]
    elif mode == "stray_rst":
        i = int(rng.integers(s, e)); d[i:i] = bytes([0xFF, 0xD3])
    return bytes(d)


def header_variant(data, kind):
    """Valid scan, hostile header: the mutation is written into the file's own marker segments (SOF0, DRI, SOS, DHT)."""
    d = bytearray(data)
    sof = d.find(b"\xff\xc0"); sos = d.find(b"\xff\xda"); dri = d.find(b"\xff\xdd"); dht = d.find(b"\xff\xc4")
    nf = d[sof + 9]
    if kind == "samp":                                           # H that does not divide Hmax, V > Vmax of the coded data
        for c, hv in zip(range(nf), (0x32, 0x21, 0x13)): d[sof + 10 + 3 * c + 1] = hv
    elif kind == "dims":
        d[sof + 5:sof + 7] = (d[sof + 5] * 256 + d[sof + 6] + 37).to_bytes(2, "big"); d[sof + 7:sof + 9] = max(9, (d[sof + 7] * 256 + d[sof + 8]) // 2 + 3).to_bytes(2, "big")
    elif kind == "prec":
        d[sof + 4] = 12
    elif kind == "dri":
        assert dri > 0; d[dri + 4:dri + 6] = (4).to_bytes(2, "big")
    elif kind == "sel" and nf == 3:
        ns = d[sos + 4]
        for c in range(ns): d[sos + 5 + 2 * c + 1] = (0x10, 0x01, 0x00)[c % 3]
    elif kind == "dcsym":                                        # a DC symbol with a run nibble: outside the parallel path's table form
        d[dht + 5 + 16 + 5] = 0xAF
    elif kind == "acsym":
        q = d.find(b"\xff\xc4", dht + 4); d[q + 5 + 16 + 9] = 0x00    # an AC code turned into EOB
    return bytes(d)


def main():
    H.build(["oracle", "synth", "ref"])
    assert H.have_ref(), "needs the compiled reference (oracle/_ref)"
    rng = np.random.default_rng(2024)
    cases = {
        "c1_444_160x120": dict(width=160, height=120, hs=1, vs=1, seed=1),
        "420_odd_141x93": dict(width=141, height=93, seed=2),
        "420_q50_128x128": dict(width=128, height=128, quality=50, seed=3),
        "420_q95_opt_96x96": dict(width=96, height=96, quality=95, optimize_huffman=1, seed=4),
        "422_rst_row_160x64": dict(width=160, height=64, hs=2, vs=1, restart_interval=10, seed=5),
        "420_rst1_64x48": dict(width=64, height=48, restart_interval=1, quality=30, seed=6),
        "gray_101x77": dict(width=101, height=77, gray=1, seed=7),
        "440_96x80": dict(width=96, height=80, hs=1, vs=2, seed=8),
        "420_flat_64x64": dict(width=64, height=64, noise_sigma=0, quality=10, seed=9),
        "420_ff_dense_128x96": dict(width=128, height=96, quality=100, noise_sigma=40, seed=10),
    }
    manifest = {"generator": "tests/golden/make_golden.py", "reference": "JPEGsnoop 1.8.0 source compiled in place (oracle/Makefile ref)", "cases": {}}
    ref = H.ref_backend()
    files = {}
    for name, kw in cases.items():
        files[name] = H.synth_jpeg(**kw)
    base_for_corrupt = ["420_odd_141x93", "422_rst_row_160x64", "c1_444_160x120", "gray_101x77"]
    for bi, bname in enumerate(base_for_corrupt):
        for mode in ("flip", "trunc", "marker", "ffff", "delete", "stray_rst"):
            files[f"bad_{mode}_{bname}"] = corrupt(files[bname], mode, rng)
    # restart bookkeeping messages on scans that decode fine: wrong RSTn numbering; a DRI that announces a shorter
    # interval than the stream uses (markers are honoured where they are, "Restart marker not detected" where they are not)
    d = bytearray(files["422_rst_row_160x64"])
    pj = H.parse_jpeg(bytes(d)); k = 0
    for i in range(pj.scan_start, len(d) - 1):
        if d[i] == 0xFF and 0xD0 <= d[i + 1] <= 0xD7:
            k += 1
            if k in (2, 5): d[i + 1] = 0xD0 + ((d[i + 1] - 0xD0 + 3) & 7)
    files["rstnum_422_rst_row_160x64"] = bytes(d)
    d = bytearray(files["422_rst_row_160x64"])
    i = d.find(b"\xff\xdd\x00\x04")
    assert i > 0 and d[i + 4:i + 6] == b"\x00\x0a"
    d[i + 5] = 4
    files["dri4_422_rst_row_160x64"] = bytes(d)
    for kind, bname in (("samp", "420_q50_128x128"), ("dims", "422_rst_row_160x64"), ("prec", "c1_444_160x120"), ("dri", "422_rst_row_160x64"),
                        ("sel", "420_odd_141x93"), ("dcsym", "420_q50_128x128"), ("acsym", "c1_444_160x120")):
        files[f"hdr_{kind}_{bname}"] = header_variant(files[bname], kind)
    for name, data in files.items():
        with open(os.path.join(HERE, name + ".jpg"), "wb") as f:
            f.write(data)
        entry = {"bytes": len(data), "sha256": H.hash_bytes(data)}
        for mode, ac in (("full_idct", 1), ("dc_only", 0)):
            ref.set_options(decode_ac=ac)
            H.drive(ref, data)
            entry[mode] = record(ref)
        ref.set_options(decode_ac=1)
        entry["histo_en"] = histo_record(ref, data)
        entry["log"] = log_record(ref, data)
        H.drive(ref, data)                                     # Export to TIFF of the plain Full-IDCT decode (FileTiff.cpp:436)
        if ref.dib() is not None:
            ycc_ok = ref.planes()[1] is not None                   # the reference's handler dereferences the chroma planes
            entry["tiff"] = {k: H.hash_bytes(ref.export_tiff(m)) if (m != 2 or ycc_ok) else None for k, m in (("rgb8", 0), ("rgb16", 1), ("ycc8", 2))}
        manifest["cases"][name] = entry
    # known-answer values of the two fp32 stages, straight from the compiled reference
    lut = ref.idct_lut()
    manifest["kat"] = {
        "idct_lut_sha256": H.hash_bytes(lut),
        "idct_lut_fnv1a64": "%016x" % H.fnv1a64(lut.tobytes()),
        "idct_lut_spot": {"[0][0]": float(lut[0, 0]).hex(), "[1][8]": float(lut[1, 8]).hex(), "[9][9]": float(lut[9, 9]).hex(), "[63][63]": float(lut[63, 63]).hex()},
    }
    ref.lib.jsref_color_exhaustive_fnv.restype = __import__("ctypes").c_uint64
    manifest["kat"]["color_exhaustive_fnv1a64"] = "%016x" % ref.lib.jsref_color_exhaustive_fnv(__import__("ctypes").c_void_p(ref.h))
    blocks = []
    for i in range(64):
        c = np.zeros(64, np.int16)
        nz = int(rng.integers(1, 40))
        idx = rng.choice(63, nz, replace=False) + 1
        c[idx] = rng.integers(-1500, 1500, nz)
        blocks.append({"coef": [int(v) for v in c], "out_sha256": H.hash_bytes(ref.idct_block(c))})
    manifest["kat"]["idct_blocks"] = blocks
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote", len(files), "fixtures,", sum(len(v) for v in files.values()), "bytes of JPEG")


if __name__ == "__main__":
    main()
