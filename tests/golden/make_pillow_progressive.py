#!/usr/bin/env python3
"""Generates the third-party progressive fixtures under tests/golden/pillow/ (run here, where Pillow is installed; the files
are committed so the GPU box -- and any machine without Pillow -- only reads them).

For every case libjpeg-turbo (through Pillow) encodes the SAME picture twice with the same quality and sub-sampling: once as a
baseline sequential file (the oracle can decode it) and once as a progressive file with libjpeg's default scan script
(jpeg_simple_progression: spectral selection + successive approximation).  Both hold the same quantised coefficients -- the
forward DCT and the quantiser do not depend on the entropy coder -- so the progressive decode must give the baseline's DIB.
That pins the progressive decoder (jsnoop_decode_progressive) to an encoder that is not this repository's own."""
import io
import json
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "pillow")

CASES = [
    dict(name="p420_96x64", w=96, h=64, sub="4:2:0", q=85),
    dict(name="p422_80x48", w=80, h=48, sub="4:2:2", q=75),
    dict(name="p444_64x64", w=64, h=64, sub="4:4:4", q=92),
    dict(name="p420_odd_77x45", w=77, h=45, sub="4:2:0", q=60),
    dict(name="pgray_72x40", w=72, h=40, sub=None, q=80),
    dict(name="p420_rst_128x96", w=128, h=96, sub="4:2:0", q=85, restart_rows=1),
    dict(name="p422_rstblk_112x64", w=112, h=64, sub="4:2:2", q=50, restart_blocks=3),
    dict(name="p420_q100_64x48", w=64, h=48, sub="4:2:0", q=100),
    dict(name="p422_rst_1920x1080", w=1920, h=1080, sub="4:2:2", q=80, restart_rows=1, noise=6),   # BASELINE config 5's shape, written by libjpeg-turbo
    dict(name="p422_nodri_1920x1080", w=1920, h=1080, sub="4:2:2", q=80, noise=6),                  # ... and as most progressive files in the wild come: no restart markers
]


def picture(w, h, gray, seed, noise=14):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    chans = []
    for c in range(1 if gray else 3):
        f = 128 + 90 * np.sin(xx / (7.0 + 3 * c) + c) * np.cos(yy / (5.0 + 2 * c)) + rng.normal(0, noise, (h, w))
        chans.append(np.clip(f, 0, 255).astype(np.uint8))
    return Image.fromarray(chans[0], "L") if gray else Image.fromarray(np.dstack(chans), "RGB")


def main():
    os.makedirs(OUT, exist_ok=True)
    manifest = []
    for i, c in enumerate(CASES):
        im = picture(c["w"], c["h"], c["sub"] is None, 1000 + i, c.get("noise", 14))
        kw = dict(quality=c["q"])
        if c["sub"] is not None:
            kw["subsampling"] = c["sub"]
        if "restart_rows" in c:
            kw["restart_marker_rows"] = c["restart_rows"]
        if "restart_blocks" in c:
            kw["restart_marker_blocks"] = c["restart_blocks"]
        files = {}
        for kind, extra in (("base", dict(progressive=False, optimize=False)), ("prog", dict(progressive=True))):
            buf = io.BytesIO()
            im.save(buf, "JPEG", **kw, **extra)
            data = buf.getvalue()
            files[kind] = data
            open(os.path.join(OUT, f"{c['name']}_{kind}.jpg"), "wb").write(data)
        a = np.asarray(Image.open(io.BytesIO(files["base"])).convert("RGB"))
        b = np.asarray(Image.open(io.BytesIO(files["prog"])).convert("RGB"))
        assert np.array_equal(a, b), c["name"]           # libjpeg itself decodes both forms to the same pixels
        manifest.append(dict(c, base_bytes=len(files["base"]), prog_bytes=len(files["prog"])))
    json.dump(dict(pillow=Image.__version__ if hasattr(Image, "__version__") else "", cases=manifest), open(os.path.join(OUT, "manifest.json"), "w"), indent=1)
    print("wrote", len(CASES), "case pairs to", OUT)


if __name__ == "__main__":
    main()
