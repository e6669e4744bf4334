"""Third-party pin of the progressive (SOF2) path -- BASELINE config 5 has no reference answer (the reference refuses SOF2,
source/JfifDecode.cpp:4827-4833), so parity is transitive; these tests anchor both ends of that chain to libjpeg (via Pillow):

* CPU: the repository's own progressive ENCODER (oracle/jpeg_synth.c, the generator behind the transitive-parity tests) writes
  files that libjpeg decodes to exactly the pixels of the baseline file with the same coefficients;
* GPU: the progressive DECODER reproduces, from progressive files written by libjpeg-turbo (committed fixtures,
  tests/golden/make_pillow_progressive.py), the DIB the oracle gives for libjpeg-turbo's baseline encoding of the same picture
  (same quality / sub-sampling => same quantised coefficients).
"""
import io
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
PIL_DIR = os.path.join(HERE, "golden", "pillow")
CASES = json.load(open(os.path.join(PIL_DIR, "manifest.json")))["cases"]

SYNTH = [
    dict(width=64, height=48),
    dict(width=160, height=96, hs=2, vs=1, restart_interval=10),
    dict(width=96, height=64, hs=1, vs=1, quality=50),
    dict(width=120, height=80, gray=1, restart_interval=3),
    dict(width=333, height=217, restart_interval=5),
    dict(width=141, height=93, hs=2, vs=1, optimize_huffman=1),
    dict(width=256, height=128, quality=98, restart_interval=4),
    dict(width=192, height=128, hs=1, vs=1, quality=100),
    dict(width=1920, height=1080, hs=2, vs=1, restart_interval=120),      # BASELINE config 5 itself
]


def pil_decode(data):
    from PIL import Image
    im = Image.open(io.BytesIO(data))
    im.load()
    return np.asarray(im), im


@pytest.mark.parametrize("kw", SYNTH, ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items()))
def test_own_progressive_encoder_against_libjpeg(harness, kw):
    pytest.importorskip("PIL")
    base, _ = pil_decode(harness.synth_jpeg(seed=61, progressive=0, **kw))
    for mode in (1, 2):
        data = harness.synth_jpeg(seed=61, progressive=mode, **kw)
        px, im = pil_decode(data)
        assert im.info.get("progressive") or im.info.get("progression"), "not a progressive file"
        assert px.shape == base.shape and np.array_equal(px, base), f"mode {mode}: libjpeg decodes different pixels than for the baseline form"


@pytest.mark.parametrize("c", CASES, ids=lambda c: c["name"])
def test_oracle_reads_the_pillow_baseline_fixtures(harness, oracle, c):
    data = open(os.path.join(PIL_DIR, c["name"] + "_base.jpg"), "rb").read()
    harness.drive(oracle, data)
    dib = oracle.dib()
    assert dib.shape[0] >= c["h"] and dib.shape[1] >= c["w"] and oracle.lib is not None
    try:                                                      # where Pillow is present: same picture up to IDCT / up-sampling differences
        vis = dib[dib.shape[0] - c["h"]:, :c["w"], :3][::-1, :, ::-1].astype(int)        # bottom-up BGRA -> top-down RGB
        px, im = pil_decode(data)
        ref = np.asarray(im.convert("RGB")).astype(int)
        assert np.abs(vis - ref).mean() < 4.0
    except ImportError:
        pass


@pytest.mark.gpu
@pytest.mark.parametrize("c", CASES, ids=lambda c: c["name"])
def test_pillow_progressive_files_decode_to_the_baseline_dib(harness, oracle, gpu, c):
    base = open(os.path.join(PIL_DIR, c["name"] + "_base.jpg"), "rb").read()
    prog = open(os.path.join(PIL_DIR, c["name"] + "_prog.jpg"), "rb").read()
    harness.drive(oracle, base)
    nsc = gpu.decode_progressive(prog)
    assert nsc > 1, gpu.lib.jsnoop_last_error()
    assert gpu.lib.jsnoop_last_path(gpu.h) == 3 and gpu.lib.jsnoop_last_flags(gpu.h) == 0
    assert gpu.image_size() == oracle.image_size()
    a, b = oracle.dib(), gpu.dib()
    H, W = c["h"], c["w"]
    assert np.array_equal(a[a.shape[0] - H:, :W], b[b.shape[0] - H:, :W]), "visible DIB differs from the oracle's decode of libjpeg's baseline file"
    for pa, pb in zip(oracle.planes(), gpu.planes()):
        if pa is not None:
            assert np.array_equal(pa[:H, :W], pb[:H, :W]), "visible planes differ"
