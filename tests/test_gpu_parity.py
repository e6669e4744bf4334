"""-m gpu: the HIP path against the oracle (and the compiled reference when its .so travelled),
through the C ABI, on seeded synthetic JPEGs.  Bit-exact for every output (integer / byte work;
the fp32 IDCT and colour stages are bit-exact by construction of the summation order)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [
    dict(width=640, height=480, hs=1, vs=1),                      # BASELINE config 1 geometry
    dict(width=333, height=217),                                   # odd size 4:2:0 -> 336x224
    dict(width=333, height=217, gray=1),
    dict(width=640, height=360, hs=2, vs=1, restart_interval=40),  # 4:2:2 with RSTn every MCU row
    dict(width=256, height=256, optimize_huffman=1, quality=95),
    dict(width=200, height=100, quality=20, restart_interval=1),
    dict(width=1920, height=1080),                                 # config 3 geometry (DIB 1920x1088)
]


def compare(H, a, b, side=True):
    da, db = a.dib(), b.dib()
    assert (da is None) == (db is None)
    if da is None:
        return
    assert a.image_size() == b.image_size()
    assert np.array_equal(da, db), "DIB differs"
    for pa, pb in zip(a.planes(), b.planes()):
        if pa is not None:
            assert np.array_equal(pa, pb), "int16 plane differs"
    if side:
        assert np.array_equal(a.mcu_map(), b.mcu_map()), "MCU file map differs"
        for pa, pb in zip(a.blk_dc(), b.blk_dc()):
            if pa is not None:
                assert np.array_equal(pa, pb), "block DC map differs"
        assert np.array_equal(a.dht_histo(), b.dht_histo()), "Huffman histogram differs"
        assert a.status() == b.status()
        assert a.bright_avg() == b.bright_avg()


@pytest.mark.parametrize("kw", CASES, ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items()))
def test_single_image_parity(harness, oracle, gpu, kw):
    data = harness.synth_jpeg(seed=11, **kw)
    harness.drive(oracle, data)
    harness.drive(gpu, data)
    assert gpu.lib.jsnoop_last_path(gpu.h) == 1 and gpu.lib.jsnoop_last_flags(gpu.h) == 0, "well-formed scan must take the parallel path"
    compare(harness, oracle, gpu)
    if harness.have_ref():
        r = harness.ref_backend()
        harness.drive(r, data)
        compare(harness, r, gpu)
        r.close()


def test_idct_lut_and_probe(harness, oracle, gpu):
    assert np.array_equal(oracle.idct_lut().view(np.uint32), gpu.idct_lut().view(np.uint32))
    rng = np.random.default_rng(5)
    for _ in range(200):
        c = np.zeros(64, np.int16)
        nz = rng.integers(1, 64)
        idx = rng.choice(64, nz, replace=False)
        c[idx] = rng.integers(-2000, 2000, nz)
        assert np.array_equal(oracle.idct_block(c).view(np.uint32), gpu.idct_block(c).view(np.uint32))


def test_dc_only_mode(harness, oracle, gpu):
    data = harness.synth_jpeg(width=320, height=240, seed=3)
    for b in (oracle, gpu):
        b.set_options(decode_ac=0)
        harness.drive(b, data)
    try:
        compare(harness, oracle, gpu)
    finally:
        for b in (oracle, gpu):
            b.set_options(decode_ac=1)


def test_corrupt_streams_exact_path(harness, oracle, gpu):
    """Malformed scans take the sequential exact-mirror device kernel and must still match the reference semantics."""
    rng = np.random.default_rng(7)
    base = [harness.synth_jpeg(width=96, height=64, seed=s, **kw) for s, kw in enumerate(
        [dict(), dict(hs=1, vs=1), dict(hs=2, vs=1, restart_interval=3), dict(gray=1), dict(restart_interval=1, quality=30)])]
    paths = {}
    for it in range(120):
        d = bytearray(base[it % len(base)])
        p = harness.parse_jpeg(bytes(d))
        s, e = p.scan_start, p.scan_end
        mode = it % 6
        if mode == 0:
            d[rng.integers(s, e)] = rng.integers(0, 256)
        elif mode == 1:
            d = d[: rng.integers(s + 1, len(d))]
        elif mode == 2:
            i = rng.integers(s, e); d[i:i] = bytes([0xFF, int(rng.integers(1, 256))])
        elif mode == 3:
            i = rng.integers(s, e); d[i:i] = bytes([0xFF] * int(rng.integers(2, 5)))
        elif mode == 4:
            i = rng.integers(s, e - 4); del d[i:i + int(rng.integers(1, 4))]
        else:
            i = rng.integers(s, e); d[i:i] = bytes([0xFF, 0xD0 + int(rng.integers(0, 8))])
        d = bytes(d)
        harness.drive(oracle, d, p)
        harness.drive(gpu, d, p)
        paths[gpu.lib.jsnoop_last_path(gpu.h)] = paths.get(gpu.lib.jsnoop_last_path(gpu.h), 0) + 1
        compare(harness, oracle, gpu)
    assert paths.get(2, 0) > 0, "some malformed scans must have been routed to the exact-mirror kernel"


@pytest.mark.parametrize("force_exact,sub_wl", [(False, 5), (False, 7), (True, 5)])
def test_batch_api(harness, oracle, force_exact, sub_wl):
    import jpegsnoop_amd as J
    kws = [dict(width=320, height=240), dict(width=333, height=217, hs=1, vs=1), dict(width=160, height=120, gray=1),
           dict(width=640, height=360, hs=2, vs=1, restart_interval=40), dict(width=1280, height=720, quality=92)]
    files = [harness.synth_jpeg(seed=20 + i, **kw) for i, kw in enumerate(kws)]
    b = J.JpegBatch(want_planes=True, force_exact=force_exact)
    b.set_tuning(sub_wl=sub_wl)                              # 128-byte or 512-byte sub-sequences
    for f in files:
        b.add_jpeg(f)
    b.tile(10)
    b.upload(); b.decode(); b.sync()
    sums = b.dib_checksums()
    assert all(b.info(i)['path'] == (2 if force_exact else 1) for i in range(10))
    for i in range(10):
        harness.drive(oracle, files[i % 5])
        assert np.array_equal(b.dib(i), oracle.dib())
        for pa, pb in zip(oracle.planes(), b.planes(i)):
            if pa is not None:
                assert np.array_equal(pa, pb)
        assert int(sums[i]) == J.dib_checksum_numpy(oracle.dib())
        assert np.array_equal(b.coefs(i), harness.oracle_coefs(oracle))
    b.close()


def test_golden_fixtures(harness, gpu):
    """The HIP path against the committed outputs of the compiled reference (tests/golden/manifest.json)."""
    from golden_util import load_case, manifest, record
    M = manifest()
    for name in sorted(M["cases"]):
        data = load_case(name)
        for mode, ac in (("full_idct", 1), ("dc_only", 0)):
            gpu.set_options(decode_ac=ac)
            try:
                harness.drive(gpu, data)
                assert record(harness, gpu) == M["cases"][name][mode], f"{name} [{mode}]"
            finally:
                gpu.set_options(decode_ac=1)
    for blk in M["kat"]["idct_blocks"]:
        assert harness.hash_bytes(gpu.idct_block(np.array(blk["coef"], np.int16))) == blk["out_sha256"]
    assert harness.hash_bytes(gpu.idct_lut()) == M["kat"]["idct_lut_sha256"]


def test_parallel_side_outputs(harness, oracle, gpu):
    """MCU file map, block-DC maps, code-length histogram and status words of images the parallel path decoded come from
    the parallel side pass (no sequential kernel): restart intervals of every length (byte-aligned interval ends
    included), what follows the scan (EOI, nothing, trailing bytes, a second marker), tiny and single-MCU images."""
    cases = []
    for i, kw in enumerate([dict(width=320, height=240, restart_interval=1), dict(width=320, height=240, restart_interval=3, hs=2, vs=1),
                            dict(width=333, height=217, restart_interval=7, hs=1, vs=1), dict(width=161, height=97, gray=1, restart_interval=2),
                            dict(width=16, height=16), dict(width=8, height=8, hs=1, vs=1), dict(width=40, height=24, restart_interval=1, quality=5, noise_sigma=0),
                            dict(width=640, height=480, quality=97, optimize_huffman=1), dict(width=96, height=80, hs=1, vs=2, restart_interval=5)]):
        cases.append(harness.synth_jpeg(seed=40 + i, **kw))
    base = cases[1]
    assert base[-2:] == b"\xff\xd9"
    cases += [base[:-2], base[:-1], base + b"\x00" * 7, base + b"\xff\xd9\xff\xe1\x00\x04ab", base[:-2] + b"\xff\xc4\x00\x02", base + b"\xff" * 5]
    for k, data in enumerate(cases):
        harness.drive(oracle, data)
        harness.drive(gpu, data)
        assert gpu.lib.jsnoop_last_path(gpu.h) == 1 and gpu.lib.jsnoop_last_flags(gpu.h) == 0, k
        try:
            compare(harness, oracle, gpu)
        except AssertionError as e:
            raise AssertionError(f"case {k}: {e}")


PROGRESSIVE = [
    dict(width=1920, height=1080, hs=2, vs=1, restart_interval=120),   # BASELINE config 5: 4:2:2, RSTn every MCU row
    dict(width=320, height=240),                                        # 4:2:0, no restart markers (one lane per scan)
    dict(width=160, height=96, hs=1, vs=1, restart_interval=7),
    dict(width=128, height=64, gray=1, restart_interval=3),
    dict(width=96, height=80, hs=1, vs=2, restart_interval=1, quality=30),
    dict(width=333, height=217, restart_interval=5),                    # not a multiple of the MCU: compare the visible region
    dict(width=141, height=93, hs=2, vs=1, optimize_huffman=1),
    dict(width=256, height=128, quality=98, restart_interval=4),       # dense blocks: refinement stretches with more than 32 correction bits
    dict(width=192, height=128, hs=1, vs=1, quality=100),               # all-ones quantisers, no restart markers
]


@pytest.mark.parametrize("mode", [1, 2], ids=["spectral", "successive"])
@pytest.mark.parametrize("kw", PROGRESSIVE, ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items()))
def test_progressive_transitive_parity(harness, oracle, gpu, kw, mode):
    """BASELINE config 5.  The reference refuses SOF2, so parity is transitive: the generator writes the
    same quantised coefficients once as a baseline file -- decoded by the oracle -- and once as a progressive multi-scan
    file with RSTn; the progressive decode must produce the baseline DIB and planes.  Blocks that lie wholly outside the
    picture are not coded by non-interleaved scans (T.81 A.2.3), so for sizes that are not MCU multiples the comparison
    covers the visible region.  mode 1: spectral selection only (DC scan + two AC bands per component); mode 2: spectral
    selection and successive approximation (DC and AC first scans with point transform, two refinement levels for
    luminance AC, DC refinement) -- a script in the style of the IJG default.  Both ends of the chain are pinned to libjpeg in
    tests/test_progressive_pillow.py: the generator's progressive output (libjpeg must decode it to the baseline form's pixels) and
    the decoder (progressive files written by libjpeg-turbo must give the oracle's DIB of libjpeg-turbo's baseline encoding)."""
    base = harness.synth_jpeg(seed=61, progressive=0, **kw)
    prog = harness.synth_jpeg(seed=61, progressive=mode, **kw)
    harness.drive(oracle, base)
    ncomp = 1 if kw.get("gray") else 3
    want_scans = 1 + 2 * ncomp if mode == 1 else (6 if ncomp == 1 else 10)
    assert gpu.decode_progressive(prog) == want_scans, gpu.lib.jsnoop_last_error()
    assert gpu.lib.jsnoop_last_path(gpu.h) == 3 and gpu.lib.jsnoop_last_flags(gpu.h) == 0
    assert gpu.image_size() == oracle.image_size()
    a, b = oracle.dib(), gpu.dib()
    H, W = kw["height"], kw["width"]
    assert np.array_equal(a[a.shape[0] - H:, :W], b[b.shape[0] - H:, :W]), "visible DIB differs"
    for pa, pb in zip(oracle.planes(), gpu.planes()):
        if pa is not None:
            assert np.array_equal(pa[:H, :W], pb[:H, :W]), "visible planes differ"
    mcu_w = 8 if kw.get("gray") else 8 * kw.get("hs", 2); mcu_h = 8 if kw.get("gray") else 8 * kw.get("vs", 2)
    if W % mcu_w == 0 and H % mcu_h == 0:
        assert np.array_equal(a, b), "DIB differs"
        assert gpu.bright_avg() == oracle.bright_avg()
    # the drop-in entry points keep refusing the file like the reference does
    import ctypes as C
    start = C.c_uint(0)
    buf = (C.c_uint8 * len(prog)).from_buffer_copy(prog)
    assert gpu.lib.jsnoop_jfif_walk(gpu.h, C.cast(buf, C.c_void_p), len(prog), C.byref(start)) == -1


def test_progressive_random_scripts(harness, oracle, gpu):
    """Randomised transitive parity of the progressive path (tools/fuzz_progressive.py runs the long version): random size,
    sampling, quality, restart interval and scan script; visible DIB and planes must equal the baseline decode."""
    rng = np.random.default_rng(2024)
    for k in range(80):
        gray = int(rng.integers(6) == 0)
        hs, vs = (1, 1) if gray else [(1, 1), (2, 1), (1, 2), (2, 2)][int(rng.integers(4))]
        kw = dict(width=int(rng.integers(8, 500)), height=int(rng.integers(8, 360)), hs=hs, vs=vs, gray=gray,
                  quality=int(rng.choice([10, 30, 50, 75, 85, 95, 100])), restart_interval=int(rng.choice([0, 0, 1, 2, 7, 33, 200])),
                  seed=int(rng.integers(1 << 30)), optimize_huffman=int(rng.integers(2)))
        mode = int(rng.integers(1, 3))
        harness.drive(oracle, harness.synth_jpeg(progressive=0, **kw))
        assert gpu.decode_progressive(harness.synth_jpeg(progressive=mode, **kw)) > 0, (k, kw, gpu.lib.jsnoop_last_error())
        assert gpu.lib.jsnoop_last_flags(gpu.h) == 0, (k, kw)
        a, b = oracle.dib(), gpu.dib()
        H, W = kw["height"], kw["width"]
        assert a.shape == b.shape and np.array_equal(a[a.shape[0] - H:, :W], b[b.shape[0] - H:, :W]), f"case {k} {kw} mode {mode}: visible DIB differs"
        for pa, pb in zip(oracle.planes(), gpu.planes()):
            if pa is not None:
                assert np.array_equal(pa[:H, :W], pb[:H, :W]), f"case {k} {kw} mode {mode}: visible planes differ"


def test_tiff_export(harness, oracle, gpu):
    """Export to TIFF (RGB 8 / 16 bit, YCC 8 bit): file bytes against the compiled reference's FileTiff output (golden
    hashes) and against the oracle's writer on a larger image."""
    from golden_util import load_case, manifest
    M = manifest()
    for name in sorted(M["cases"]):
        if "tiff" not in M["cases"][name]:
            continue
        harness.drive(gpu, load_case(name))
        for key, mode in (("rgb8", 0), ("rgb16", 1), ("ycc8", 2)):
            want = M["cases"][name]["tiff"][key]
            got = gpu.export_tiff(mode)
            assert (got is None) == (want is None), (name, key)
            if want is not None:
                assert harness.hash_bytes(got) == want, (name, key)
    data = harness.synth_jpeg(width=1920, height=1080, seed=31)
    harness.drive(oracle, data); harness.drive(gpu, data)
    for mode in (0, 1, 2):
        assert gpu.export_tiff(mode) == oracle.export_tiff(mode), mode


def test_decode_log_text(harness, gpu):
    """The text DecodeScanImg writes to the log (messages of the decode loop, statistics report, YCC clip warnings),
    line for line against what the compiled reference wrote (tests/golden/manifest.json): Full IDCT, histogram path,
    DC-only and quiet mode; well-formed, corrupted and restart-bookkeeping cases."""
    from golden_util import load_case, manifest
    M = manifest()
    bad = []
    for name in sorted(M["cases"]):
        data = load_case(name)
        want = M["cases"][name]["log"]
        for key, opt, quiet in (("plain", dict(decode_ac=1), 0), ("histo", dict(decode_ac=1, histo_en=1), 0), ("dc_only", dict(decode_ac=0), 0), ("quiet", dict(decode_ac=1), 1)):
            gpu.set_options(**opt)
            harness.drive(gpu, data, quiet=quiet)
            got = gpu.log_lines()
            if got != want[key]:
                k = next((i for i, (a, b) in enumerate(zip(got, want[key])) if a != b), min(len(got), len(want[key])))
                bad.append(f"{name} [{key}] line {k}: got {got[k] if k < len(got) else None!r} want {want[key][k] if k < len(want[key]) else None!r} ({len(got)} vs {len(want[key])} lines)")
    gpu.set_options()
    assert not bad, "\n".join(bad[:25]) + f"\n... {len(bad)} mismatching logs"


def test_decode_log_text_with_y_histogram_dump(harness, gpu):
    """bDumpHistoY (one of the five CSnoopConfig fields that reach DecodeScanImg, :2730): with bHistoEn the log ends with
    ReportHistogramY's 256 lines of the 2048-bin Y histogram (:3740-3741, :3845-3868) -- whole log, line for line, against what the
    compiled reference wrote (tests/golden/histo_dump.json, tests/golden/make_histo_dump.py)."""
    import json, os
    from golden_util import load_case
    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "histo_dump.json")))
    try:
        for name, lines in want.items():
            gpu.set_options(decode_ac=1, histo_en=1)
            gpu.set_dump_histo_y(1)
            harness.drive(gpu, load_case(name), quiet=0)
            got = gpu.log_lines()
            assert len(lines) > 300 and got == lines, (name, next((i, a, b) for i, (a, b) in enumerate(zip(got + [None], lines + [None])) if a != b))
            gpu.set_dump_histo_y(0)
            harness.drive(gpu, load_case(name), quiet=0)
            assert not any("Y Histogram in DC" in l for l in gpu.log_lines())
    finally:
        gpu.set_dump_histo_y(0); gpu.set_options()


def test_histogram_path(harness, oracle, gpu):
    """bHistoEn / bStatClipEn colour statistics: the committed records of the compiled reference,
    then the oracle on fresh streams (incl. corrupted ones whose DC drift trips the YCC range checks and their
    10-warning budget), after the decode and after preview re-renders; and the batch form of the same statistics."""
    import jpegsnoop_amd as J
    from golden_util import histo_record, load_case, manifest
    M = manifest()
    for name in sorted(M["cases"]):
        assert histo_record(harness, gpu, load_case(name)) == M["cases"][name]["histo_en"], name

    def stats_equal(a, b):
        sa, sb = a.color_stats(), b.color_stats()
        return all((sa[k] == sb[k]) if k == "count" else np.array_equal(sa[k], sb[k]) for k in sa)

    rng = np.random.default_rng(123)
    files = [harness.synth_jpeg(width=333, height=217, seed=21), harness.synth_jpeg(width=200, height=120, hs=2, vs=1, restart_interval=5, seed=22),
             harness.synth_jpeg(width=97, height=61, gray=1, seed=23)]
    for base in list(files):
        p = harness.parse_jpeg(base)
        for _ in range(4):
            d = bytearray(base)
            for _ in range(int(rng.integers(1, 6))):
                d[int(rng.integers(p.scan_start, p.scan_end))] ^= 1 << int(rng.integers(0, 8))
            files.append(bytes(d))
    try:
        for data in files:
            for opt in (dict(histo_en=1), dict(stat_clip_en=1), dict(histo_en=1, stat_clip_en=1, decode_ac=0)):
                for b in (oracle, gpu):
                    b.set_options(**opt)
                    harness.drive(b, data)
                if oracle.dib() is None:
                    assert gpu.dib() is None
                    continue
                assert np.array_equal(oracle.dib(), gpu.dib()) and stats_equal(oracle, gpu), opt
                for b in (oracle, gpu):
                    b.set_preview_mode(4)
                    b.set_preview_ycc_offset(2, 1, 700, -250, 90)
                assert np.array_equal(oracle.dib(), gpu.dib()) and stats_equal(oracle, gpu), opt
                for b in (oracle, gpu):
                    b.set_preview_ycc_offset(0, 0, 0, 0, 0)
                    b.set_preview_mode(1)
    finally:
        for b in (oracle, gpu):
            b.set_options()
    # batch API: one fresh pass per image
    batch = J.JpegBatch(want_planes=True)
    for data in files[:6]:
        batch.add_jpeg(data)
    batch.upload(); batch.decode(); batch.sync()
    oracle.set_options(histo_en=1)
    try:
        for i, data in enumerate(files[:6]):
            harness.drive(oracle, data)
            st = oracle.color_stats()
            want = np.concatenate([st["histo"].view(np.uint32), np.array([st["count"]], np.uint32), st["clip"], st["rgb"].ravel(), st["yfull"]])
            assert np.array_equal(batch.color_stats(i), want), i
    finally:
        oracle.set_options()
        batch.close()


def test_color_sweep(harness, oracle, gpu):
    """All 2^24 clamped (Y, Cb, Cr) triples through the device colour conversion against the oracle's
    ConvertYCCtoRGBFastFloat (true IEEE division by 0.587f): pins the FMA-corrected reciprocal form used on the device."""
    import ctypes as C
    out = np.zeros(1 << 24, np.uint32)
    assert gpu.lib.jsnoop_color_sweep(gpu.h, out.ctypes.data_as(C.c_void_p)) == 0
    cb, cr = np.meshgrid(np.arange(-128, 128, dtype=np.int32), np.arange(-128, 128, dtype=np.int32), indexing="ij")
    for y in range(-128, 128):
        ycc = np.stack([np.full(cb.size, y * 8, np.int32), cb.ravel() * 8, cr.ravel() * 8], -1).copy()
        rgb = np.zeros((ycc.shape[0], 3), np.uint8)
        oracle.lib.orc_color_fast(ycc.ctypes.data_as(C.c_void_p), rgb.ctypes.data_as(C.c_void_p), C.c_size_t(ycc.shape[0]))
        want = rgb[:, 2].astype(np.uint32) | (rgb[:, 1].astype(np.uint32) << 8) | (rgb[:, 0].astype(np.uint32) << 16)
        got = out[(y + 128) << 16:(y + 129) << 16]
        assert np.array_equal(got, want), f"Y={y}: {int((got != want).sum())} triples differ"


def test_preview_modes_and_shift(harness, oracle, gpu):
    """SetPreviewMode / SetPreviewYccOffset re-render (reference :633-659): colour kernel only, on the retained data."""
    data = harness.synth_jpeg(width=160, height=96, seed=8)
    harness.drive(gpu, data)
    base = gpu.dib().copy()
    planes = oracle and None
    harness.drive(oracle, data)
    py, pcb, pcr = [p.astype(np.int64) for p in oracle.planes()]
    H8, W8 = py.shape

    def expect(mode, sy=0, scb=0, scr=0, smx=0, smy=0):
        import ctypes as C
        mi = (np.arange(H8)[:, None] // 16) * (W8 // 16) + (np.arange(W8)[None, :] // 16)
        sel = mi >= smy * (W8 // 16) + smx
        ycc = np.stack([py + sel * sy, pcb + sel * scb, pcr + sel * scr], -1).astype(np.int32).reshape(-1, 3)
        rgb = np.zeros((ycc.shape[0], 3), np.uint8)
        oracle.lib.orc_color_fast(ycc.ctypes.data_as(C.c_void_p), rgb.ctypes.data_as(C.c_void_p), C.c_size_t(ycc.shape[0]))
        fin = np.clip(ycc >> 3, -128, 127) + 128
        r, g, b = rgb[:, 0], rgb[:, 1], rgb[:, 2]
        fy, fcb, fcr = fin[:, 0], fin[:, 1], fin[:, 2]
        R, G, B = {1: (r, g, b), 2: (fcr, fy, fcb), 3: (r, r, r), 4: (g, g, g), 5: (b, b, b), 6: (fy, fy, fy), 7: (fcb, fcb, fcb), 8: (fcr, fcr, fcr)}[mode]
        out = np.zeros((H8, W8, 4), np.uint8)
        out[..., 0], out[..., 1], out[..., 2] = B.reshape(H8, W8), G.reshape(H8, W8), R.reshape(H8, W8)
        return out[::-1]

    assert np.array_equal(base, expect(1))
    for mode in range(2, 9):
        gpu.set_preview_mode(mode)
        assert np.array_equal(gpu.dib(), expect(mode)), mode
    gpu.set_preview_mode(1)
    gpu.set_preview_ycc_offset(3, 2, 160, -80, 40)
    assert np.array_equal(gpu.dib(), expect(1, 160, -80, 40, 3, 2))
    gpu.set_preview_ycc_offset(0, 0, 0, 0, 0)
    assert np.array_equal(gpu.dib(), base)


def test_header_variations(harness, oracle, gpu):
    """Sampling factors the encoder cannot produce (up to 4x4, self-overlapping replication): rewrite the SOF of a valid
    file.  The scan then decodes to garbage, deterministically -- the geometry / replication arithmetic of the back end and
    the exact-mirror kernel must still match the reference semantics bit for bit."""
    data = harness.synth_jpeg(width=160, height=96, seed=4)
    p = harness.parse_jpeg(data)
    for samp in ([(4, 1), (1, 1), (1, 1)], [(1, 4), (1, 1), (1, 1)], [(4, 2), (2, 1), (1, 2)], [(2, 2), (2, 1), (1, 1)],
                 [(1, 1), (2, 2), (2, 2)], [(3, 1), (1, 1), (1, 1)], [(2, 2), (2, 2), (2, 2)], [(4, 4), (1, 1), (2, 2)]):
        q = harness.parse_jpeg(data)
        q.comps = [(c[0], h, v, c[3]) for c, (h, v) in zip(p.comps, samp)]
        harness.drive(oracle, data, q)
        harness.drive(gpu, data, q)
        compare(harness, oracle, gpu)
    # random factors 1..4 for every component: H that does not divide Hmax (expansion truncates, part of the MCU stays zero) etc.
    rng = np.random.default_rng(3)
    for _ in range(60):
        f = [int(x) for x in rng.integers(1, 5, 6)]
        q = harness.parse_jpeg(data)
        q.comps = [(c[0], f[2 * i], f[2 * i + 1], c[3]) for i, c in enumerate(p.comps)]
        harness.drive(oracle, data, q)
        harness.drive(gpu, data, q)
        try:
            compare(harness, oracle, gpu)
        except AssertionError as e:
            raise AssertionError(f"sampling factors {f}: {e}")


def test_precision_divider(harness, oracle, gpu):
    """SOF precision other than 8 (reference :1234-1238: value /= 1 << (P - 8), truncating; nothing for P < 8), on the
    parallel path and -- with a corrupted scan -- on the exact-mirror path."""
    data = harness.synth_jpeg(width=160, height=96, seed=6, restart_interval=4)
    p = harness.parse_jpeg(data)
    bad = bytearray(data); bad[p.scan_start + 200] ^= 0x40; bad[p.scan_start + 900] ^= 0x08
    for prec in (12, 16, 9, 5):
        for stream in (data, bytes(bad)):
            q = harness.parse_jpeg(stream)
            q.precision = prec
            harness.drive(oracle, stream, q)
            harness.drive(gpu, stream, q)
            compare(harness, oracle, gpu)
        assert True


def test_error_limit_option(harness, oracle, gpu):
    """nErrMaxDecodeScan other than the default 20: warning counter, status words and pixels on the corrupted golden files."""
    from golden_util import load_case, manifest
    M = manifest()
    names = [n for n in sorted(M["cases"]) if n.startswith("bad_")]
    try:
        for em in (1, 3, 50):
            for n in names:
                data = load_case(n)
                for b in (oracle, gpu):
                    b.set_options(err_max=em)
                    harness.drive(b, data)
                compare(harness, oracle, gpu)
    finally:
        for b in (oracle, gpu):
            b.set_options()


def test_fuzz_headers_and_scans(harness, oracle, gpu):
    """tests/fuzz_util.py: 400 hostile variants (corrupted scan bytes, mutated dimensions / sampling factors / precision /
    restart interval / table selectors / scan start, one-component scans, damaged Huffman tables), every output of the
    decoder incl. the colour statistics on half of them, HIP path vs oracle.  (tools/fuzz_gpu.py runs larger sweeps.)"""
    import fuzz_util as F
    B = F.bases(harness)
    rng = np.random.default_rng(4242)
    try:
        for k in range(400):
            data, q, mode = F.mutate(harness, rng, B[int(rng.integers(len(B)))])
            histo = int(rng.integers(2))
            for b in (oracle, gpu):
                b.set_options(histo_en=histo)
            harness.drive(oracle, data, q)
            harness.drive(gpu, data, q)
            r = F.differs(oracle, gpu, stats=bool(histo))
            assert r is None, (k, mode, r)
    finally:
        for b in (oracle, gpu):
            b.set_options()


def test_jfif_front_end_variations(harness, oracle, gpu):
    """The minimal JFIF walk on header layouts the synthetic encoder does not emit: marker padding
    (FF fill bytes), APPn / COM segments in between, all DQT tables in one segment with 16-bit entries, all DHT tables in one
    segment, and a Motion-JPEG frame (APP0 'AVI1', no DHT: the standard tables are imported, JfifDecode.cpp:4405-4421)."""
    import ctypes as C
    import jpegsnoop_amd as J
    data = harness.synth_jpeg(width=160, height=96, seed=12)
    harness.drive(oracle, data)
    want = oracle.dib().copy()

    def segments(d):
        out, pos = [], 2
        while d[pos + 1] != 0xDA:
            ln = d[pos + 2] * 256 + d[pos + 3]
            out.append((d[pos + 1], bytes(d[pos + 4:pos + 2 + ln]))); pos += 2 + ln
        return out, bytes(d[pos:])

    def build(segs, tail, pad=b""):
        return b"\xff\xd8" + b"".join(pad + bytes([0xFF, m]) + (len(b) + 2).to_bytes(2, "big") + b for m, b in segs) + pad + tail

    segs, tail = segments(data)
    variants = {}
    variants["padding_and_comments"] = build([(0xFE, b"hello")] + segs[:2] + [(0xE1, b"Exif\0\0" + b"\0" * 20)] + segs[2:], tail, pad=b"\xff\xff")
    dqt = b"".join(bytes([0x10 | b[0]]) + b"".join(bytes([0, v]) for v in b[1:65]) for m, b in segs if m == 0xDB)     # Pq = 1: 16-bit entries
    dht = b"".join(b for m, b in segs if m == 0xC4)
    merged = [(m, b) for m, b in segs if m not in (0xDB, 0xC4)]
    merged.insert(1, (0xDB, dqt)); merged.append((0xC4, dht))
    variants["merged_tables_16bit_dqt"] = build(merged, tail)
    mj = [(m, (b"AVI1" + b"\0" * 10) if m == 0xE0 else b) for m, b in segs if m != 0xC4]
    variants["mjpeg_no_dht"] = build(mj, tail)
    for name, v in variants.items():
        start = C.c_uint(0)
        buf = (C.c_uint8 * len(v)).from_buffer_copy(v)
        assert gpu.lib.jsnoop_jfif_walk(gpu.h, C.cast(buf, C.c_void_p), len(v), C.byref(start)) == 0, (name, gpu.lib.jsnoop_last_error())
        gpu.decode_scan_img(C.cast(buf, C.c_void_p), len(v), start.value, 1, 1)
        assert np.array_equal(gpu.dib(), want), name
        b = J.JpegBatch(); b.add_jpeg(v); b.upload(); b.decode(); b.sync()
        assert np.array_equal(b.dib(0), want), name
        b.close()


def test_config2_single_4k(harness, oracle):
    """BASELINE config 2: one 3840x2160 4:2:0 image end to end through the parallel path."""
    import jpegsnoop_amd as J
    data = harness.synth_jpeg(width=3840, height=2160, seed=21)
    b = J.JpegBatch(want_planes=True)
    b.add_jpeg(data); b.upload(); b.decode(); b.sync()
    assert b.info(0)["path"] == 1 and b.info(0)["flags"] == 0
    harness.drive(oracle, data)
    assert np.array_equal(b.dib(0), oracle.dib())
    for pa, pb in zip(oracle.planes(), b.planes(0)):
        assert np.array_equal(pa, pb)
    b.close()


def test_config5_baseline_422_restart(harness, oracle):
    """The reference-decodable form of BASELINE config 5: 1920x1080 4:2:2 with a restart marker every MCU row
    (the reference refuses SOF2, source/JfifDecode.cpp:4827-4833, so its progressive form has no reference answer)."""
    import jpegsnoop_amd as J
    data = harness.synth_jpeg(width=1920, height=1080, hs=2, vs=1, restart_interval=120, seed=22)
    b = J.JpegBatch()
    b.add_jpeg(data); b.upload(); b.decode(); b.sync()
    assert b.info(0)["path"] == 1 and b.info(0)["flags"] == 0
    harness.drive(oracle, data)
    assert oracle.status()["restart_read"] == 134
    assert np.array_equal(b.dib(0), oracle.dib())
    b.close()
    # the same pixels without restart markers decode to the same DIB (RST invariance)
    plain = harness.synth_jpeg(width=1920, height=1080, hs=2, vs=1, restart_interval=0, seed=22)
    harness.drive(oracle, plain)
    b2 = J.JpegBatch(); b2.add_jpeg(plain); b2.upload(); b2.decode(); b2.sync()
    assert np.array_equal(b2.dib(0), oracle.dib())
    b2.close()
