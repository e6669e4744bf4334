"""CPU: the oracle against the compiled reference itself (oracle/_ref/libjsnoop_ref.so), when available:
in this container it is built from /root/reference by `make -C oracle ref`.  Covers well-formed streams,
header variations (odd sampling factors) and a fuzz sweep over corrupted scans; every output byte-equal."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def ref(harness):
    if not harness.have_ref():
        pytest.skip("compiled reference not available (no /root/reference and no prebuilt oracle/_ref)")
    b = harness.ref_backend()
    yield b
    b.close()


def same(H, a, b):
    da, db = a.dib(), b.dib()
    if (da is None) != (db is None):
        return False
    if da is None:
        return a.image_size() == b.image_size()
    ok = np.array_equal(da, db) and a.image_size() == b.image_size()
    ok = ok and all(np.array_equal(x, y) for x, y in zip(a.planes(), b.planes()) if x is not None)
    ok = ok and np.array_equal(a.mcu_map(), b.mcu_map()) and np.array_equal(a.dht_histo(), b.dht_histo())
    ok = ok and all(np.array_equal(x, y) for x, y in zip(a.blk_dc(), b.blk_dc()) if x is not None)
    return ok and a.status() == b.status() and a.bright_avg() == b.bright_avg() and a.is_preview_ready() == b.is_preview_ready()


CASES = [dict(width=640, height=480, hs=1, vs=1), dict(width=333, height=217), dict(width=333, height=217, gray=1),
         dict(width=640, height=360, hs=2, vs=1, restart_interval=40), dict(width=256, height=256, optimize_huffman=1, quality=95),
         dict(width=200, height=100, quality=20, restart_interval=1), dict(width=96, height=80, hs=1, vs=2)]


@pytest.mark.parametrize("kw", CASES, ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items()))
def test_wellformed(harness, oracle, ref, kw):
    data = harness.synth_jpeg(seed=5, **kw)
    for ac in (1, 0):
        for b in (oracle, ref):
            b.set_options(decode_ac=ac)
            harness.drive(b, data)
        assert same(harness, ref, oracle)
    for b in (oracle, ref):
        b.set_options(decode_ac=1)


def test_table_kats(harness, oracle, ref):
    assert np.array_equal(ref.idct_lut().view(np.uint32), oracle.idct_lut().view(np.uint32))
    data = harness.synth_jpeg(width=64, height=64, optimize_huffman=1, seed=9)
    harness.drive(ref, data)
    harness.drive(oracle, data)
    assert np.array_equal(ref.lookupfast(), oracle.lookupfast())


def test_header_variations(harness, oracle, ref):
    """Sampling factors the synthetic encoder cannot produce: rewrite the SOF of a valid file.  The scan then
    decodes to garbage, but deterministically, and exercises the replication/geometry arithmetic."""
    data = harness.synth_jpeg(width=160, height=96, seed=4)
    p = harness.parse_jpeg(data)
    for samp in ([(4, 1), (1, 1), (1, 1)], [(1, 4), (1, 1), (1, 1)], [(4, 2), (2, 1), (1, 2)], [(2, 2), (2, 1), (1, 1)],
                 [(1, 1), (2, 2), (2, 2)], [(3, 1), (1, 1), (1, 1)], [(2, 2), (2, 2), (2, 2)], [(4, 4), (1, 1), (2, 2)]):
        q = harness.parse_jpeg(data)
        q.comps = [(c[0], h, v, c[3]) for c, (h, v) in zip(p.comps, samp)]
        harness.drive(ref, data, q)
        harness.drive(oracle, data, q)
        assert same(harness, ref, oracle), samp
    rng = np.random.default_rng(3)                                 # random factors 1..4 (non-dividing ones included)
    for _ in range(60):
        f = [int(x) for x in rng.integers(1, 5, 6)]
        q = harness.parse_jpeg(data)
        q.comps = [(c[0], f[2 * i], f[2 * i + 1], c[3]) for i, c in enumerate(p.comps)]
        harness.drive(ref, data, q)
        harness.drive(oracle, data, q)
        assert same(harness, ref, oracle), f


def test_fuzz_corrupt_scans(harness, oracle, ref):
    rng = np.random.default_rng(7)
    base = [harness.synth_jpeg(width=96, height=64, seed=s, **kw) for s, kw in enumerate(
        [dict(), dict(hs=1, vs=1), dict(hs=2, vs=1, restart_interval=3), dict(gray=1), dict(restart_interval=1, quality=30), dict(optimize_huffman=1)])]
    for it in range(400):
        d = bytearray(base[it % len(base)])
        p = harness.parse_jpeg(bytes(d))
        s, e = p.scan_start, p.scan_end
        mode = it % 8
        if mode == 0:
            for _ in range(int(rng.integers(1, 4))):
                d[int(rng.integers(s, e))] = int(rng.integers(0, 256))
        elif mode == 1:
            d = d[: int(rng.integers(s + 1, len(d)))]
        elif mode == 2:
            i = int(rng.integers(s, e)); d[i:i] = bytes([0xFF, int(rng.integers(1, 256))])
        elif mode == 3:
            i = int(rng.integers(s, e)); d[i:i] = bytes([0xFF] * int(rng.integers(2, 5)))
        elif mode == 4:
            i = int(rng.integers(s, e - 4)); del d[i:i + int(rng.integers(1, 4))]
        elif mode == 5:
            i = int(rng.integers(s, e)); d[i:i] = bytes([0xFF, 0xD0 + int(rng.integers(0, 8))])
        elif mode == 6:
            i = int(rng.integers(s, e)); d[i] ^= 1 << int(rng.integers(0, 8))
        else:
            i = int(rng.integers(s, e)); d[i:e] = bytes(e - i)
        d = bytes(d)
        harness.drive(ref, d, p)
        harness.drive(oracle, d, p)
        assert same(harness, ref, oracle), (it, mode)


def stats_equal(a, b):
    sa, sb = a.color_stats(), b.color_stats()
    return all((sa[k] == sb[k]) if k == "count" else np.array_equal(sa[k], sb[k]) for k in sa)


def test_histogram_path(harness, oracle, ref):
    """bHistoEn / bStatClipEn colour path: min/max/sum records, clip counters (the YCC ones stop at 10 warnings),
    the 128-bin RGB and 2048-bin Y histograms -- after the decode and after preview re-renders (the reference keeps
    accumulating) -- for well-formed and corrupted streams."""
    rng = np.random.default_rng(77)
    files = [harness.synth_jpeg(width=160, height=96, seed=3), harness.synth_jpeg(width=200, height=120, hs=2, vs=1, restart_interval=5, seed=4),
             harness.synth_jpeg(width=96, height=64, gray=1, seed=5)]
    for base in list(files):
        p = harness.parse_jpeg(base)
        for _ in range(6):
            d = bytearray(base)
            for _ in range(int(rng.integers(1, 6))):
                d[int(rng.integers(p.scan_start, p.scan_end))] ^= 1 << int(rng.integers(0, 8))
            files.append(bytes(d))
    try:
        for data in files:
            for opt in (dict(histo_en=1), dict(stat_clip_en=1), dict(histo_en=1, stat_clip_en=1, decode_ac=0)):
                for b in (oracle, ref):
                    b.set_options(**opt)
                    harness.drive(b, data)
                assert same(harness, ref, oracle) and stats_equal(ref, oracle), opt
                if ref.dib() is None:
                    continue
                for b in (oracle, ref):
                    b.set_preview_mode(4)
                    b.set_preview_ycc_offset(2, 1, 700, -250, 90)
                assert np.array_equal(ref.dib(), oracle.dib()) and stats_equal(ref, oracle), opt
                for b in (oracle, ref):
                    b.set_preview_ycc_offset(0, 0, 0, 0, 0)
                    b.set_preview_mode(1)
    finally:
        for b in (oracle, ref):
            b.set_options()


def test_precision_divider(harness, oracle, ref):
    """SOF precision other than 8: the reference divides every decoded value by 1 << (P - 8) (truncating, :1234-1238) and
    does nothing for P < 8."""
    data = harness.synth_jpeg(width=160, height=96, seed=6, restart_interval=4)
    for prec in (12, 16, 9, 5):
        q = harness.parse_jpeg(data)
        q.precision = prec
        for ac in (1, 0):
            for b in (oracle, ref):
                b.set_options(decode_ac=ac)
                harness.drive(b, data, q)
            assert same(harness, ref, oracle), (prec, ac)
    for b in (oracle, ref):
        b.set_options()


def test_error_limit_option(harness, oracle, ref):
    """nErrMaxDecodeScan (CSnoopConfig, read at :2732): the warning counter saturates there; decoding continues."""
    from golden_util import load_case, manifest
    M = manifest()
    names = [n for n in sorted(M["cases"]) if n.startswith("bad_")][:10]
    for em in (1, 3, 50):
        for n in names:
            data = load_case(n)
            for b in (oracle, ref):
                b.set_options(err_max=em)
                harness.drive(b, data)
            assert same(harness, ref, oracle), (em, n)
    for b in (oracle, ref):
        b.set_options()


def test_fuzz_headers_and_scans(harness, oracle, ref):
    """tests/fuzz_util.py: 500 hostile variants -- corrupted scan bytes and mutated header fields (dimensions, sampling
    factors, precision, restart interval, table selectors, one-component scans of a three-component frame, shifted scan
    start, damaged Huffman tables) -- oracle vs the compiled reference, every output."""
    import fuzz_util as F
    B = F.bases(harness)
    rng = np.random.default_rng(2025)
    for k in range(500):
        data, q, mode = F.mutate(harness, rng, B[int(rng.integers(len(B)))])
        harness.drive(ref, data, q)
        harness.drive(oracle, data, q)
        assert F.differs(ref, oracle) is None, (k, mode, F.differs(ref, oracle))
