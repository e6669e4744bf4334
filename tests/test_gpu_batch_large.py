"""GPU: batch shapes the small parity cases do not reach.

* a batch above the 96 MB threshold at which the library switches to 512-byte sub-sequences on its own (the production
  configuration of the 1024-image bench, jsnoop_host.cpp upload()): 192 x 1080p, every DIB checked against the oracle;
* one batch mixing MCU geometries up to 4 x 4 sampling (a 32 x 32 MCU needs more than 64 KiB of LDS per workgroup: the
  opt-in path of js_launch_idct_color) with ordinary images, through the general and the short colour path at once;
* two batches on two streams of one device driven from two host threads (the C ABI's per-thread device default).
"""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_batch_above_the_long_subsequence_threshold(harness, oracle):
    import jpegsnoop_amd as J
    files = [harness.synth_jpeg(width=1920, height=1080, hs=2, vs=2, quality=85, seed=500 + i) for i in range(8)]
    n = 192
    assert n * min(len(f) for f in files) >= 96 << 20, "batch too small to cross the threshold"
    b = J.JpegBatch()
    for f in files:
        b.add_jpeg(f)
    b.tile(n)
    b.upload(); b.decode(); b.sync()
    assert b.split_parts() == 2                                     # the library's default for a batch this size: two halves on two streams
    sums = b.dib_checksums()
    assert all(b.info(i)["path"] == 1 and b.info(i)["flags"] == 0 for i in range(n))
    for j, f in enumerate(files):
        harness.drive(oracle, f)
        want = J.dib_checksum_numpy(oracle.dib())
        assert all(int(sums[i]) == want for i in range(j, n, len(files))), f"picture {j}"
    assert np.array_equal(b.dib(n - 1), oracle.dib())            # one full DIB compared byte for byte as well
    b.decode(); b.sync()                                            # a second decode of the resident batch gives the same answer
    assert np.array_equal(b.dib_checksums(), sums)
    b.close()


def test_mixed_geometry_batch_with_4x4_sampling(harness, oracle):
    import jpegsnoop_amd as J
    kws = [dict(width=320, height=240), dict(width=200, height=136, hs=4, vs=4), dict(width=160, height=120, gray=1),
           dict(width=256, height=96, hs=4, vs=1), dict(width=333, height=217, hs=1, vs=1), dict(width=96, height=160, hs=1, vs=2),
           dict(width=192, height=128, hs=2, vs=1, restart_interval=7), dict(width=144, height=144, hs=2, vs=4)]
    files = [harness.synth_jpeg(seed=700 + i, **kw) for i, kw in enumerate(kws)]
    b = J.JpegBatch(want_planes=True)
    for f in files:
        b.add_jpeg(f)
    b.tile(2 * len(files))
    b.upload(); b.decode(); b.sync()
    for i in range(2 * len(files)):
        harness.drive(oracle, files[i % len(files)])
        assert b.info(i)["flags"] == 0, (i, kws[i % len(files)])
        assert np.array_equal(b.dib(i), oracle.dib()), (i, kws[i % len(files)])
        for pa, pb in zip(oracle.planes(), b.planes(i)):
            if pa is not None:
                assert np.array_equal(pa, pb), (i, kws[i % len(files)])
    b.close()


def test_two_batches_two_streams_two_host_threads(harness, oracle):
    import jpegsnoop_amd as J
    lib = J.load()
    sets = [[harness.synth_jpeg(width=640, height=480, seed=800 + i) for i in range(4)],
            [harness.synth_jpeg(width=512, height=384, hs=2, vs=1, seed=900 + i) for i in range(4)]]
    want = []
    for files in sets:
        w = []
        for f in files:
            harness.drive(oracle, f)
            w.append(J.dib_checksum_numpy(oracle.dib()))
        want.append(w)
    errors = []

    def worker(k):
        try:
            assert lib.jsnoop_set_device(0) == 0                 # per-thread default device
            b = J.JpegBatch()                                    # own non-blocking stream
            for f in sets[k]:
                b.add_jpeg(f)
            b.tile(64)
            for _ in range(5):
                b.upload(); b.decode(); b.sync()
                sums = b.dib_checksums()
                for i in range(64):
                    if int(sums[i]) != want[k][i % 4] or b.info(i)["flags"]:
                        errors.append((k, i))
            b.close()
        except Exception as e:                                   # surfaced by the main thread
            errors.append((k, repr(e)))

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:5]


def _device_workers(harness, oracle, devices):
    """One host thread per entry of `devices`: jsnoop_set_device(dev), an own batch holding an ordinary image AND a 4 x 4 sampled one
    (32 x 32 MCU: the back end's LDS tile needs the > 64 KiB opt-in, which is a per-device attribute), decoded three times."""
    import jpegsnoop_amd as J
    lib = J.load()
    kws = [dict(width=640, height=480), dict(width=200, height=136, hs=4, vs=4), dict(width=512, height=384, hs=2, vs=1)]
    files = [harness.synth_jpeg(seed=1300 + i, **kw) for i, kw in enumerate(kws)]
    want = []
    for f in files:
        harness.drive(oracle, f)
        want.append(J.dib_checksum_numpy(oracle.dib()))
    errors = []
    start = threading.Barrier(len(devices))

    def worker(k, dev):
        try:
            assert lib.jsnoop_set_device(dev) == 0, J.last_error()   # per-thread default device
            b = J.JpegBatch()
            for f in files:
                b.add_jpeg(f)
            b.tile(48)
            start.wait(timeout=120)                                  # all threads launch their first back end together
            for _ in range(3):
                b.upload(); b.decode(); b.sync()
                sums = b.dib_checksums()
                for i in range(48):
                    if int(sums[i]) != want[i % 3] or b.info(i)["flags"]:
                        errors.append((k, dev, i))
            b.close()
        except Exception as e:                                       # surfaced by the main thread
            errors.append((k, dev, repr(e)))

    ts = [threading.Thread(target=worker, args=(k, d)) for k, d in enumerate(devices)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:5]


def test_one_process_threads_share_a_device_with_large_tiles(harness, oracle):
    _device_workers(harness, oracle, [0, 0, 0])


def test_one_process_one_host_thread_per_device(harness, oracle):
    """The one-process N-device entry of SURVEY.md 8(e): N host threads, jsnoop_set_device(k) each.  Needs two visible devices."""
    import jpegsnoop_amd as J
    n = J.load().jsnoop_device_count()
    if n < 2:
        pytest.skip(f"{n} HIP device(s) visible: the multi-device test needs two")
    _device_workers(harness, oracle, list(range(min(n, 8))))


def test_staging_pipeline_overlaps_and_stays_exact(harness, oracle):
    """Two slots cycled by jsnoop_pipeline_run (H2D of the next batch while the current one decodes, D2H of the previous one on
    request): every slot's DIBs equal the oracle's afterwards, and the overlapped time per batch is below the sum of its pieces."""
    import jpegsnoop_amd as J
    files = [harness.synth_jpeg(width=1280, height=720, seed=300 + i) for i in range(4)]
    want = []
    for f in files:
        harness.drive(oracle, f)
        want.append(J.dib_checksum_numpy(oracle.dib()))
    pipe = J.JpegPipeline(2)
    for b in pipe.slots:
        for f in files:
            b.add_jpeg(f)
        b.tile(128)
    r2 = pipe.run(6, d2h=False)
    r3 = pipe.run(5, d2h=True)
    for b in pipe.slots:
        sums = b.dib_checksums()
        assert all(int(sums[i]) == want[i % 4] for i in range(128))
        assert all(b.info(i)["flags"] == 0 for i in range(128))
    assert r2["h2d_ms"] > 0 and r2["decode_ms"] > 0 and r3["d2h_ms"] > 0
    assert r2["ms_per_batch"] < 0.95 * (r2["h2d_ms"] + r2["decode_ms"]), r2          # the transfer hides behind the decode (or the other way round)
    # with the read-back the PCIe D2H of 472 MB bounds a batch (8.3 of the 10.5 ms the three pieces take one after the other): what can hide is small,
    # and the measured ratio sits at 0.94-0.95 for three batches -- five batches and a bound that a run without any overlap (ratio >= 1) still fails
    assert r3["ms_per_batch"] < 0.98 * (r3["h2d_ms"] + r3["decode_ms"] + r3["d2h_ms"]), r3
    pipe.close()


def test_progressive_batch_matches_single_decodes_and_is_faster(harness, oracle):
    """BASELINE config 5 as a batch citizen: progressive (SOF2) files through jsnoop_batch_add_jpeg, every scan of every image
    decoded together (one launch per dependency level).  DIBs equal the oracle's decode of the baseline form of the same
    coefficients (own encoder and libjpeg-turbo fixtures mixed in one batch), and a 64-image batch of the config-5 file is at
    least 8x faster per image than 64 single-file calls (12x measured since round 5 made the single call a quarter faster; 15x before)."""
    import json, os, time
    import jpegsnoop_amd as J
    here = os.path.dirname(os.path.abspath(__file__))
    cases = []                                                   # (progressive bytes, baseline bytes, W, H)
    for i, kw in enumerate([dict(width=160, height=96, hs=2, vs=1, restart_interval=10), dict(width=333, height=217, restart_interval=5),
                            dict(width=120, height=80, gray=1, restart_interval=3), dict(width=192, height=128, hs=1, vs=1, quality=100)]):
        cases.append((harness.synth_jpeg(seed=90 + i, progressive=1 + i % 2, **kw), harness.synth_jpeg(seed=90 + i, progressive=0, **kw), kw["width"], kw["height"]))
    for c in json.load(open(os.path.join(here, "golden", "pillow", "manifest.json")))["cases"]:
        rd = lambda kind: open(os.path.join(here, "golden", "pillow", f"{c['name']}_{kind}.jpg"), "rb").read()
        cases.append((rd("prog"), rd("base"), c["w"], c["h"]))
    b = J.JpegBatch()
    for prog, _, _, _ in cases:
        b.add_jpeg(prog)
    b.tile(2 * len(cases))
    b.upload(); b.decode(); b.sync()
    for i in range(2 * len(cases)):
        _, base, W, H = cases[i % len(cases)]
        harness.drive(oracle, base)
        a, g = oracle.dib(), b.dib(i)
        assert b.info(i)["path"] == 3 and b.info(i)["flags"] == 0
        assert a.shape == g.shape and np.array_equal(a[a.shape[0] - H:, :W], g[g.shape[0] - H:, :W]), f"image {i}: visible DIB differs"
    # a batch refuses to mix the two kinds
    with pytest.raises(RuntimeError):
        b.add_jpeg(cases[0][1])
    b.close()
    # throughput: 64 x config 5 (1920x1080 4:2:2, RSTn per MCU row, 10 scans)
    kw5 = dict(width=1920, height=1080, hs=2, vs=1, restart_interval=120, quality=85, seed=55)
    prog5, base5 = harness.synth_jpeg(progressive=2, **kw5), harness.synth_jpeg(progressive=0, **kw5)
    dec = J.CimgDecode()
    dec.DecodeProgressive(prog5)
    t0 = time.perf_counter()
    for _ in range(8):
        dec.DecodeProgressive(prog5)
    single_ms = (time.perf_counter() - t0) / 8 * 1e3
    harness.drive(oracle, base5)
    assert np.array_equal(dec.GetBitmapPtr(), oracle.dib())
    dec.close()
    bb = J.JpegBatch(); bb.add_jpeg(prog5); bb.tile(64); bb.upload(); bb.decode(); bb.sync()
    want = J.dib_checksum_numpy(oracle.dib())
    assert all(int(s) == want for s in bb.dib_checksums())
    ms, _ = bb.decode_timed(3)
    bb.close()
    print(f"config 5: single call {single_ms:.2f} ms per image, 64-image batch {ms / 64:.3f} ms per image")
    assert ms / 64 * 8 <= single_ms, (ms / 64, single_ms)
