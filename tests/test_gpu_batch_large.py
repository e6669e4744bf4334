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


def test_batch_above_the_long_subsequence_threshold(harness, oracle, monkeypatch):
    import jpegsnoop_amd as J
    monkeypatch.delenv("JSNOOP_SUB_WL", raising=False)
    files = [harness.synth_jpeg(width=1920, height=1080, hs=2, vs=2, quality=85, seed=500 + i) for i in range(8)]
    n = 192
    assert n * min(len(f) for f in files) >= 96 << 20, "batch too small to cross the threshold"
    b = J.JpegBatch()
    for f in files:
        b.add_jpeg(f)
    b.tile(n)
    b.upload(); b.decode(); b.sync()
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
    r3 = pipe.run(3, d2h=True)
    for b in pipe.slots:
        sums = b.dib_checksums()
        assert all(int(sums[i]) == want[i % 4] for i in range(128))
        assert all(b.info(i)["flags"] == 0 for i in range(128))
    assert r2["h2d_ms"] > 0 and r2["decode_ms"] > 0 and r3["d2h_ms"] > 0
    assert r2["ms_per_batch"] < 0.95 * (r2["h2d_ms"] + r2["decode_ms"]), r2          # the transfer hides behind the decode (or the other way round)
    assert r3["ms_per_batch"] < 0.95 * (r3["h2d_ms"] + r3["decode_ms"] + r3["d2h_ms"]), r3
    pipe.close()
