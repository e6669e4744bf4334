"""GPU: the tunable forms of the parallel entropy path give the answer of the default one.

* every sub-sequence length the kernels are instantiated for (JsnoopTuning.sub_wl = 4 ... 8: 64 B ... 1 KiB per lane; the library
  picks 4, 5 or 7 by batch size on its own) -- same DIBs, same side outputs, parallel path taken;
* the first form of the write pass (k_write<., false>, cross_checks = JSNOOP_XC_WRITE_V1) against the
  second (k_write2, the one the main path launches): the checksums of a mixed batch, RSTn streams and 4:4:4 / 4:2:2 / 4:2:0 / gray
  layouts included, must be identical and equal to the oracle's.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KWS = [dict(width=640, height=480, hs=1, vs=1), dict(width=1280, height=720, hs=2, vs=2), dict(width=800, height=600, hs=2, vs=1, restart_interval=25),
       dict(width=512, height=512, gray=1), dict(width=1000, height=700, hs=2, vs=2, restart_interval=1), dict(width=333, height=217, hs=1, vs=2, quality=97),
       dict(width=1920, height=1080, hs=2, vs=2, quality=30)]


def _files(harness):
    return [harness.synth_jpeg(seed=900 + i, **kw) for i, kw in enumerate(KWS)]


@pytest.mark.parametrize("wl", [4, 5, 6, 7, 8])
def test_every_subsequence_length(harness, oracle, wl):
    import jpegsnoop_amd as J
    files = _files(harness)
    b = J.JpegBatch()
    b.set_tuning(sub_wl=wl)
    for f in files:
        b.add_jpeg(f)
    b.tile(3 * len(files))
    b.upload(); b.decode(); b.sync()
    for i in range(3 * len(files)):
        harness.drive(oracle, files[i % len(files)])
        assert b.info(i)["path"] == 1 and b.info(i)["flags"] == 0, (wl, i)
        assert np.array_equal(b.dib(i), oracle.dib()), (wl, i, KWS[i % len(files)])
    b.close()


def _decode_all(J, harness, **tuning):
    b = J.JpegBatch()
    b.set_tuning(**tuning)
    for f in _files(harness):
        b.add_jpeg(f)
    n = 4 * len(KWS)
    b.tile(n)
    b.upload(); b.decode(); b.sync()
    out = {"sums": [int(x) for x in b.dib_checksums()], "paths": [b.info(i)["path"] for i in range(n)], "flags": [b.info(i)["flags"] for i in range(n)]}
    b.close()
    return out


def test_write_pass_first_and_second_form_agree(harness, oracle):
    import jpegsnoop_amd as J
    from jpegsnoop_amd import capi
    v2 = _decode_all(J, harness)
    v1 = _decode_all(J, harness, cross_checks=capi.XC_WRITE_V1)
    assert v1["sums"] == v2["sums"]
    assert set(v1["paths"]) == {1} and set(v2["paths"]) == {1} and not any(v1["flags"]) and not any(v2["flags"])
    for j, f in enumerate(_files(harness)):
        harness.drive(oracle, f)
        want = J.dib_checksum_numpy(oracle.dib())
        assert all(s == want for s in v2["sums"][j::len(KWS)]), KWS[j]


def test_two_stream_split_gives_the_same_batch(harness, oracle):
    """jsnoop_batch_set_split(2): the two halves of a batch on two streams, same arenas -- every DIB as with one stream, flags clean,
    a stream with restart markers and a grayscale image among them; then back to one stream.  (0 = automatic: one stream for a batch this small.)"""
    import jpegsnoop_amd as J
    files = _files(harness)
    n = 5 * len(files) + 3                                          # an odd count: halves of different size
    b = J.JpegBatch()
    for f in files:
        b.add_jpeg(f)
    b.tile(n)
    b.set_split(1)
    b.upload(); b.decode(); b.sync()
    one = b.dib_checksums().copy()
    assert b.split_parts() == 1
    b.set_split(0); assert b.split_parts() == 1                    # automatic: below 8 MB of scan data
    b.set_split(2); assert b.split_parts() == 2
    for _ in range(3):
        b.decode()
    b.sync()
    assert np.array_equal(b.dib_checksums(), one)
    assert all(b.info(i)["path"] == 1 and b.info(i)["flags"] == 0 for i in range(n))
    ms, stages = b.decode_timed(2)                                  # the timed form of a split decode: mean of the halves' stage times
    assert ms > 0 and all(v >= 0 for v in stages.values())
    assert np.array_equal(b.dib_checksums(), one)
    for j, f in enumerate(files):
        harness.drive(oracle, f)
        assert int(one[j]) == J.dib_checksum_numpy(oracle.dib()), KWS[j]
    b.set_split(1); b.decode(); b.sync()
    assert np.array_equal(b.dib_checksums(), one)
    b.close()


def test_fused_unstuffing_agrees_with_the_three_pass_form(harness, oracle):
    """The decode un-stuffs in ONE pass over the file bytes (k_unstuff_write<true> for 64-byte sub-sequences, k_unstuff_fused for longer ones,
    which also writes the interleaved layout directly: a chunk's place in its image comes from a decoupled look-back over the chunks before
    it); the count / scan / write (/ transpose) form it replaced stays behind JSNOOP_XC_UNSTUFF_3PASS.  Same DIBs, and -- repeated decodes
    of the same resident batch -- the epoch tag of the scan state keeps one decode's words apart from the next one's."""
    import jpegsnoop_amd as J
    from jpegsnoop_amd import capi
    for wl in (0, 5, 6, 7, 8):
        fused = _decode_all(J, harness, sub_wl=wl)
        three = _decode_all(J, harness, sub_wl=wl, cross_checks=capi.XC_UNSTUFF_3PASS)
        assert fused["sums"] == three["sums"] and set(fused["paths"]) == {1} and not any(fused["flags"]) and not any(three["flags"]), wl
    files = _files(harness) + [harness.synth_jpeg(seed=77, width=3840, height=2160, hs=2, vs=2, restart_interval=7)]     # (a scan of ~600 chunks: look-back over several trips)
    for wl in (0, 7):
        b = J.JpegBatch()
        b.set_tuning(sub_wl=wl)
        for f in files:
            b.add_jpeg(f)
        b.upload()
        for _ in range(300):                                       # past the wrap of the 8-bit epoch
            b.decode()
        b.sync()
        for i, f in enumerate(files):
            harness.drive(oracle, f)
            assert b.info(i)["path"] == 1 and b.info(i)["flags"] == 0, (wl, i)
            assert int(b.dib_checksums()[i]) == J.dib_checksum_numpy(oracle.dib()), (wl, i)
        b.close()


def test_list_rounds_and_plain_launches_of_the_synchronisation_by_rounds(harness, oracle):
    """A job that synchronises by rounds runs ONE cut launch of k_sync (tail walks + one whole walk), per-image lists of the sub-sequences whose entry state is not
    their left neighbour's exit state, and list rounds over the whole job (js_launch_sync_rounds); JsnoopTuning.sync_launches > 0 asks for that many plain launches
    of k_sync instead (the form before round 6).  Same DIBs either way, for every sub-sequence length, on one stream and two, repeated decodes included."""
    import jpegsnoop_amd as J
    want = _decode_all(J, harness)
    for wl in (4, 5, 6, 7, 8):
        for split in (1, 2):
            got = _decode_all(J, harness, sub_wl=wl, cand_rounds=-1, split=split)
            assert got["sums"] == want["sums"] and set(got["paths"]) == {1} and not any(got["flags"]), (wl, split)
        plain = _decode_all(J, harness, sub_wl=wl, cand_rounds=-1, sync_launches=2)
        assert plain["sums"] == want["sums"] and set(plain["paths"]) == {1} and not any(plain["flags"]), wl
    files = _files(harness) + [harness.synth_jpeg(seed=78, width=3840, height=2160, hs=2, vs=2, restart_interval=7)]
    b = J.JpegBatch()
    b.set_tuning(sub_wl=6, cand_rounds=-1)
    for f in files:
        b.add_jpeg(f)
    b.upload()
    for _ in range(5):
        b.decode()
    b.sync()
    for i, f in enumerate(files):
        harness.drive(oracle, f)
        assert b.info(i)["path"] == 1 and b.info(i)["flags"] == 0, i
        assert int(b.dib_checksums()[i]) == J.dib_checksum_numpy(oracle.dib()), i
    b.close()
