"""GPU: damaged files at device speed.  What the parallel path can vouch for stays (every block before the first anomaly it saw); the
exact-mirror reader takes over at the top of the MCU that holds the anomaly (k_entropy_exact in tail mode), and once it has run out of
file bytes ONE decoded MCU is replicated over the rest of the image (a truncated file decodes zeros: CwindowBuf::Buf returns 0 past the
end, source/WindowBuf.cpp:639).  Everything is compared with the oracle: DIB, planes, and -- through the side-only pass of the whole
mirror -- MCU file map, block-DC maps, code-length histogram, status words.  Reference behaviour mirrored: coefficient-index overflow
source/ImgDecode.cpp:1723-1735, bad codes :1178-1186, markers inside the scan :1486-1561, restart handling :1644-1680, scan stop :3623-3625."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _damage(harness, base, kind, frac):
    p = harness.parse_jpeg(base)
    d = bytearray(base)
    i = p.scan_start + int((p.scan_end - p.scan_start) * frac)
    if kind == "cut":
        d = d[:i]
    elif kind == "marker":
        d[i:i + 2] = b"\xff\xe3"
    elif kind == "rst":
        d[i:i] = b"\xff\xd5"
    elif kind == "rst2":                                              # two restart markers back to back
        j = next(k for k in range(i, p.scan_end - 1) if d[k] == 0xFF and 0xD0 <= d[k + 1] <= 0xD7)
        d[j:j] = b"\xff\xd2"
    elif kind == "delete":
        del d[i:i + 3]
    elif kind == "zeros":
        d[i:i + 40] = bytes(40)
    elif kind == "ones":
        d[i:i + 6] = b"\xff\x00" * 3
    return bytes(d)


CASES = [(dict(width=1920, height=1080), "cut", 0.5), (dict(width=1920, height=1080, restart_interval=120), "cut", 0.93),
         (dict(width=640, height=480), "marker", 0.8), (dict(width=640, height=480, restart_interval=40), "rst", 0.6),
         (dict(width=640, height=480, hs=1, vs=1, restart_interval=1), "rst2", 0.5), (dict(width=800, height=600, hs=2, vs=1), "delete", 0.9),
         (dict(width=640, height=480), "zeros", 0.7), (dict(width=640, height=480, gray=1), "ones", 0.85),
         (dict(width=333, height=217, restart_interval=5), "cut", 0.4)]


@pytest.mark.parametrize("kw,kind,frac", CASES)
def test_damaged_file_is_exact_and_stays_on_the_device_fast_path(harness, oracle, gpu, kw, kind, frac):
    from fuzz_util import differs
    base = harness.synth_jpeg(seed=17, **kw)
    data = _damage(harness, base, kind, frac)
    harness.drive(oracle, data)
    harness.drive(gpu, data)
    if kind not in ("delete", "ones", "zeros"):                       # (lost or overwritten bytes may leave a stream that still parses: nothing to flag)
        assert gpu.lib.jsnoop_last_flags(gpu.h) != 0, "the damage must leave a trace"
    if kind != "rst2":                                                # (two markers back to back: the one restart anomaly the walks leave to the whole mirror)
        assert gpu.lib.jsnoop_last_path(gpu.h) == 1, "the parallel path's blocks must have been kept"
    assert differs(oracle, gpu) is None


def test_truncated_1080p_decodes_in_milliseconds(harness, oracle):
    import jpegsnoop_amd as J
    base = harness.synth_jpeg(width=1920, height=1080, seed=23)
    data = _damage(harness, base, "cut", 0.5)
    b = J.JpegBatch(); b.add_jpeg(data); b.upload(); b.decode(); b.sync()
    t = time.perf_counter(); b.decode(); b.sync(); ms = (time.perf_counter() - t) * 1e3
    harness.drive(oracle, data)
    assert b.info(0)["path"] == 1 and b.info(0)["flags"] != 0
    assert np.array_equal(b.dib(0), oracle.dib())
    assert ms < 100.0, f"{ms:.1f} ms: the sequential mirror over half a 1080p picture of zero bits takes seconds"
    b.close()


@pytest.mark.parametrize("kind", ["ones16", "garbage"])
def test_codes_that_match_nothing_are_decoded_by_the_parallel_path(harness, oracle, gpu, kind):
    """A code that matches nothing far from any marker (ReadScanVal: one bit consumed, RSV_UNDERFLOW, :1178-1186 / :1270-1282; DecodeScanComp
    gives the block up without an IDCT, :1737-1757): the walks of the parallel path do the same -- the block keeps its DC difference, its AC
    part is emptied, the next block starts one bit on -- so a 1080p file with sixteen one-bits, or 256 random bytes (FF among them: the scan
    is then decoded a second time through its stray markers), at 30 % of its scan decodes in well under 50 ms instead of the seconds the
    sequential mirror takes for the remaining 70 %; DIB and planes are the oracle's, and so are the side outputs (the mirror's side-only
    pass, on request)."""
    import jpegsnoop_amd as J
    from fuzz_util import differs
    base = harness.synth_jpeg(width=1920, height=1080, seed=31)
    p = harness.parse_jpeg(base)
    d = bytearray(base); i = p.scan_start + int((p.scan_end - p.scan_start) * 0.3)
    if kind == "ones16":
        d[i:i + 4] = b"\xff\x00\xff\x00"
    else:
        d[i:i + 256] = np.random.RandomState(5).randint(0, 256, 256).astype(np.uint8).tobytes()
    data = bytes(d)
    b = J.JpegBatch(want_planes=True); b.add_jpeg(data); b.upload(); b.decode(); b.sync()
    t = time.perf_counter(); b.decode(); b.sync(); ms = (time.perf_counter() - t) * 1e3
    harness.drive(oracle, data)
    inf = b.info(0)
    assert inf["path"] == 1 and (inf["flags"] & 0x0001) and not (inf["flags"] & 0x0100), inf      # BAD_CODE, not BAD_EDGE
    assert np.array_equal(b.dib(0), oracle.dib())
    for pa, pb in zip(oracle.planes(), b.planes(0)):
        assert np.array_equal(pa, pb)
    assert ms < 50.0, f"{ms:.1f} ms"
    b.close()
    harness.drive(gpu, data)                                          # the single-image API: side outputs and status words too
    assert differs(oracle, gpu) is None


def test_restart_marker_inside_a_block_is_followed_by_the_parallel_path(harness, oracle, gpu):
    """Two bytes lost in front of the second RSTn of a 1080p file: the reference meets the marker INSIDE a block, clears its DC predictors and
    goes on with the block in the new interval (DecodeScanComp :1644-1680) -- every later marker then falls off an MCU boundary too.  The walks
    keep coefficient index and block position across the marker and the DC scan clears its sums in front of the block in progress: the file
    decodes in well under 50 ms (round 3: the sequential mirror, 2.3 s), bit for bit the oracle's DIB, planes and side outputs."""
    import jpegsnoop_amd as J
    from fuzz_util import differs
    base = harness.synth_jpeg(width=1920, height=1080, restart_interval=120, seed=33)
    p = harness.parse_jpeg(base)
    d = bytearray(base); j = bytes(d).index(b"\xff\xd1", p.scan_start); del d[j - 2:j]
    data = bytes(d)
    b = J.JpegBatch(want_planes=True); b.add_jpeg(data); b.upload(); b.decode(); b.sync()
    t = time.perf_counter(); b.decode(); b.sync(); ms = (time.perf_counter() - t) * 1e3
    harness.drive(oracle, data)
    inf = b.info(0)
    assert inf["path"] == 1 and (inf["flags"] & 0x0008) and not (inf["flags"] & 0x0100), inf
    assert np.array_equal(b.dib(0), oracle.dib())
    for pa, pb in zip(oracle.planes(), b.planes(0)):
        assert np.array_equal(pa, pb)
    assert ms < 50.0, f"{ms:.1f} ms"
    b.close()
    harness.drive(gpu, data)
    assert differs(oracle, gpu) is None


def test_damaged_images_inside_a_batch(harness, oracle):
    import jpegsnoop_amd as J
    base = harness.synth_jpeg(width=640, height=480, seed=29)
    files = [base, _damage(harness, base, "cut", 0.6), _damage(harness, base, "marker", 0.3), base, _damage(harness, base, "delete", 0.5),
             _damage(harness, harness.synth_jpeg(width=640, height=480, seed=30, restart_interval=40), "rst", 0.45)]
    b = J.JpegBatch(want_planes=True)
    for f in files:
        b.add_jpeg(f)
    b.upload(); b.decode(); b.sync()
    for i, f in enumerate(files):
        harness.drive(oracle, f)
        assert np.array_equal(b.dib(i), oracle.dib()), i
        for pa, pb in zip(oracle.planes(), b.planes(i)):
            if pa is not None:
                assert np.array_equal(pa, pb), i
        so = b.side_outputs(i)
        assert np.array_equal(so["mcu_map"], oracle.mcu_map()), i
        assert {k: int(v) for k, v in so["status"].items()} == {("rst_count" if k == "restart_read" else k): int(v) for k, v in oracle.status().items()}, i
    assert b.info(0)["flags"] == 0 and all(b.info(i)["path"] == 1 for i in range(len(files)))
    b.close()


def test_overflow_only_files_get_their_bookkeeping_from_the_parallel_side_pass(harness, oracle, gpu):
    """85 % of the damaged files that leave a trace leave only coefficient-index overflows (tools/damage_survey.py).  Their pixels are the
    parallel path's; their status words, maps, histogram and the log text (two messages per overflow, sharing the reference's warning
    budget with the end-of-scan markers) come from the parallel side pass + the overflow records -- compared here with the compiled
    reference itself (its log) where its library travelled, with the oracle otherwise."""
    from fuzz_util import differs
    ref = harness.ref_backend() if harness.have_ref() else None
    rng = np.random.default_rng(77)
    bases = [harness.synth_jpeg(width=320, height=240, seed=41), harness.synth_jpeg(width=333, height=217, hs=2, vs=1, restart_interval=7, seed=42),
             harness.synth_jpeg(width=256, height=192, gray=1, seed=43), harness.synth_jpeg(width=200, height=152, hs=1, vs=1, quality=95, seed=44)]
    found = 0
    try:
        for trial in range(400):
            base = bases[trial % len(bases)]
            p = harness.parse_jpeg(base)
            d = bytearray(base)
            for _ in range(int(rng.integers(1, 4))):
                d[int(rng.integers(p.scan_start, p.scan_end - 2))] ^= 1 << int(rng.integers(8))
            d = bytes(d)
            em = int(rng.choice([20, 20, 3, 2, 1]))
            gpu.set_options(decode_ac=1, err_max=em)
            try:
                q = harness.parse_jpeg(d)
            except Exception:
                continue
            harness.drive(gpu, d, q, quiet=0)
            if gpu.lib.jsnoop_last_flags(gpu.h) != 0x0004 or gpu.lib.jsnoop_last_path(gpu.h) != 1:
                continue
            found += 1
            got_log = gpu.log_lines()
            oracle.set_options(decode_ac=1, err_max=em)
            harness.drive(oracle, d, q)
            assert differs(oracle, gpu) is None, (trial, em)
            if ref is not None:
                ref.set_options(decode_ac=1, err_max=em)
                harness.drive(ref, d, q, quiet=0)
                assert got_log == ref.log_lines(), (trial, em, next((i, a, b) for i, (a, b) in enumerate(zip(got_log + [None], ref.log_lines() + [None])) if a != b))
            if found >= 25:
                break
        assert found >= 10, found
    finally:
        gpu.set_options(); oracle.set_options()
        if ref is not None:
            ref.set_options(); ref.close()


@pytest.mark.parametrize("tuning", [dict(split=2), dict(split=2, sub_wl=7, cand_rounds=-1), dict(split=1, sub_wl=7, cand_rounds=-1),
                                    dict(split=2, sub_wl=7, cand_rounds=-1, sync_launches=2), dict(split=1, sub_wl=5, cand_rounds=-1)])   # (cand_rounds = -1: rounds -- list rounds unless sync_launches asks for plain k_sync launches)
def test_damaged_files_inside_a_batch_on_two_streams(harness, oracle, tuning):
    """Damaged and healthy files mixed in one batch, decoded as two halves on two streams (the default form of a large batch) and with the 512-byte
    sub-sequences of large batches: flags are per image, the repair passes (tail take-over, second attempt, resumed chain) run behind the join of
    the streams -- every DIB is the oracle's, the healthy neighbours of a damaged file stay on the parallel path with no flag."""
    import jpegsnoop_amd as J
    base = [harness.synth_jpeg(width=640, height=480, seed=31), harness.synth_jpeg(width=800, height=600, hs=2, vs=1, restart_interval=25, seed=32),
            harness.synth_jpeg(width=512, height=384, hs=1, vs=1, seed=33)]
    files = [base[0], _damage(harness, base[0], "cut", 0.6), base[1], _damage(harness, base[1], "rst", 0.5), _damage(harness, base[2], "marker", 0.7),
             base[2], _damage(harness, base[0], "zeros", 0.3), _damage(harness, base[1], "delete", 0.8), base[1]]
    healthy = {0, 2, 5, 8}
    b = J.JpegBatch()
    b.set_tuning(**tuning)
    for f in files:
        b.add_jpeg(f)
    b.upload()
    for _ in range(2):
        b.decode(); b.sync()
        assert b.split_parts() == tuning["split"]
        for i, f in enumerate(files):
            harness.drive(oracle, f)
            assert np.array_equal(b.dib(i), oracle.dib()), (tuning, i)
            if i in healthy:
                assert b.info(i)["path"] == 1 and b.info(i)["flags"] == 0, (tuning, i, b.info(i))
    b.close()


def _overrun_file(harness, J, seed=35):
    """A 1080p file with restart markers in which, behind a marker met INSIDE a block (two bytes lost in front of the second RSTn: every later
    marker then falls off an MCU boundary), the value bits of some symbol run past the end of a restart interval -- the reference's register
    over-reads there (ReadScanVal :1229-1282), scan_end / scan_bad stay set, and its decode of the image is over (:3623-3625).  Found by cutting
    the tail off the interval in front of a later marker until the parallel path reports the overrun."""
    base = harness.synth_jpeg(width=1920, height=1080, restart_interval=120, seed=seed)
    p = harness.parse_jpeg(base)
    d = bytearray(base); j = bytes(d).index(b"\xff\xd1", p.scan_start); del d[j - 2:j]
    marks = [k for k in range(p.scan_start, len(d) - 1) if d[k] == 0xFF and 0xD0 <= d[k + 1] <= 0xD7]
    b = J.JpegBatch()
    try:
        for which in (len(marks) // 3, len(marks) // 2, 2 * len(marks) // 3):
            for cut in range(1, 7):
                e = bytearray(d); k = marks[which]
                if 0xFF in e[k - cut - 1:k]:
                    continue
                del e[k - cut:k]
                b.clear(); b.add_jpeg(bytes(e)); b.upload(); b.decode(); b.sync()
                if b.info(0)["flags"] & 0x0002:
                    return bytes(e)
    finally:
        b.close()
    pytest.skip("no cut produced a value-bit overrun on this picture")


def test_value_bits_past_an_interval_end_end_the_decode_on_the_device(harness, oracle, gpu):
    """The slow class round 4 left (every one of the hostile 1080p files above 50 ms carried JSNOOP_FLAG_OVERRUN beside a misplaced marker: whole
    mirror, 0.7-2.4 s): the decode of such a file ENDS at the overrun -- the block in progress keeps its DC difference, every later block fails
    at its first read, the MCU row stops and of each later row only the first MCU is reached.  The parallel path keeps everything up to that
    block and fills the rest in on the device (k_dead_fill, DC scan, k_dead_rows)."""
    import jpegsnoop_amd as J
    from fuzz_util import differs
    data = _overrun_file(harness, J)
    b = J.JpegBatch(want_planes=True); b.add_jpeg(data); b.upload(); b.decode(); b.sync()
    t = time.perf_counter(); b.decode(); b.sync(); ms = (time.perf_counter() - t) * 1e3
    harness.drive(oracle, data)
    inf = b.info(0)
    assert inf["path"] == 1 and (inf["flags"] & 0x0002) and (inf["flags"] & 0x0008), inf                 # OVERRUN beside RST_MISALIGN: round 4's whole-mirror class
    assert np.array_equal(b.dib(0), oracle.dib())
    for pa, pb in zip(oracle.planes(), b.planes(0)):
        assert np.array_equal(pa, pb)
    assert ms < 50.0, f"{ms:.1f} ms"
    b.close()
    harness.drive(gpu, data)                                          # single-image API: side outputs and status words (the mirror's side-only pass) too
    assert differs(oracle, gpu) is None


def test_one_hostile_file_does_not_cost_the_batch_its_time(harness, oracle):
    """A batch of 192 x 1080p (the production form: 512-byte pieces, two streams) with ONE hostile file in it -- the overrun file above, the worst class of
    tools/fuzz_1080p_timing.py -- takes at most twice the time of the clean batch: the repair touches that image alone (its fill-in, its DC scan, its own
    workgroups of the back end), not the batch's back end."""
    import jpegsnoop_amd as J
    hostile = _overrun_file(harness, J)
    files = [harness.synth_jpeg(width=1920, height=1080, seed=700 + i) for i in range(8)]
    def run(extra):
        b = J.JpegBatch()
        for f in files:
            b.add_jpeg(f)
        b.tile(191)
        b.add_jpeg(extra)
        b.upload(); b.decode(); b.sync()
        best = 1e9
        for _ in range(3):
            t = time.perf_counter(); b.decode(); b.sync(); best = min(best, (time.perf_counter() - t) * 1e3)
        return b, best
    bc, clean = run(files[0])
    assert all(bc.info(i)["flags"] == 0 for i in range(192))
    bc.close()
    bh, dirty = run(hostile)
    harness.drive(oracle, hostile)
    assert bh.info(191)["path"] == 1 and bh.info(191)["flags"] & 0x0002
    assert np.array_equal(bh.dib(191), oracle.dib())
    sums = bh.dib_checksums()
    for j, f in enumerate(files):
        harness.drive(oracle, f)
        want = J.dib_checksum_numpy(oracle.dib())
        assert all(int(s) == want for s in sums[j:191:8]), j
    bh.close()
    assert dirty <= 2.0 * clean, f"clean {clean:.2f} ms, with one hostile file {dirty:.2f} ms"


def test_decode_end_in_the_first_block_behind_a_restart(harness, oracle, gpu):
    """tools/fuzz_gpu.py seed 4244 case 352: a 4:4:4 file with a restart marker behind every MCU and a stray RSTn in the middle of an MCU.  The value bits of a
    symbol of the MCU's FIRST block run into the stray marker -- the decode ends in a block whose restart mark (set in front of the block, possibly by the lane
    before its owner) is the reference's, while a mark the walk sets on the same block afterwards is not: the fill-in keeps the first kind (mark_reset's bit 7)."""
    from fuzz_util import differs
    base = harness.synth_jpeg(width=128, height=64, hs=1, vs=1, restart_interval=1, seed=3)
    for at in (844, 700, 1200, 2000):
        d = bytearray(base); d[at:at] = b"\xff\xd0"
        data = bytes(d)
        harness.drive(oracle, data)
        harness.drive(gpu, data)
        assert differs(oracle, gpu) is None, at


def test_stray_restart_marker_shortly_behind_a_regular_one(harness, oracle, gpu):
    """tools/fuzz_1080p_timing.py seed 23 case 974 (255 ms in the sequential mirror): an RSTn inserted 43 bytes behind a regular restart marker -- inside the
    FIRST MCU of the interval, so that MCU has two resets of the DC predictors: the regular one on its boundary and the stray one in front of (or inside) a later
    block.  The MCU's mark holds both (mark_reset's bit 6), the DC scan clears its sums twice, and the file stays on the device's fast path."""
    import re
    import jpegsnoop_amd as J
    from fuzz_util import differs
    base = harness.synth_jpeg(width=1920, height=1080, restart_interval=8, seed=63)
    p = harness.parse_jpeg(base)
    marks = [m.start() + p.scan_start for m in re.finditer(rb"\xff[\xd0-\xd7]", base[p.scan_start:p.scan_end])]
    fast = 0
    for which, off in ((len(marks) // 2, 43), (len(marks) // 3, 9), (5, 20), (len(marks) - 3, 31), (17, 64)):
        at = marks[which] + 2 + off
        d = bytearray(base); d[at:at] = b"\xff\xd0"
        data = bytes(d)
        b = J.JpegBatch(); b.add_jpeg(data); b.upload(); b.decode(); b.sync()
        t = time.perf_counter(); b.decode(); b.sync(); ms = (time.perf_counter() - t) * 1e3
        harness.drive(oracle, data)
        inf = b.info(0)
        assert np.array_equal(b.dib(0), oracle.dib()), (which, off, inf)
        if not (inf["flags"] & 0x0100):
            fast += 1
            assert inf["path"] == 1 and ms < 20.0, (which, off, inf, ms)
        b.close()
        harness.drive(gpu, data)
        assert differs(oracle, gpu) is None, (which, off)
    assert fast >= 3, fast


def test_filler_bytes_between_two_restart_markers(harness, oracle, gpu):
    """tools/fuzz_gpu.py seed 11 case 1038 (round 6; wrong pixels in rounds 1-5): RSTn, FF FF, RSTm -- an interval of two filler bytes kept as data
    (BuffAddByte :1486-1561) that holds no code.  The reference enters it inside the retry of DecodeScanComp's restart handling (:1644-1680), the retry
    comes back with RSV_RST_TERM as well, which is NOT handled there (the ASSERT of :1676 is all there is): the read counts as a coefficient and the
    second restart happens one read later.  An interval left before a bit of it was consumed is the mirror's (BAD_EDGE), like two markers back to back."""
    import re
    from fuzz_util import differs
    base = harness.synth_jpeg(width=144, height=96, hs=2, vs=1, restart_interval=3, seed=71)
    p = harness.parse_jpeg(base)
    marks = [m.start() + p.scan_start for m in re.finditer(rb"\xff[\xd0-\xd7]", base[p.scan_start:p.scan_end])]
    for which in (len(marks) - 1, len(marks) // 2, 3):
        at = marks[which] + 2
        d = bytearray(base); d[at:at] = bytes([0xFF, 0xFF, 0xFF, 0xD0 + ((base[marks[which] + 1] - 0xD0 + 1) & 7)])
        data = bytes(d)
        harness.drive(oracle, data)
        harness.drive(gpu, data)
        assert gpu.lib.jsnoop_last_flags(gpu.h) & 0x0100, "an interval without a code must be left to the mirror"
        assert differs(oracle, gpu) is None, which


def test_report_of_a_damaged_file_comes_from_chunked_exact_readers(harness, oracle, gpu):
    """Round 6: log text and side outputs of a flagged file -- what a CjfifDecode asks DecodeScanImg for (source/JfifDecode.cpp:5299) -- no longer wait for the
    sequential mirror over the whole image (1.2 s per 1080p file, against ~140 ms of the CPU reference).  The walks vouch for the bit position of every MCU top
    of a stream they followed the reference's way (bad codes :1178-1186, restarts inside a block :1644-1680, coefficient overflows :1723-1735, the end of
    the reference's own decode :3623-3625); k_side_chunks puts one exact reader on every chunk of eight MCUs and the host merges what they logged.  Compared
    with the compiled reference's log line for line where its library travelled, with the oracle's side outputs and status words always."""
    from fuzz_util import differs
    ref = harness.ref_backend() if harness.have_ref() else None
    base = harness.synth_jpeg(width=1920, height=1080, seed=31)
    base_rst = harness.synth_jpeg(width=1920, height=1080, restart_interval=120, seed=82)
    p = harness.parse_jpeg(base)
    def at(frac, b=base):
        q = harness.parse_jpeg(b); return q.scan_start + int((q.scan_end - q.scan_start) * frac)
    cases = []
    d = bytearray(base); d[at(0.3):at(0.3) + 4] = b"\xff\x00\xff\x00"; cases.append(("ones16", bytes(d)))
    d = bytearray(base); i = at(0.3); d[i:i + 256] = np.random.RandomState(5).randint(0, 255, 256).astype(np.uint8).tobytes(); cases.append(("garbage", bytes(d)))
    d = bytearray(base); d[at(0.6)] ^= 0x10; cases.append(("bitflip", bytes(d)))
    d = bytearray(base_rst); j = bytes(d).index(b"\xff\xd1", harness.parse_jpeg(base_rst).scan_start); del d[j - 2:j]; cases.append(("rst_in_block", bytes(d)))
    d = bytearray(base_rst); i = at(0.45, base_rst); del d[i:i + 3]; cases.append(("deleted_bytes_rst", bytes(d)))
    modes = {}
    try:
        for name, data in cases:
            for em in (20, 2):
                for b in (oracle, gpu) + ((ref,) if ref else ()):
                    b.set_options(decode_ac=1, err_max=em)
                q = harness.parse_jpeg(data)                       # (the header walk is the caller's -- Python here -- and not part of the call)
                harness.drive(gpu, data, q, quiet=0)               # (first call: allocations)
                t = time.perf_counter(); harness.drive(gpu, data, q, quiet=0); ms = (time.perf_counter() - t) * 1e3
                got = gpu.log_lines()
                fl, sm = gpu.lib.jsnoop_last_flags(gpu.h), gpu.lib.jsnoop_last_side_mode(gpu.h)
                if fl == 0:                                        # (lost bytes may leave a stream that still parses: nothing to report)
                    continue
                modes[name] = sm
                harness.drive(oracle, data)
                assert differs(oracle, gpu) is None, (name, em)
                if ref is not None:
                    harness.drive(ref, data, quiet=0)
                    want = ref.log_lines()
                    assert got == want, (name, em, next((i, a, b) for i, (a, b) in enumerate(zip(got + [None], want + [None])) if a != b))
                if sm == 3:
                    assert ms < 15.0, (name, em, ms)               # upload + decode + side pass + chunked readers + report (typically 3-4 ms; the mirror: 600-1200)
        assert sum(1 for v in modes.values() if v == 3) >= 2, modes
    finally:
        for b in (oracle, gpu) + ((ref,) if ref else ()):
            b.set_options()
        if ref is not None:
            ref.close()


def test_scan_that_begins_with_a_restart_marker(harness, oracle, gpu):
    """tools/fuzz_gpu.py seed 202 case 4625 (round 6; wrong MCU file map entry since round 1): an RSTn as the scan's first two bytes.  The reference's very first
    refill meets the marker and loads nothing (:3007-3019): at the top of MCU 0 its register is empty, the position array holds DecodeRestartScanBuf's zeros
    (:4038-4075) -- LookupFilePosMcu(0, 0) is 0 --, the marker's index is reported ABOVE the "*** Decoding SCAN Data ***" heading, and the restart is handled
    inside MCU 0.  No flag is raised (nothing is wrong with the stream the walks see): the parallel side pass has to know."""
    from fuzz_util import differs
    ref = harness.ref_backend() if harness.have_ref() else None
    try:
        for kw, n in ((dict(width=104, height=64, hs=1, vs=1), 3), (dict(width=160, height=96, hs=2, vs=2), 0), (dict(width=128, height=64, hs=2, vs=1, restart_interval=4), 5)):
            base = harness.synth_jpeg(seed=91, **kw)
            p = harness.parse_jpeg(base)
            d = bytearray(base); d[p.scan_start:p.scan_start] = bytes([0xFF, 0xD0 + n])
            data = bytes(d)
            harness.drive(oracle, data)
            harness.drive(gpu, data, quiet=0)
            got = gpu.log_lines()
            assert differs(oracle, gpu) is None, (kw, n)
            assert int(np.asarray(gpu.mcu_map()).ravel()[0]) == 0
            if ref is not None:
                harness.drive(ref, data, quiet=0)
                want = ref.log_lines()
                assert got == want, (kw, n, next((i, a, b) for i, (a, b) in enumerate(zip(got + [None], want + [None])) if a != b))
    finally:
        if ref is not None:
            ref.close()


def test_restart_marker_behind_the_last_mcu(harness, oracle, gpu):
    """tools/fuzz_damaged_log.py seed 701 case 2164 (round 6; older than the round): garbage over an RST7 -- the decode runs through where the marker was, the image's
    MCUs are used up while the file still holds two restart intervals, and the look-ahead behind the LAST MCU meets RST0 where RST7 is expected: the reference
    reports that like any other wrong index (:1416-1423).  The end-of-scan reader of the parallel side pass kept only the "Scan Data encountered marker" messages;
    now it records the marker with the expectation left open and the host, which has followed the markers up to there, fills it in (or drops the record).
    The file is the fuzz case itself (tests/golden/fuzz/)."""
    import os
    from fuzz_util import differs
    data = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz", "rst_behind_last_mcu_s701_c2164.jpg"), "rb").read()
    ref = harness.ref_backend() if harness.have_ref() else None
    try:
        for b in (oracle, gpu) + ((ref,) if ref else ()): b.set_options(decode_ac=1, err_max=20)
        harness.drive(oracle, data)
        harness.drive(gpu, data, quiet=0)
        got = gpu.log_lines()
        assert differs(oracle, gpu) is None
        assert gpu.lib.jsnoop_last_path(gpu.h) == 1 and gpu.lib.jsnoop_last_side_mode(gpu.h) == 1      # (the parallel side pass with its overflow records)
        assert any("Expected RST marker index RST7 got RST0 @ 0x000018C1.0" in l for l in got)
        if ref is not None:
            harness.drive(ref, data, quiet=0)
            want = ref.log_lines()
            assert got == want, next((i, a, b) for i, (a, b) in enumerate(zip(got + [None], want + [None])) if a != b)
    finally:
        for b in (oracle, gpu): b.set_options()
        if ref is not None:
            ref.set_options(); ref.close()
