"""GPU: the two forms of the sub-sequence synchronisation stage agree with the oracle and with each other.

Small jobs synchronise by candidates (k_cand_spec / _walk / _chain / _apply: one speculative walk per block-in-MCU index, memo walks, a chain
of slot maps), larger ones by k_sync's rounds; both must leave the chain at the fixed point that IS the sequential decode of
CimgDecode::DecodeScanImg (source/ImgDecode.cpp:3021-3645: ReadScanVal / DecodeScanComp in scan order).  The library reads its switches at
upload: JSNOOP_CAND=0 (rounds only), JSNOOP_CAND=1 (one walk round of the chain: what stays open trips the write pass's verification and goes through k_sync's verification mode),
JSNOOP_CAND_LANES=1 (job "too large": rounds), default (up to sixteen walk rounds).  The smallest jobs also run the write pass with two lanes per
sub-sequence, the second one entering at the middle state the selected memo walk reported (JSNOOP_NO_HALF=1: one lane); with one walk round
only, the middle states of what k_sync repairs afterwards are stale and the write pass must notice (its verification fails, the decode resumes
with one lane per sub-sequence) -- the result is the oracle's in every case."""
import os

import pytest

pytestmark = pytest.mark.gpu

MODES = [("default", {}), ("rounds_only", {"JSNOOP_CAND": "0"}), ("one_walk_round", {"JSNOOP_CAND": "1"}), ("too_large", {"JSNOOP_CAND_LANES": "1"}),
         ("128_byte_pieces", {"JSNOOP_SUB_WL": "5"}), ("one_write_lane_per_piece", {"JSNOOP_NO_HALF": "1"})]


def _job(harness):
    kinds = [dict(width=1920, height=1080, hs=2, vs=2, quality=85), dict(width=1280, height=720, hs=2, vs=1, quality=75), dict(width=640, height=480, hs=1, vs=1, quality=92),
             dict(width=801, height=601, hs=2, vs=2, quality=50, restart_interval=7), dict(width=1024, height=768, gray=1, quality=80), dict(width=1920, height=1080, hs=2, vs=2, quality=85, noise_sigma=30),
             dict(width=2048, height=1536, hs=2, vs=2, quality=90, noise_sigma=2), dict(width=333, height=77, hs=2, vs=2, quality=85, optimize_huffman=1)]
    # seeds chosen so that several scans end on a byte boundary or close to the end of a 64-byte piece (the sub-sequences behind the data)
    return [harness.synth_jpeg(seed=500 + 7 * i + s, **k) for i, k in enumerate(kinds) for s in range(2)]


@pytest.fixture(scope="module")
def job(harness, oracle):
    import jpegsnoop_amd as J
    files = _job(harness)
    want = []
    for f in files:
        harness.drive(oracle, f)
        want.append(J.dib_checksum_numpy(oracle.dib()))
    return files, want


@pytest.mark.parametrize("mode,env", MODES, ids=[m for m, _ in MODES])
def test_synchronisation_forms_agree_with_the_oracle(job, mode, env):
    import jpegsnoop_amd as J
    files, want = job
    saved = {k: os.environ.get(k) for k in ("JSNOOP_CAND", "JSNOOP_CAND_LANES", "JSNOOP_SUB_WL", "JSNOOP_NO_HALF")}
    try:
        for k in saved:
            os.environ.pop(k, None)
        os.environ.update(env)
        for group in (files, files[:1], files[5:6], files[2:9]):      # the whole job, single images (one of them noisy), a mixed handful
            b = J.JpegBatch()
            for f in group:
                b.add_jpeg(f)
            b.upload(); b.decode(); b.sync()
            sums = b.dib_checksums()
            base = files.index(group[0])
            for i in range(len(group)):
                inf = b.info(i)
                assert inf["path"] == 1 and inf["flags"] == 0, (mode, base + i, inf)
                assert int(sums[i]) == want[base + i], (mode, base + i)
            b.decode(); b.sync()                                       # a second decode of the resident batch: same arenas, same answer
            assert [int(s) for s in b.dib_checksums()] == [int(s) for s in sums], mode
            b.close()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
