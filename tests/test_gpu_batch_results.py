"""GPU: everything DecodeScanImg leaves behind besides pixels, per image of a BATCH (jsnoop_batch_side_outputs / jsnoop_batch_log /
jsnoop_batch_export_tiff) -- what the per-file pass of the reference's batch loop produces (CJPEGsnoopCore::DoBatchFileProcess,
source/JPEGsnoopCore.cpp:765-845; log body source/ImgDecode.cpp:3021-3025, :3126-3135, :3630-3745).

All 43 golden files (well-formed, corrupted scans, hostile headers) go through ONE batch; every image's record must equal what the
compiled reference produced for that file on its own (tests/golden/manifest.json): MCU file map, block-DC maps, Huffman histogram,
status words, brightest pixel / average Y, the log text under two option sets and in quiet mode, and the three TIFF exports."""
import os
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PREFIX = ("", "W:", "E:")


def _golden_batch(harness, gpu, decode_ac=True, force_exact=False):
    import jpegsnoop_amd as J
    from golden_util import load_case, manifest
    M = manifest()
    names = sorted(M["cases"])
    b = J.JpegBatch(want_planes=True, decode_ac=decode_ac, force_exact=force_exact)
    b.enable_log(True)
    for name in names:
        data = load_case(name)
        p = harness.parse_jpeg(data)
        harness.push_tables(gpu, p)                       # the setter calls CjfifDecode makes for this header, on a table-state object
        assert b.add(gpu.h, data, p.scan_start) == len(b) - 1, name
    b.upload(); b.decode(); b.sync()
    return M, names, b


def _record(harness, b, i):
    so = b.side_outputs(i)
    keys = ("scan_bad", "scan_end", "restart_read", "num_pixels", "pos0", "align", "warn_bad", "first")
    return {"mcu_map": harness.hash_bytes(np.ascontiguousarray(so["mcu_map"])),
            "blk_dc": [harness.hash_bytes(np.ascontiguousarray(p)) if p is not None else None for p in so["blk_dc"]],
            "histo": harness.hash_bytes(np.ascontiguousarray(so["dht_histo"])),
            "status": dict(zip(keys, (int(v) for v in so["status"].values()))),
            "bright_avg": [int(v) for v in so["bright_avg"]]}


@pytest.mark.parametrize("mode,decode_ac", [("full_idct", True), ("dc_only", False)])
def test_batch_side_outputs_equal_the_single_file_records(harness, gpu, mode, decode_ac):
    M, names, b = _golden_batch(harness, gpu, decode_ac=decode_ac)
    try:
        paths = set()
        for i, name in enumerate(names):
            want = M["cases"][name][mode]
            got = _record(harness, b, i)
            assert harness.hash_bytes(b.dib(i)) == want["dib"], name
            for k in got:
                assert got[k] == want[k], (name, k, got[k], want[k])
            assert _record(harness, b, i) == got, name            # asking twice gives the same answer (the pass does not run, or log, twice)
            paths.add(b.info(i)["path"])
        assert paths == {1, 2}, "the batch must hold images of both the parallel and the exact-mirror path"
    finally:
        b.close()


def test_batch_log_text_equals_the_single_file_log(harness, gpu):
    M, names, b = _golden_batch(harness, gpu)
    try:
        bad = []
        for i, name in enumerate(names):
            want = M["cases"][name]["log"]
            for key, kw in (("plain", {}), ("histo", dict(histo_en=True)), ("quiet", dict(quiet=True))):
                got = [PREFIX[min(max(l, 0), 2)] + t for l, t in b.log_lines(i, **kw)]
                if got != want[key]:
                    k = next((j for j, (x, y) in enumerate(zip(got, want[key])) if x != y), min(len(got), len(want[key])))
                    bad.append(f"{name} [{key}] line {k}: got {got[k] if k < len(got) else None!r} want {want[key][k] if k < len(want[key]) else None!r} ({len(got)} vs {len(want[key])})")
        assert not bad, "\n".join(bad[:20]) + f"\n... {len(bad)} mismatching logs"
    finally:
        b.close()


def test_batch_log_dc_only_and_tiff(harness, gpu):
    M, names, b = _golden_batch(harness, gpu, decode_ac=False)
    try:
        for i, name in enumerate(names):
            got = [PREFIX[min(max(l, 0), 2)] + t for l, t in b.log_lines(i)]
            assert got == M["cases"][name]["log"]["dc_only"], name
    finally:
        b.close()
    M, names, b = _golden_batch(harness, gpu)
    try:
        n_tiff = 0
        with tempfile.TemporaryDirectory() as td:
            for i, name in enumerate(names):
                if "tiff" not in M["cases"][name]:
                    continue
                for key, mode in (("rgb8", 0), ("rgb16", 1), ("ycc8", 2)):
                    want = M["cases"][name]["tiff"][key]
                    path = os.path.join(td, f"{i}_{key}.tif")
                    try:
                        b.export_tiff(i, path, mode)
                        got = open(path, "rb").read()
                    except RuntimeError:
                        got = None
                    assert (got is None) == (want is None), (name, key)
                    if want is not None:
                        assert harness.hash_bytes(got) == want, (name, key)
                        n_tiff += 1
        assert n_tiff >= 9
    finally:
        b.close()


def test_batch_results_refuse_what_they_cannot_answer(harness, gpu):
    import jpegsnoop_amd as J
    f = harness.synth_jpeg(width=160, height=120, seed=5)
    b = J.JpegBatch(want_planes=False)
    b.add_jpeg(f)
    with pytest.raises(RuntimeError, match="not been decoded"):
        b.side_outputs(0)
    b.upload(); b.decode(); b.sync()
    with pytest.raises(RuntimeError, match="want_planes"):
        b.side_outputs(0)                                          # brightest pixel of a three-component image needs the planes
    so = b.side_outputs(0, bright=False)                            # ... everything else does not
    assert so["mcu_map"].shape == (8, 10) and int(so["dht_histo"].sum()) > 0
    with pytest.raises(RuntimeError, match="enable_log"):
        b.log_lines(0)
    with pytest.raises(RuntimeError, match="out of range"):
        b.side_outputs(3)
    b.close()
