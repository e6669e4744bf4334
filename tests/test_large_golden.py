"""Large pictures against digests recorded from the COMPILED REFERENCE (tests/golden/large_manifest.json, written by
tests/golden/make_large.py where /root/reference exists): 640x480 ... 3840x2160, odd sizes, long restart-marker streams.
The .jpg files are not committed -- oracle/jpeg_synth.c reproduces them from the recorded parameters, and their SHA-256 is
checked first.  CPU: pins the oracle at the sizes the bench runs (BASELINE configs 1, 2, 3, 5).  GPU: the HIP path against the
same digests, through the single-image API and as one mixed batch -- no reference, no oracle in the comparison."""
import hashlib
import json
import os

import numpy as np
import pytest

M = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "large_manifest.json")))


def _digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _record(backend):
    return {"dib": _digest(backend.dib()), "planes": [_digest(p) if p is not None else None for p in backend.planes()],
            "mcu_map": _digest(backend.mcu_map()), "blk_dc": [_digest(p) if p is not None else None for p in backend.blk_dc()],
            "status": {k: int(v) for k, v in backend.status().items()}, "bright_avg": [int(v) for v in backend.bright_avg()]}


def _file(harness, name):
    data = harness.synth_jpeg(**M[name]["params"])
    assert hashlib.sha256(data).hexdigest() == M[name]["jpeg_sha256"] and len(data) == M[name]["jpeg_bytes"], "the generator no longer reproduces " + name
    return data


@pytest.mark.parametrize("name", sorted(M))
def test_oracle_matches_the_reference_on_large_pictures(harness, oracle, name):
    harness.drive(oracle, _file(harness, name))
    got = _record(oracle)
    for k, v in got.items():
        assert v == M[name][k], (name, k)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(M))
def test_gpu_matches_the_reference_on_large_pictures(harness, gpu, name):
    harness.drive(gpu, _file(harness, name))
    got = _record(gpu)
    for k, v in got.items():
        assert v == M[name][k], (name, k)


@pytest.mark.gpu
def test_gpu_batch_of_the_large_pictures(harness):
    import jpegsnoop_amd as J
    names = sorted(M)
    b = J.JpegBatch(want_planes=True)
    for n in names:
        b.add_jpeg(_file(harness, n))
    b.tile(2 * len(names))
    b.upload(); b.decode(); b.sync()
    for i in range(2 * len(names)):
        n = names[i % len(names)]
        assert b.info(i)["flags"] == 0 and b.info(i)["path"] == 1, n
        assert _digest(b.dib(i)) == M[n]["dib"], n
        for got, want in zip(b.planes(i), M[n]["planes"]):
            if want is not None:
                assert _digest(got) == want, n
    b.close()
