"""CPU: libjsnoop_gpu.so builds, loads, and exports every entry point include/jsnoop_gpu.h declares
(no compute calls: there is no GPU here), and refuses to work without a device instead of falling back."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as G
    G.build()
    import jpegsnoop_amd
    return jpegsnoop_amd.load(require_device=False)


def header_symbols():
    src = open(os.path.join(ROOT, "include", "jsnoop_gpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(jsnoop_[a-z0-9_]+)\s*\(", src)) - {"jsnoop_log_fn"})


def test_every_declared_symbol_is_exported(lib):
    from jpegsnoop_amd import capi
    names = header_symbols()
    assert len(names) >= 55
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/jsnoop_gpu.h but not exported"
    assert sorted(capi.SIGNATURES) == names, "python binding and header disagree"
    assert lib.jsnoop_abi_version() == 1
    # ... and nothing jsnoop_* is exported that the header does not declare
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "jpegsnoop_amd", "libjsnoop_gpu.so")]).decode()
    exported = sorted(set(re.findall(r"\bT (jsnoop_[a-z0-9_]+)", out)))
    assert exported == names, "exports and header disagree"


def test_decode_table_builders_selftest(lib):
    """Host logic, no device: random canonical Huffman tables through the builders of the parallel path's decode tables (two-level
    tables, pair entries of the sync pass, value-pair entries of the write pass) against a plain search through the code list."""
    assert lib.jsnoop_selftest_tables(1, 40) == 0
    assert lib.jsnoop_selftest_tables(20260924, 40) == 0
    assert lib.jsnoop_selftest_bytes(1, 2000) == 0 and lib.jsnoop_selftest_bytes(77, 2000) == 0      # staging's marker searches (SSE2) against byte loops


def test_no_cpu_fallback(lib):
    import jpegsnoop_amd
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    assert lib.jsnoop_device_count() == 0
    assert not lib.jsnoop_create()
    assert b"no CPU fallback" in lib.jsnoop_last_error()
    with pytest.raises(RuntimeError):
        jpegsnoop_amd.load()
    with pytest.raises(RuntimeError):
        jpegsnoop_amd.CimgDecode()


def test_product_does_not_link_the_oracle(lib):
    """The shipped library must not depend on anything under oracle/."""
    import subprocess
    out = subprocess.check_output(["ldd", os.path.join(ROOT, "jpegsnoop_amd", "libjsnoop_gpu.so")]).decode()
    assert "oracle" not in out and "jsnoop_ref" not in out
    for root, _dirs, files in os.walk(os.path.join(ROOT, "jpegsnoop_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                txt = open(os.path.join(root, f), errors="replace").read()
                assert "oracle_imgdecode" not in txt and "liboracle" not in txt and "from oracle" not in txt, f
