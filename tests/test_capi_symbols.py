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


def test_tuning_defaults_come_from_the_environment_once_and_nothing_on_the_decode_path_reads_it(lib):
    """JsnoopTuning (include/jsnoop_gpu.h): the environment variables of tools/README.md only preset the struct jsnoop_tuning_defaults
    returns -- per process, read once; the sources of the library hold no getenv outside that one function."""
    import ctypes as C
    import subprocess
    import sys
    from jpegsnoop_amd import capi
    t = capi.Tuning(); lib.jsnoop_tuning_defaults(C.byref(t))
    assert t.struct_size == C.sizeof(capi.Tuning)
    child = ("import sys, ctypes as C; sys.path.insert(0, %r); from jpegsnoop_amd import capi; lib = capi.load(require_device=False); "
             "t = capi.Tuning(); lib.jsnoop_tuning_defaults(C.byref(t)); "
             "print(t.sub_wl, t.cand_rounds, t.cand_max_walks, t.sync_launches, t.write_lanes, t.split, t.mcus_per_wave, t.pg_lanes, t.cross_checks, t.debug)") % ROOT
    env = {k: v for k, v in os.environ.items() if not k.startswith("JSNOOP_")}
    out = subprocess.check_output([sys.executable, "-c", child], env=env).decode().split()
    assert [int(x) for x in out] == [0] * 10                                        # no variable set: every field automatic
    env.update(JSNOOP_SUB_WL="6", JSNOOP_CAND="0", JSNOOP_CAND_LANES="123", JSNOOP_SYNC_LAUNCHES="3", JSNOOP_NO_HALF="1", JSNOOP_SPLIT="2", JSNOOP_MPW="16",
               JSNOOP_PG_LANES="64", JSNOOP_WRITE_V1="1", JSNOOP_SIDE_EXACT="1", JSNOOP_DEBUG_CAND="2", JSNOOP_DEBUG_TIMING="1")
    out = subprocess.check_output([sys.executable, "-c", child], env=env).decode().split()
    assert [int(x) for x in out] == [6, -1, 123, 3, 1, 2, 16, 64, capi.XC_WRITE_V1 | capi.XC_SIDE_EXACT, capi.DBG_CAND | capi.DBG_CAND_LINKS | capi.DBG_TIMING]
    hits = []
    for f in os.listdir(os.path.join(ROOT, "jpegsnoop_amd", "csrc")):
        if f.endswith((".cpp", ".hip", ".h")):
            for n, line in enumerate(open(os.path.join(ROOT, "jpegsnoop_amd", "csrc", f), errors="replace"), 1):
                if "getenv(" in line:
                    hits.append((f, n))
    assert hits and all(f == "jsnoop_host.cpp" for f, _ in hits) and len(hits) == 2, hits       # the two lambdas of js_env_tuning
    # presets out of range fall back to "automatic" (a later set_tuning writes the whole struct back: it must not be refused for them)
    env.update(JSNOOP_SYNC_LAUNCHES="100", JSNOOP_MPW="8192")
    out = subprocess.check_output([sys.executable, "-c", child], env=env).decode().split()
    assert int(out[3]) == 0 and int(out[6]) == 0


def test_no_experiment_switches_in_the_product_sources():
    """Ablations and candidate rewrites are patch files under tools/variants/ (tools/build_variant.sh applies them to a private copy of the kernels):
    what ships holds no JS_EXP_ / JS_TRY_ switch -- code that compiles to wrong results does not sit in the loop a maintainer reads."""
    for d in (os.path.join(ROOT, "jpegsnoop_amd"), os.path.join(ROOT, "include")):
        for root, _dirs, files in os.walk(d):
            for f in files:
                if f.endswith((".cpp", ".hip", ".h", ".py")):
                    txt = open(os.path.join(root, f), errors="replace").read()
                    assert "JS_EXP_" not in txt and "JS_TRY_" not in txt, os.path.join(root, f)
    assert any(f.endswith(".patch") for f in os.listdir(os.path.join(ROOT, "tools", "variants")))
