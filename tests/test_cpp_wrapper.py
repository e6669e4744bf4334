"""The C++ CimgDecode-shaped wrapper (jpegsnoop_amd/csrc/ImgDecodeGpu.h) builds against the C ABI with plain g++;
on a GPU it decodes a file and its DIB matches the oracle, on CPU it fails loudly (no fallback)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "wrapper_demo")


def build_demo():
    import __graft_entry__ as G
    G.build()
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", EXE, os.path.join(ROOT, "tests", "cpp", "wrapper_demo.cpp"),
                           "-L" + os.path.join(ROOT, "jpegsnoop_amd"), "-ljsnoop_gpu", "-Wl,-rpath," + os.path.join(ROOT, "jpegsnoop_amd"),
                           "-Wl,-rpath,/opt/rocm/lib"])


SITES = os.path.join(ROOT, "tests", "cpp", "jfif_callsites")


def build_callsites():
    import __graft_entry__ as G
    G.build()
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-o", SITES, os.path.join(ROOT, "tests", "cpp", "jfif_callsites.cpp"),
                           "-L" + os.path.join(ROOT, "jpegsnoop_amd"), "-ljsnoop_gpu", "-Wl,-rpath," + os.path.join(ROOT, "jpegsnoop_amd"),
                           "-Wl,-rpath,/opt/rocm/lib"])


def test_every_call_site_of_cjfifdecode_compiles_against_the_wrapper():
    """The reference's CjfifDecode reaches its CimgDecode through twelve methods and three public members (source/JfifDecode.cpp:115 ... :7379);
    tests/cpp/jfif_callsites.cpp holds one function per call site with the reference's argument types -- it must build (-Wall -Werror) and link
    against ImgDecodeGpu.h + the C ABI."""
    build_callsites()
    assert subprocess.check_output([SITES], text=True).strip() == "built"
    src = open(os.path.join(ROOT, "tests", "cpp", "jfif_callsites.cpp")).read()
    for m in ("Reset()", "ResetState()", "SetDhtEntry(", "SetDhtSize(", "SetDqtEntry(", "SetDqtTables(", "SetPrecision(", "SetSofSampFactors(", "SetDhtTables(",
              "SetImageDetails(", "DecodeScanImg(", "SetImageDimensions(", "->m_pDibTemp", "->m_bDibTempReady", "->m_bPreviewIsJpeg"):
        assert "m_pImgDec->" + m.lstrip("->") in src or m in src, m


@pytest.mark.gpu
def test_call_sites_drive_a_psd_style_preview_and_a_scan_decode(harness, tmp_path):
    build_callsites()
    p = tmp_path / "x.jpg"
    p.write_bytes(harness.synth_jpeg(width=333, height=217, seed=78))
    out = subprocess.check_output([SITES, str(p)], text=True).strip().splitlines()
    assert out[0] == "psd ready=0 bits=1 first=1 dims=5x3", out          # IsPreviewReady() is m_bPreviewIsJpeg (:3753); GetBitmapPtr hands out what the PSD decoder wrote
    assert out[1] == "jpeg ready=1 temp_ready=1 is_jpeg=1 bits=1 dims=336x224", out


def test_wrapper_builds_and_refuses_without_gpu(tmp_path):
    import torch
    build_demo()
    if torch.cuda.is_available():
        pytest.skip("GPU visible: covered by the gpu-marked test")
    p = tmp_path / "x.jpg"
    p.write_bytes(open(os.path.join(ROOT, "tests", "golden", "c1_444_160x120.jpg"), "rb").read())
    r = subprocess.run([EXE, str(p)], capture_output=True, text=True)
    assert r.returncode == 3 and "no CPU fallback" in r.stdout


@pytest.mark.gpu
def test_wrapper_decodes_like_the_oracle(harness, oracle, tmp_path):
    build_demo()
    data = harness.synth_jpeg(width=333, height=217, seed=77)
    p = tmp_path / "x.jpg"
    p.write_bytes(data)
    # a byte overlay in the middle of the entropy-coded segment (the reference's fault-injection tool, source/WindowBuf.cpp:516-590)
    at = len(data) // 2
    patch = bytes([data[at] ^ 0x5A, data[at + 1] ^ 0x33, 0x00]) if data[at + 1] != 0xFF else bytes([0x12, 0x34, 0x56])
    out = subprocess.check_output([EXE, str(p), str(at), patch.hex()], text=True)
    patched = bytearray(data); patched[at:at + len(patch)] = patch
    harness.drive(oracle, bytes(patched))
    want_patched = "%016x" % harness.fnv1a64(oracle.dib().tobytes())
    harness.drive(oracle, data)
    want = "%016x" % harness.fnv1a64(oracle.dib().tobytes())
    lines = dict(l.split(" ", 1) for l in out.strip().splitlines())
    assert f"dib_fnv={want}" in lines["single"] and "ready=1" in lines["single"] and "size=336x224" in lines["single"]
    assert f"y0={int(oracle.planes()[0][0, 0])} " in lines["single"]
    mm = oracle.mcu_map()[0, 0]
    assert f"mcu0={mm >> 4}.{mm & 7}" in lines["single"]
    assert f"dib_fnv={want}" in lines["batch"] and "count=3" in lines["batch"]
    # the per-file pass of the batch loop through the C++ facade: log text + side outputs of a batch image
    bl = lines["batchlog"]
    assert "ok=1" in bl and "first=[*** Decoding SCAN Data ***]" in bl and "finished_at=" in bl and " finished_at=0 " not in bl
    mm0 = oracle.mcu_map()[0, 0]
    assert f"mcu0={mm0 >> 4}.{mm0 & 7}" in bl and f"pixels={int(oracle.status()['num_pixels'])}" in bl and "bright_valid=1" in bl
    # the CJPEGsnoopCore facade: AnalyzeFile, I_* accessors, overlay + re-decode
    core = lines["core"]
    assert f"dib_fnv={want}" in core and "ready=1" in core and "size=336x224" in core
    assert "mcu=6,3 lin=%d blk=12,6" % (3 * 21 + 6) in core                   # 100 / 16, 50 / 16; 21 MCUs across; 100 / 8, 50 / 8
    mm = oracle.mcu_map()[3, 6]
    assert f"pos={mm >> 4}.{mm & 7}" in core
    bd = oracle.blk_dc(); y, cb, cr = int(bd[0][6, 12]), int(bd[1][6, 12]), int(bd[2][6, 12])
    assert f"ycc={y},{cb},{cr} " in core and "mode=1" in core
    assert f"installed=1 seen={patch[0]:02x} clean={data[at]:02x} dib_fnv={want_patched}" in lines["overlay"]
    assert want_patched != want and f"dib_fnv={want}" in lines["restored"]
