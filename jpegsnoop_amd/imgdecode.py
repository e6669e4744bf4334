"""Host-side mirror of the reference's operator interface for the scan-decode path.

`CimgDecode` keeps the reference's method names, argument meaning and error behaviour
(reference source/ImgDecode.h:286-356) on top of the C ABI; `JpegBatch` is the batched
submit that replaces the strictly sequential per-file loop of
CJPEGsnoopCore::DoBatchFileProcess (source/JPEGsnoopCore.cpp:765).  Python is only the
binding layer here: decode happens in libjsnoop_gpu.so on the GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


class CimgDecode:
    """One decoder object == one `CimgDecode` of the reference."""

    PREVIEW_RGB, PREVIEW_YCC, PREVIEW_R, PREVIEW_G, PREVIEW_B, PREVIEW_Y, PREVIEW_CB, PREVIEW_CR = range(1, 9)

    def __init__(self, log=None):
        self._lib = capi.load()
        self._h = self._lib.jsnoop_create()
        if not self._h:
            raise RuntimeError("jsnoop_create failed: " + capi.last_error())
        self._log_cb = None
        if log is not None:
            self._log_cb = capi.LOG_FN(lambda _u, lvl, txt: log(lvl, txt.decode(errors="replace")))
            self._lib.jsnoop_set_log_callback(self._h, self._log_cb, None)
        self._buf = None

    def close(self):
        if self._h:
            self._lib.jsnoop_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- lifecycle / options -------------------------------------------------------
    def Reset(self): self._lib.jsnoop_reset(self._h)
    def ResetState(self): self._lib.jsnoop_reset_state(self._h)
    def SetOptions(self, bDecodeScanImgAc=True, bHistoEn=False, bStatClipEn=False, nErrMaxDecodeScan=20):
        self._lib.jsnoop_set_options(self._h, int(bDecodeScanImgAc), int(bHistoEn), int(bStatClipEn), nErrMaxDecodeScan)

    # --- tables / geometry ------------------------------------------------------------
    def SetDqtEntry(self, nTblDestId, nCoeffInd, nCoeffIndZz, nCoeffVal):
        return bool(self._lib.jsnoop_set_dqt_entry(self._h, nTblDestId, nCoeffInd, nCoeffIndZz, nCoeffVal))
    def SetDqtTables(self, nCompInd, nTbl): return bool(self._lib.jsnoop_set_dqt_tables(self._h, nCompInd, nTbl))
    def GetDqtEntry(self, nTblDestId, nCoeffInd): return self._lib.jsnoop_get_dqt_entry(self._h, nTblDestId, nCoeffInd)
    def SetDhtEntry(self, nDestId, nClass, nInd, nLen, nBits, nMask, nCode):
        return bool(self._lib.jsnoop_set_dht_entry(self._h, nDestId, nClass, nInd, nLen, nBits, nMask, nCode))
    def SetDhtSize(self, nDestId, nClass, nSize): return bool(self._lib.jsnoop_set_dht_size(self._h, nDestId, nClass, nSize))
    def SetDhtTables(self, nCompInd, nTblDc, nTblAc): return bool(self._lib.jsnoop_set_dht_tables(self._h, nCompInd, nTblDc, nTblAc))
    def SetSofSampFactors(self, nCompInd, nSampFactH, nSampFactV): self._lib.jsnoop_set_sof_samp_factors(self._h, nCompInd, nSampFactH, nSampFactV)
    def SetPrecision(self, nPrecision): self._lib.jsnoop_set_precision(self._h, nPrecision)
    def SetImageDetails(self, nDimX, nDimY, nCompsSOF, nCompsSOS, bRstEn, nRstInterval):
        self._lib.jsnoop_set_image_details(self._h, nDimX, nDimY, nCompsSOF, nCompsSOS, int(bRstEn), nRstInterval)

    # --- decode -------------------------------------------------------------------------
    def DecodeScanImg(self, file_bytes: bytes, nStart: int, bDisplay=True, bQuiet=False):
        """`file_bytes` stands for what the reference reads through CwindowBuf::Buf."""
        self._buf = (C.c_uint8 * len(file_bytes)).from_buffer_copy(file_bytes)
        self._lib.jsnoop_decode_scan_img(self._h, C.cast(self._buf, C.c_void_p), len(file_bytes), nStart, int(bDisplay), int(bQuiet))

    def DecodeProgressive(self, file_bytes: bytes) -> int:
        """Beyond the reference (it refuses SOF2): decode every scan of a progressive file; returns the number of scans."""
        self._buf = (C.c_uint8 * len(file_bytes)).from_buffer_copy(file_bytes)
        n = self._lib.jsnoop_decode_progressive(self._h, C.cast(self._buf, C.c_void_p), len(file_bytes))
        if n < 0:
            raise RuntimeError("jsnoop_decode_progressive: " + capi.last_error())
        return n
    def GetColorStats(self):
        out = np.zeros(2482, np.uint32)
        self._lib.jsnoop_get_color_stats(self._h, out.ctypes.data)
        return out
    def ExportTiff(self, path: str, nMode: int = 0) -> bool:
        return self._lib.jsnoop_export_tiff(self._h, path.encode(), nMode) == 0

    # --- results ---------------------------------------------------------------------------
    def IsPreviewReady(self): return bool(self._lib.jsnoop_is_preview_ready(self._h))
    def GetImageSize(self):
        x, y = C.c_uint(), C.c_uint()
        self._lib.jsnoop_get_image_size(self._h, C.byref(x), C.byref(y))
        return x.value, y.value
    def GetBitmapPtr(self):
        x, y = self.GetImageSize()
        p = self._lib.jsnoop_get_bitmap_ptr(self._h)
        if not p:
            return None
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(y, x, 4))
    def GetPixMapPtrs(self):
        g = (C.c_uint * 8)()
        self._lib.jsnoop_get_geometry(self._h, g)
        ptrs = [C.c_void_p() for _ in range(3)]
        self._lib.jsnoop_get_pixmap_ptrs(self._h, *[C.byref(q) for q in ptrs])
        return [np.ctypeslib.as_array(C.cast(q, C.POINTER(C.c_int16)), shape=(g[5] * 8, g[4] * 8)) if q.value else None for q in ptrs]
    def LookupFilePosMcu(self, nMcuX, nMcuY):
        a, b = C.c_uint(), C.c_uint()
        self._lib.jsnoop_lookup_file_pos_mcu(self._h, nMcuX, nMcuY, C.byref(a), C.byref(b))
        return a.value, b.value
    def LookupFilePosPix(self, nPixX, nPixY):
        a, b = C.c_uint(), C.c_uint()
        self._lib.jsnoop_lookup_file_pos_pix(self._h, nPixX, nPixY, C.byref(a), C.byref(b))
        return a.value, b.value
    def LookupBlkYCC(self, nBlkX, nBlkY):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self._lib.jsnoop_lookup_blk_ycc(self._h, nBlkX, nBlkY, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value
    def PixelToMcu(self, nPixX, nPixY):
        a, b = C.c_uint(), C.c_uint()
        self._lib.jsnoop_pixel_to_mcu(self._h, nPixX, nPixY, C.byref(a), C.byref(b))
        return a.value, b.value
    def PixelToBlk(self, nPixX, nPixY):
        a, b = C.c_uint(), C.c_uint()
        self._lib.jsnoop_pixel_to_blk(self._h, nPixX, nPixY, C.byref(a), C.byref(b))
        return a.value, b.value
    def McuXyToLinear(self, nMcuX, nMcuY): return self._lib.jsnoop_mcu_xy_to_linear(self._h, nMcuX, nMcuY)
    def SetDumpHistoY(self, bDumpHistoY): self._lib.jsnoop_set_dump_histo_y(self._h, int(bDumpHistoY))
    # CwindowBuf overlays (source/WindowBuf.cpp:516-620): patched bytes seen by the next DecodeScanImg
    def OverlayInstall(self, data: bytes, nBegin: int) -> bool:
        buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data or b"\0")
        return bool(self._lib.jsnoop_overlay_install(self._h, C.cast(buf, C.c_void_p), len(data), nBegin))
    def OverlayRemoveAll(self): self._lib.jsnoop_overlay_remove_all(self._h)
    def OverlayGetNum(self): return self._lib.jsnoop_overlay_get_num(self._h)
    def OverlayGet(self, nOvrInd):
        p, n, b = C.c_void_p(), C.c_uint(), C.c_uint()
        if not self._lib.jsnoop_overlay_get(self._h, nOvrInd, C.byref(p), C.byref(n), C.byref(b)):
            return None
        return bytes((C.c_uint8 * n.value).from_address(p.value)), b.value
    def SetPreviewMode(self, nMode): self._lib.jsnoop_set_preview_mode(self._h, nMode)
    def GetPreviewMode(self): return self._lib.jsnoop_get_preview_mode(self._h)
    def SetPreviewYccOffset(self, nMcuX, nMcuY, nY, nCb, nCr): self._lib.jsnoop_set_preview_ycc_offset(self._h, nMcuX, nMcuY, nY, nCb, nCr)
    def LastPath(self): return self._lib.jsnoop_last_path(self._h)
    def LastFlags(self): return self._lib.jsnoop_last_flags(self._h)
    def LastSideMode(self): return self._lib.jsnoop_last_side_mode(self._h)


class JpegBatch:
    """N JPEG files -> N DIBs resident in HBM (device-side batch)."""

    def __init__(self, stream=None, decode_ac=True, want_planes=False, force_exact=False):
        self._lib = capi.load()
        self._h = self._lib.jsnoop_batch_create(C.c_void_p(stream) if stream else None)
        if not self._h:
            raise RuntimeError("jsnoop_batch_create failed: " + capi.last_error())
        self._lib.jsnoop_batch_set_options(self._h, int(decode_ac), int(want_planes), int(force_exact))
        self.want_planes = want_planes
        self._borrowed = False                   # True: the handle belongs to a JpegPipeline, which destroys it

    def close(self):
        if self._h:
            if not getattr(self, "_borrowed", False):
                self._lib.jsnoop_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc < 0:
            raise RuntimeError(f"{what} failed: {capi.last_error()}")
        return rc

    def add_jpeg(self, data: bytes) -> int:
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
        return self._chk(self._lib.jsnoop_batch_add_jpeg(self._h, C.cast(buf, C.c_void_p), len(data)), "batch_add_jpeg")

    def tile(self, total: int) -> int: return self._chk(self._lib.jsnoop_batch_tile(self._h, total), "batch_tile")
    def set_split(self, parts: int) -> None:
        """parts = 2: later decodes run the two halves of the batch on two streams side by side (same results); 1: one stream; 0: the library decides."""
        self._chk(self._lib.jsnoop_batch_set_split(self._h, parts), "batch_set_split")
    def split_parts(self) -> int: return int(self._lib.jsnoop_batch_split_parts(self._h))
    def tuning(self) -> "capi.Tuning":
        t = capi.Tuning(); self._lib.jsnoop_batch_get_tuning(self._h, C.byref(t)); return t
    def set_tuning(self, **fields) -> None:
        """Replaces fields of the batch's JsnoopTuning (e.g. sub_wl=5, cand_rounds=-1); call before upload()."""
        t = self.tuning()
        for k, v in fields.items():
            if not hasattr(t, k): raise AttributeError("JsnoopTuning has no field " + k)
            setattr(t, k, v)
        self._chk(self._lib.jsnoop_batch_set_tuning(self._h, C.byref(t)), "batch_set_tuning")
    def clear(self): self._lib.jsnoop_batch_clear(self._h)
    def __len__(self): return self._lib.jsnoop_batch_count(self._h)
    def upload(self): self._chk(self._lib.jsnoop_batch_upload(self._h), "batch_upload")
    def decode(self): self._chk(self._lib.jsnoop_batch_decode(self._h), "batch_decode")
    def sync(self): self._chk(self._lib.jsnoop_batch_sync(self._h), "batch_sync")

    def decode_timed(self, reps=1):
        st = (C.c_double * capi.NUM_STAGES)()
        ms = self._lib.jsnoop_batch_decode_timed(self._h, reps, st)
        if ms < 0:
            raise RuntimeError("batch_decode_timed failed: " + capi.last_error())
        return ms, {self._lib.jsnoop_stage_name(i).decode(): st[i] for i in range(capi.NUM_STAGES)}

    def info(self, i):
        o = (C.c_uint * 16)()
        self._chk(self._lib.jsnoop_batch_image_info(self._h, i, o), "batch_image_info")
        keys = "dim_x dim_y img_x img_y mcu_w mcu_h mcu_xmax mcu_ymax blk_xmax blk_ymax scan_bytes flags path ncomp file_len total_blocks".split()
        return dict(zip(keys, o))

    def dib(self, i):
        inf = self.info(i)
        out = np.empty((inf["img_y"], inf["img_x"], 4), np.uint8)
        self._chk(self._lib.jsnoop_batch_read_dib(self._h, i, out.ctypes.data), "batch_read_dib")
        return out

    def color_stats(self, i, histo_en=True):
        """bHistoEn (or only bStatClipEn) statistics of image i: the JSNOOP_STATS_WORDS record of include/jsnoop_gpu.h."""
        out = np.zeros(2482, np.uint32)
        self._chk(self._lib.jsnoop_batch_color_stats(self._h, i, int(histo_en), out.ctypes.data), "batch_color_stats")
        return out

    def planes(self, i):
        inf = self.info(i)
        shp = (inf["blk_ymax"] * 8, inf["blk_xmax"] * 8)
        ps = [np.zeros(shp, np.int16) for _ in range(3)]
        self._chk(self._lib.jsnoop_batch_read_planes(self._h, i, *[p.ctypes.data for p in ps]), "batch_read_planes")
        return ps[: inf["ncomp"]]

    def coefs(self, i):
        inf = self.info(i)
        out = np.empty((inf["total_blocks"], 64), np.int16)
        self._chk(self._lib.jsnoop_batch_read_coefs(self._h, i, out.ctypes.data, inf["total_blocks"]), "batch_read_coefs")
        return out

    def dib_checksums(self):
        out = np.zeros(len(self), np.uint64)
        self._chk(self._lib.jsnoop_batch_dib_hashes(self._h, out.ctypes.data), "batch_dib_hashes")
        return out

    # --- what DecodeScanImg leaves behind besides pixels, per image (the per-file pass of DoBatchFileProcess) -----------------
    def add(self, tables: "CimgDecode", data: bytes, scan_start: int) -> int:
        """Adds one image under the table / frame state `tables` holds (after the setter calls of its header), like jsnoop_batch_add."""
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
        h = getattr(tables, "_h", None) or getattr(tables, "h", None) or tables
        return self._chk(self._lib.jsnoop_batch_add(self._h, h, C.cast(buf, C.c_void_p), len(data), scan_start), "batch_add")

    def enable_log(self, on=True): self._chk(self._lib.jsnoop_batch_enable_log(self._h, int(on)), "batch_enable_log")

    def side_outputs(self, i, bright=True):
        """MCU file map, block-DC maps, Huffman code-length histogram, status words, brightest pixel / average Y of image i."""
        inf = self.info(i)
        nmcu, nblk = inf["mcu_xmax"] * inf["mcu_ymax"], inf["blk_xmax"] * inf["blk_ymax"]
        mcu = np.zeros(nmcu, np.uint32)
        dcs = [np.zeros(nblk, np.int16) for _ in range(3)]
        histo = np.zeros(2 * 4 * 17, np.uint32)
        st, ba = (C.c_uint * 8)(), (C.c_int * 10)()
        self._chk(self._lib.jsnoop_batch_side_outputs(self._h, i, mcu.ctypes.data, dcs[0].ctypes.data, dcs[1].ctypes.data, dcs[2].ctypes.data, histo.ctypes.data,
                                                      st, ba if bright else None), "batch_side_outputs")
        keys = "scan_bad scan_end rst_count num_pixels pos0 align warn_bad first".split()
        return {"mcu_map": mcu.reshape(inf["mcu_ymax"], inf["mcu_xmax"]),
                "blk_dc": [d.reshape(inf["blk_ymax"], inf["blk_xmax"]) if (c == 0 or inf["ncomp"] == 3) else None for c, d in enumerate(dcs)],
                "dht_histo": histo.reshape(2, 4, 17), "status": dict(zip(keys, st)), "bright_avg": list(ba) if bright else None}

    def log_lines(self, i, histo_en=False, stat_clip_en=False, quiet=False):
        """The text DecodeScanImg writes to CDocLog for image i, as (level, line) pairs."""
        out = []
        cb = capi.LOG_FN(lambda _u, lvl, txt: out.append((lvl, txt.decode(errors="replace"))))
        self._chk(self._lib.jsnoop_batch_log(self._h, i, int(histo_en), int(stat_clip_en), int(quiet), cb, None), "batch_log")
        return out

    def export_tiff(self, i, path: str, mode: int = 0):
        self._chk(self._lib.jsnoop_batch_export_tiff(self._h, i, path.encode(), mode), "batch_export_tiff")

    def algorithmic_bytes(self): return int(self._lib.jsnoop_batch_algorithmic_bytes(self._h))
    def pixels(self): return int(self._lib.jsnoop_batch_pixels(self._h))


class JpegPipeline:
    """Overlapped staging (the CwindowBuf replacement at batch scale): `slots` JpegBatch slots cycled by jsnoop_pipeline_run --
    H2D of the next batch and, on request, D2H of the previous one overlap the decode of the current one."""

    def __init__(self, slots=2):
        self._lib = capi.load()
        self._h = self._lib.jsnoop_pipeline_create(slots)
        if not self._h:
            raise RuntimeError("jsnoop_pipeline_create failed: " + capi.last_error())
        self.slots = []
        for i in range(slots):
            b = JpegBatch.__new__(JpegBatch)
            b._lib, b._h, b.want_planes, b._borrowed = self._lib, self._lib.jsnoop_pipeline_slot(self._h, i), False, True
            self.slots.append(b)

    def run(self, batches, d2h=False):
        out = (C.c_double * 6)()
        if self._lib.jsnoop_pipeline_run(self._h, batches, int(d2h), out) < 0:
            raise RuntimeError("jsnoop_pipeline_run failed: " + capi.last_error())
        return {"ms_per_batch": out[0], "h2d_ms": out[1], "decode_ms": out[2], "d2h_ms": out[3], "compressed_bytes": int(out[4]), "dib_bytes": int(out[5])}

    def close(self):
        if self._h:
            for b in self.slots:
                b._h = None                      # owned by the pipeline
            self._lib.jsnoop_pipeline_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def dib_checksum_numpy(dib: np.ndarray) -> int:
    """The position-keyed checksum of k_dib_checksum, recomputed on the host from a DIB array
    (used by tests to compare a device DIB against the oracle's without a D2H copy)."""
    px = np.ascontiguousarray(dib).view(np.uint32).reshape(-1).astype(np.uint64)
    z = (np.arange(px.size, dtype=np.uint64) << np.uint64(32)) | px
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
        return int(z.sum(dtype=np.uint64))
