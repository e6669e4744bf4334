"""oracle/harness.py -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

Python glue shared by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg:

* ``synth_jpeg``      deterministic synthetic JPEG bytes (oracle/libjsnoop_synth.so)
* ``parse_jpeg``      a tiny marker walk that extracts exactly what the reference's
                      CjfifDecode hands to CimgDecode (reference source/JfifDecode.cpp:3401-3612
                      DHT, :4576-4650 DQT, :4802-5026 SOF, :5105-5182 SOS, :5310-5324 DRI)
* ``Backend``         uniform ctypes view of the three implementations that share one
                      C entry-point set: the compiled reference (``jsref_*``), the in-repo
                      C restatement (``orc_*``) and -- through jpegsnoop_amd -- the HIP path
* ``drive``           replays the reference's setter call sequence on a backend and
                      runs DecodeScanImg
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(HERE, "_ref", "libjsnoop_ref.so")
_ALT = os.environ.get("JSNOOP_ORACLE_DIR")              # sanitizer builds of the checker libraries (tools/sanitize/run_sanitizers.sh)
ORC_SO = os.path.join(_ALT or HERE, "liboracle_imgdecode.so")
SYNTH_SO = os.path.join(_ALT or HERE, "libjsnoop_synth.so")

ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34,
          27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
          58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
UNZIGZAG = [0] * 64
for _k, _n in enumerate(ZIGZAG):
    UNZIGZAG[_n] = _k


def build(targets=("oracle", "synth", "ref")) -> None:
    subprocess.check_call(["make", "-s", "-C", HERE, *targets])


def fnv1a64(data) -> int:
    """FNV-1a 64 over a bytes-like object (vectorised in chunks; pure integer arithmetic)."""
    a = np.frombuffer(memoryview(data).cast("B"), dtype=np.uint8)
    h = 0xCBF29CE484222325
    for b in a.tobytes():          # small inputs only; large buffers use hash_bytes()
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def hash_bytes(data) -> str:
    """Content hash used by the golden fixtures (sha256 hex)."""
    import hashlib
    return hashlib.sha256(memoryview(data).cast("B")).hexdigest()


# --------------------------------------------------------------------------- synth
class _SynthParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                "width height hs vs quality restart_interval gray optimize_huffman progressive noise_sigma".split()] + \
               [("seed", C.c_uint32)]


_synth = None


def _synth_lib():
    global _synth
    if _synth is None:
        _synth = C.CDLL(SYNTH_SO)
        _synth.jsynth_encode.restype = C.c_size_t
        _synth.jsynth_encode.argtypes = [C.POINTER(_SynthParams), C.c_void_p, C.c_size_t]
        _synth.jsynth_image_rgb.argtypes = [C.POINTER(_SynthParams), C.c_void_p]
    return _synth


def synth_jpeg(width=640, height=480, hs=2, vs=2, quality=85, restart_interval=0, gray=0,
               optimize_huffman=0, progressive=0, noise_sigma=12, seed=1) -> bytes:
    p = _SynthParams(width, height, hs, vs, quality, restart_interval, gray, optimize_huffman,
                     progressive, noise_sigma, seed)
    cap = width * height * 3 + 65536
    buf = np.empty(cap, np.uint8)
    n = _synth_lib().jsynth_encode(C.byref(p), buf.ctypes.data, cap)
    if n > cap:
        buf = np.empty(n, np.uint8)
        n = _synth_lib().jsynth_encode(C.byref(p), buf.ctypes.data, n)
    return buf[:n].tobytes()


# --------------------------------------------------------------------------- parse
@dataclass
class ParsedJpeg:
    dqt: dict = field(default_factory=dict)        # tq -> list[64] natural order
    dht: dict = field(default_factory=dict)        # (class, th) -> (counts[16], values)
    sof: int = 0
    precision: int = 8
    x: int = 0
    y: int = 0
    comps: list = field(default_factory=list)      # [(ident, H, V, Tq)]
    scan_comps: list = field(default_factory=list) # [(selector, Td, Ta)]
    ss: int = 0
    se: int = 63
    ahal: int = 0
    rst_en: bool = False
    rst_interval: int = 0
    scan_start: int = 0
    scan_end: int = 0                              # offset of the marker that ends the first scan


def parse_jpeg(data: bytes) -> ParsedJpeg:
    """Walks markers up to and including the first SOS (the only scan the reference decodes,
    reference source/ImgDecode.h:23)."""
    p = ParsedJpeg()
    pos = 2
    n = len(data)
    while pos + 4 <= n:
        if data[pos] != 0xFF:
            pos += 1
            continue
        m = data[pos + 1]
        if m == 0xFF:
            pos += 1
            continue
        pos += 2
        if m in (0xD8, 0x01) or 0xD0 <= m <= 0xD7:
            continue
        if m == 0xD9:
            break
        ln = (data[pos] << 8) | data[pos + 1]
        seg = data[pos + 2: pos + ln]
        if m == 0xDB:
            i = 0
            while i < len(seg):
                pq, tq = seg[i] >> 4, seg[i] & 15
                i += 1
                tbl = [0] * 64
                for k in range(64):
                    if pq:
                        v = (seg[i] << 8) | seg[i + 1]
                        i += 2
                    else:
                        v = seg[i]
                        i += 1
                    tbl[ZIGZAG[k]] = v
                p.dqt[tq] = tbl
        elif m == 0xC4:
            i = 0
            while i < len(seg):
                tc, th = seg[i] >> 4, seg[i] & 15
                counts = list(seg[i + 1: i + 17])
                tot = sum(counts)
                vals = list(seg[i + 17: i + 17 + tot])
                p.dht[(tc, th)] = (counts, vals)
                i += 17 + tot
        elif m in (0xC0, 0xC1, 0xC2):
            p.sof = m
            p.precision = seg[0]
            p.y = (seg[1] << 8) | seg[2]
            p.x = (seg[3] << 8) | seg[4]
            nf = seg[5]
            p.comps = [(seg[6 + 3 * c], seg[7 + 3 * c] >> 4, seg[7 + 3 * c] & 15, seg[8 + 3 * c]) for c in range(nf)]
        elif m == 0xDD:
            p.rst_interval = (seg[0] << 8) | seg[1]
            p.rst_en = p.rst_interval != 0
        elif m == 0xDA:
            ns = seg[0]
            p.scan_comps = [(seg[1 + 2 * c], seg[2 + 2 * c] >> 4, seg[2 + 2 * c] & 15) for c in range(ns)]
            p.ss, p.se, p.ahal = seg[1 + 2 * ns], seg[2 + 2 * ns], seg[3 + 2 * ns]
            p.scan_start = pos + ln
            q = p.scan_start
            while q + 1 < n:
                if data[q] == 0xFF and data[q + 1] != 0 and not (0xD0 <= data[q + 1] <= 0xD7):
                    break
                q += 1
            p.scan_end = q
            return p
        pos += ln
    return p


# ------------------------------------------------------------------------- backends
STATS_WORDS = 2482      # 37 PixelCcHisto + 13 PixelCcClip + 3 x 128 RGB bins + 2048 Y bins


class Backend:
    """ctypes view of one implementation of the shared entry-point set."""

    def __init__(self, lib: C.CDLL, prefix: str, name: str):
        self.lib, self.prefix, self.name = lib, prefix, name
        f = self._f
        f("create").restype = C.c_void_p
        f("create").argtypes = []
        for fn, args in (
            ("destroy", [C.c_void_p]), ("reset_state", [C.c_void_p]), ("reset", [C.c_void_p]),
            ("set_dqt_entry", [C.c_void_p] + [C.c_uint] * 4), ("set_dqt_tables", [C.c_void_p] + [C.c_uint] * 2),
            ("set_dht_entry", [C.c_void_p] + [C.c_uint] * 7), ("set_dht_size", [C.c_void_p] + [C.c_uint] * 3),
            ("set_dht_tables", [C.c_void_p] + [C.c_uint] * 3), ("set_sof_samp_factors", [C.c_void_p] + [C.c_uint] * 3),
            ("set_precision", [C.c_void_p, C.c_uint]),
            ("set_image_details", [C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_int, C.c_uint]),
            ("decode_scan_img", [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint, C.c_int, C.c_int]),
            ("is_preview_ready", [C.c_void_p]),
            ("get_image_size", [C.c_void_p, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]),
            ("get_pixmap_ptrs", [C.c_void_p] + [C.POINTER(C.c_void_p)] * 3),
            ("get_geometry", [C.c_void_p, C.POINTER(C.c_uint)]),
            ("blk_dc_ptrs", [C.c_void_p] + [C.POINTER(C.c_void_p)] * 3),
            ("scan_status", [C.c_void_p, C.POINTER(C.c_uint)]),
            ("bright_avg", [C.c_void_p, C.POINTER(C.c_int)]),
            ("lookup_file_pos_mcu", [C.c_void_p, C.c_uint, C.c_uint, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]),
            ("lookup_blk_ycc", [C.c_void_p, C.c_uint, C.c_uint] + [C.POINTER(C.c_int)] * 3),
        ):
            f(fn).argtypes = args
        for fn in ("set_dqt_entry", "set_dqt_tables", "set_dht_entry", "set_dht_size", "set_dht_tables", "is_preview_ready"):
            f(fn).restype = C.c_int
        for fn in ("get_bitmap_ptr", "mcu_file_map", "dht_histo", "idct_lut", "dht_lookupfast"):
            f(fn).restype = C.c_void_p
            f(fn).argtypes = [C.c_void_p]
        f("idct_block").argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        f("set_preview_mode").argtypes = [C.c_void_p, C.c_uint]
        f("set_preview_ycc_offset").argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_int]
        f("get_color_stats" if prefix == "jsnoop_" else "color_stats").argtypes = [C.c_void_p, C.c_void_p]
        self.h = f("create")()
        self._log = []
        if prefix == "jsnoop_":           # CDocLog replacement: collect (level, text) like the shim's ShimLog does for the reference
            LOGFN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_char_p)
            self._log_cb = LOGFN(lambda _u, level, text: self._log.append(("", "W:", "E:")[min(max(level, 0), 2)] + text.decode()))
            f("set_log_callback").argtypes = [C.c_void_p, LOGFN, C.c_void_p]
            f("set_log_callback")(self.h, self._log_cb, None)
        elif prefix == "jsref_":
            lib.jsref_log_count.restype = C.c_size_t
            lib.jsref_log_line.restype = C.c_char_p
            lib.jsref_log_line.argtypes = [C.c_size_t]

    def _f(self, name):
        return getattr(self.lib, self.prefix + name)

    def __getattr__(self, name):          # backend.set_precision(8) -> prefix_set_precision(handle, 8)
        fn = self._f(name)
        return lambda *a: fn(self.h, *a)

    def set_options(self, decode_ac=1, histo_en=0, stat_clip_en=0, err_max=20):
        fn = self._f("set_options")
        if self.prefix == "jsref_":       # the reference keeps its options in one global config object
            fn(int(decode_ac), int(histo_en), int(stat_clip_en), C.c_uint(err_max))
        else:
            fn(C.c_void_p(self.h), int(decode_ac), int(histo_en), int(stat_clip_en), C.c_uint(err_max))

    def set_dump_histo_y(self, on):
        """bDumpHistoY: ReportHistogramY's 256 lines at the end of the decode log (reference and HIP library; the oracle keeps no log)."""
        if self.prefix == "jsref_":
            self.lib.jsref_set_dump_histo_y(int(on))
        elif self.prefix == "jsnoop_":
            self.lib.jsnoop_set_dump_histo_y(C.c_void_p(self.h), int(on))

    def close(self):
        if self.h:
            self._f("destroy")(C.c_void_p(self.h))
            self.h = None

    # ---- results as numpy copies -------------------------------------------------
    def image_size(self):
        x, y = C.c_uint(), C.c_uint()
        self._f("get_image_size")(self.h, C.byref(x), C.byref(y))
        return x.value, y.value

    def geometry(self):
        g = (C.c_uint * 8)()
        self._f("get_geometry")(self.h, g)
        return list(g)

    def dib(self):
        x, y = self.image_size()
        p = self._f("get_bitmap_ptr")(self.h)
        if not p or not x or not y:
            return None
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(y, x, 4)).copy()

    def planes(self):
        g = self.geometry()
        w, h = g[4] * 8, g[5] * 8
        ptrs = [C.c_void_p() for _ in range(3)]
        self._f("get_pixmap_ptrs")(self.h, *[C.byref(q) for q in ptrs])
        return [np.ctypeslib.as_array(C.cast(q, C.POINTER(C.c_int16)), shape=(h, w)).copy() if q.value else None
                for q in ptrs]

    def mcu_map(self):
        g = self.geometry()
        p = self._f("mcu_file_map")(self.h)
        if not p:
            return None
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(g[3], g[2])).copy()

    def blk_dc(self):
        g = self.geometry()
        ptrs = [C.c_void_p() for _ in range(3)]
        self._f("blk_dc_ptrs")(self.h, *[C.byref(q) for q in ptrs])
        return [np.ctypeslib.as_array(C.cast(q, C.POINTER(C.c_int16)), shape=(g[5], g[4])).copy() if q.value else None
                for q in ptrs]

    def dht_histo(self):
        p = self._f("dht_histo")(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(2, 4, 17)).copy()

    def status(self):
        s = (C.c_uint * 8)()
        self._f("scan_status")(self.h, s)
        return dict(zip(("scan_bad", "scan_end", "restart_read", "num_pixels", "pos0", "align", "warn_bad", "first"), s))

    def bright_avg(self):
        s = (C.c_int * 10)()
        self._f("bright_avg")(self.h, s)
        return list(s)

    def export_tiff(self, mode):
        """Export-to-TIFF bytes of the decoded image (mode 0 RGB8, 1 RGB16, 2 YCC8), or None when the backend refuses."""
        import tempfile
        fn = self._f("export_tiff")
        fn.argtypes = [C.c_void_p, C.c_char_p, C.c_int]; fn.restype = C.c_int
        with tempfile.NamedTemporaryFile(suffix=".tif") as t:
            if fn(self.h, t.name.encode(), int(mode)) != 0:
                return None
            return open(t.name, "rb").read()

    def decode_progressive(self, data: bytes) -> int:
        """HIP path only: decode a progressive (SOF2) file, all scans; returns the number of scans or -1."""
        fn = self._f("decode_progressive")
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]; fn.restype = C.c_int
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
        self.log_reset()
        return fn(self.h, C.cast(buf, C.c_void_p), len(data))

    def log_reset(self):
        self._log.clear()

    def log_lines(self):
        """Text written to the log since the last decode: 'W:' / 'E:' prefix AddLineWarn / AddLineErr lines."""
        if self.prefix == "jsref_":
            return [self.lib.jsref_log_line(i).decode() for i in range(self.lib.jsref_log_count())]
        return list(self._log)

    def color_stats(self):
        """bHistoEn / bStatClipEn statistics: dict of the PixelCcHisto ints, the PixelCcClip counters and the histograms."""
        o = np.zeros(STATS_WORDS, np.uint32)
        self._f("get_color_stats" if self.prefix == "jsnoop_" else "color_stats")(self.h, o.ctypes.data)
        return {"histo": o[:36].view(np.int32).copy(), "count": int(o[36]), "clip": o[37:50].copy(),
                "rgb": o[50:434].reshape(3, 128).copy(), "yfull": o[434:2482].copy()}

    def idct_lut(self):
        p = self._f("idct_lut")(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(64, 64)).copy()

    def lookupfast(self):
        p = self._f("dht_lookupfast")(self.h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(2, 4, 1024)).copy()

    def idct_block(self, coef64):
        c = np.ascontiguousarray(coef64, np.int16)
        o = np.empty(64, np.float32)
        self._f("idct_block")(self.h, c.ctypes.data, o.ctypes.data)
        return o


def have_ref() -> bool:
    return os.path.exists(REF_SO)


def ref_backend() -> Backend:
    return Backend(C.CDLL(REF_SO), "jsref_", "reference")


def oracle_backend() -> Backend:
    lib = C.CDLL(ORC_SO)
    lib.orc_coef_ptr.restype = C.c_void_p
    lib.orc_coef_ptr.argtypes = [C.c_void_p]
    lib.orc_coef_blocks.restype = C.c_size_t
    lib.orc_coef_blocks.argtypes = [C.c_void_p]
    lib.orc_color_exhaustive_fnv.restype = C.c_uint64
    return Backend(lib, "orc_", "oracle")


def oracle_coefs(b: Backend):
    n = b.lib.orc_coef_blocks(b.h)
    p = b.lib.orc_coef_ptr(b.h)
    return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int16)), shape=(n, 64)).copy()


# ---------------------------------------------------------------------------- drive
def push_tables(b: Backend, p: ParsedJpeg) -> None:
    """The call sequence CjfifDecode makes while walking the header (see module docstring)."""
    b.reset_state()
    for tq, tbl in sorted(p.dqt.items()):
        for nat in range(64):
            b.set_dqt_entry(tq, nat, UNZIGZAG[nat], tbl[nat])
    for ci, (_ident, h, v, tq) in enumerate(p.comps, start=1):
        b.set_dqt_tables(ci, tq)
        b.set_precision(p.precision)
    for ci, (_ident, h, v, tq) in enumerate(p.comps, start=1):
        b.set_sof_samp_factors(ci, h, v)
    for (tc, th), (counts, vals) in p.dht.items():
        code, ind, k = 0, 0, 0
        for ln in range(1, 17):
            for _ in range(counts[ln - 1]):
                mask = (((1 << ln) - 1) << (32 - ln)) & 0xFFFFFFFF
                b.set_dht_entry(th, tc, ind, ln, (code << (32 - ln)) & 0xFFFFFFFF, mask, vals[k])
                ind += 1
                code += 1
                k += 1
            code <<= 1
        b.set_dht_size(th, tc, ind)
    for si, (_sel, td, ta) in enumerate(p.scan_comps, start=1):
        b.set_dht_tables(si, td, ta)
    b.set_image_details(p.x, p.y, len(p.comps), len(p.scan_comps), int(p.rst_en), p.rst_interval)


def drive(b: Backend, data: bytes, parsed: ParsedJpeg | None = None, display=1, quiet=1):
    p = parsed or parse_jpeg(data)
    push_tables(b, p)
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    b.log_reset()
    b.decode_scan_img(C.cast(buf, C.c_void_p), len(data), p.scan_start, display, quiet)
    return p
