"""ctypes binding of libjsnoop_gpu.so (the C ABI declared in include/jsnoop_gpu.h).

Plumbing only: the product is the shared library.  Loading fails loudly when the
library has not been built or no HIP device is visible -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libjsnoop_gpu.so")

NUM_STAGES = 8
LOG_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_char_p)

class Tuning(C.Structure):
    """JsnoopTuning of include/jsnoop_gpu.h: how the library decodes (0 = automatic everywhere), never what it produces."""
    _fields_ = [("struct_size", C.c_uint32), ("sub_wl", C.c_int32), ("cand_rounds", C.c_int32), ("cand_max_walks", C.c_uint64),
                ("sync_launches", C.c_int32), ("write_lanes", C.c_int32), ("split", C.c_int32), ("mcus_per_wave", C.c_int32),
                ("pg_lanes", C.c_int32), ("cross_checks", C.c_uint32), ("debug", C.c_uint32)]


XC_BACKEND_GENERIC, XC_WRITE_V1, XC_NO_TAIL, XC_SIDE_EXACT, XC_CAND_VERIFY, XC_UNSTUFF_3PASS = 1, 2, 4, 8, 16, 32
DBG_CAND, DBG_CAND_LINKS, DBG_TAIL, DBG_TIMING = 1, 2, 4, 8

_u, _i, _p, _sz = C.c_uint, C.c_int, C.c_void_p, C.c_size_t
_PU, _PI = C.POINTER(C.c_uint), C.POINTER(C.c_int)

# name -> (restype, argtypes); every symbol include/jsnoop_gpu.h declares
SIGNATURES = {
    "jsnoop_abi_version": (_i, []),
    "jsnoop_selftest_tables": (_i, [C.c_uint, C.c_uint]),
    "jsnoop_selftest_bytes": (_i, [C.c_uint, C.c_uint]),
    "jsnoop_last_error": (C.c_char_p, []),
    "jsnoop_device_count": (_i, []),
    "jsnoop_set_device": (_i, [_i]),
    "jsnoop_create": (_p, []),
    "jsnoop_destroy": (None, [_p]),
    "jsnoop_reset": (None, [_p]),
    "jsnoop_reset_state": (None, [_p]),
    "jsnoop_reset_dqt_tables": (None, [_p]),
    "jsnoop_reset_dht_lookup": (None, [_p]),
    "jsnoop_set_image_dimensions": (None, [_p, _u, _u]),
    "jsnoop_get_image_dimensions": (None, [_p, _PU, _PU]),
    "jsnoop_dib_temp_create": (_p, [_p, _u, _u]),
    "jsnoop_set_dib_temp_ready": (None, [_p, _i]),
    "jsnoop_get_dib_temp_ready": (_i, [_p]),
    "jsnoop_set_preview_is_jpeg": (None, [_p, _i]),
    "jsnoop_set_log_callback": (None, [_p, LOG_FN, _p]),
    "jsnoop_set_options": (None, [_p, _i, _i, _i, _u]),
    "jsnoop_set_dqt_entry": (_i, [_p, _u, _u, _u, _u]),
    "jsnoop_set_dqt_tables": (_i, [_p, _u, _u]),
    "jsnoop_get_dqt_entry": (_u, [_p, _u, _u]),
    "jsnoop_set_dht_entry": (_i, [_p, _u, _u, _u, _u, _u, _u, _u]),
    "jsnoop_set_dht_size": (_i, [_p, _u, _u, _u]),
    "jsnoop_set_dht_tables": (_i, [_p, _u, _u, _u]),
    "jsnoop_set_sof_samp_factors": (None, [_p, _u, _u, _u]),
    "jsnoop_set_precision": (None, [_p, _u]),
    "jsnoop_set_image_details": (None, [_p, _u, _u, _u, _u, _i, _u]),
    "jsnoop_jfif_walk": (_i, [_p, _p, _sz, _PU]),
    "jsnoop_decode_scan_img": (None, [_p, _p, _sz, _u, _i, _i]),
    "jsnoop_decode_progressive": (_i, [_p, _p, _sz]),
    "jsnoop_is_preview_ready": (_i, [_p]),
    "jsnoop_get_image_size": (None, [_p, _PU, _PU]),
    "jsnoop_get_bitmap_ptr": (_p, [_p]),
    "jsnoop_get_bitmap_dev": (_p, [_p]),
    "jsnoop_get_pixmap_ptrs": (None, [_p, C.POINTER(_p), C.POINTER(_p), C.POINTER(_p)]),
    "jsnoop_lookup_file_pos_mcu": (None, [_p, _u, _u, _PU, _PU]),
    "jsnoop_lookup_file_pos_pix": (None, [_p, _u, _u, _PU, _PU]),
    "jsnoop_lookup_blk_ycc": (None, [_p, _u, _u, _PI, _PI, _PI]),
    "jsnoop_pixel_to_mcu": (None, [_p, _u, _u, _PU, _PU]),
    "jsnoop_pixel_to_blk": (None, [_p, _u, _u, _PU, _PU]),
    "jsnoop_mcu_xy_to_linear": (_u, [_p, _u, _u]),
    "jsnoop_set_dump_histo_y": (None, [_p, _i]),
    "jsnoop_overlay_install": (_i, [_p, _p, _u, _u]),
    "jsnoop_overlay_remove_all": (None, [_p]),
    "jsnoop_overlay_get_num": (_u, [_p]),
    "jsnoop_overlay_get": (_i, [_p, _u, C.POINTER(_p), _PU, _PU]),
    "jsnoop_set_preview_mode": (None, [_p, _u]),
    "jsnoop_get_preview_mode": (_u, [_p]),
    "jsnoop_set_preview_ycc_offset": (None, [_p, _u, _u, _i, _i, _i]),
    "jsnoop_get_preview_ycc_offset": (None, [_p, _PU, _PU, _PI, _PI, _PI]),
    "jsnoop_set_preview_mcu_insert": (None, [_p, _u, _u, _i]),
    "jsnoop_get_preview_mcu_insert": (None, [_p, _PU, _PU, _PU]),
    "jsnoop_get_geometry": (None, [_p, _PU]),
    "jsnoop_mcu_file_map": (_p, [_p]),
    "jsnoop_blk_dc_ptrs": (None, [_p, C.POINTER(_p), C.POINTER(_p), C.POINTER(_p)]),
    "jsnoop_dht_histo": (_p, [_p]),
    "jsnoop_scan_status": (None, [_p, _PU]),
    "jsnoop_bright_avg": (None, [_p, _PI]),
    "jsnoop_get_color_stats": (None, [_p, _p]),
    "jsnoop_export_tiff": (_i, [_p, C.c_char_p, _i]),
    "jsnoop_idct_lut": (_p, [_p]),
    "jsnoop_dht_lookupfast": (_p, [_p]),
    "jsnoop_idct_block": (None, [_p, _p, _p]),
    "jsnoop_color_sweep": (C.c_int, [_p, _p]),
    "jsnoop_last_path": (_i, [_p]),
    "jsnoop_last_flags": (C.c_uint32, [_p]),
    "jsnoop_last_side_mode": (_i, [_p]),
    "jsnoop_batch_create": (_p, [_p]),
    "jsnoop_batch_destroy": (None, [_p]),
    "jsnoop_batch_clear": (None, [_p]),
    "jsnoop_batch_set_options": (None, [_p, _i, _i, _i]),
    "jsnoop_batch_add": (_i, [_p, _p, _p, _sz, _u]),
    "jsnoop_batch_add_jpeg": (_i, [_p, _p, _sz]),
    "jsnoop_batch_tile": (_i, [_p, _i]),
    "jsnoop_batch_set_split": (_i, [_p, _i]),
    "jsnoop_batch_split_parts": (_i, [_p]),
    "jsnoop_tuning_defaults": (None, [C.POINTER(Tuning)]),
    "jsnoop_tuning_defaults_sized": (None, [C.POINTER(Tuning), C.c_uint32]),
    "jsnoop_batch_set_tuning": (_i, [_p, C.POINTER(Tuning)]),
    "jsnoop_batch_get_tuning": (None, [_p, C.POINTER(Tuning)]),
    "jsnoop_set_tuning": (_i, [_p, C.POINTER(Tuning)]),
    "jsnoop_batch_count": (_i, [_p]),
    "jsnoop_batch_upload": (_i, [_p]),
    "jsnoop_batch_decode": (_i, [_p]),
    "jsnoop_batch_sync": (_i, [_p]),
    "jsnoop_batch_decode_timed": (C.c_double, [_p, _i, C.POINTER(C.c_double)]),
    "jsnoop_stage_name": (C.c_char_p, [_i]),
    "jsnoop_batch_image_info": (_i, [_p, _i, _PU]),
    "jsnoop_batch_dib_dev": (_p, [_p, _i]),
    "jsnoop_batch_read_dib": (_i, [_p, _i, _p]),
    "jsnoop_batch_read_planes": (_i, [_p, _i, _p, _p, _p]),
    "jsnoop_batch_read_coefs": (_i, [_p, _i, _p, _sz]),
    "jsnoop_batch_color_stats": (_i, [_p, _i, _i, _p]),
    "jsnoop_batch_dib_hashes": (_i, [_p, _p]),
    "jsnoop_batch_enable_log": (_i, [_p, _i]),
    "jsnoop_batch_side_outputs": (_i, [_p, _i, _p, _p, _p, _p, _p, _PU, _PI]),
    "jsnoop_batch_log": (_i, [_p, _i, _i, _i, _i, LOG_FN, _p]),
    "jsnoop_batch_export_tiff": (_i, [_p, _i, C.c_char_p, _i]),
    "jsnoop_batch_algorithmic_bytes": (C.c_uint64, [_p]),
    "jsnoop_batch_add_progressive": (_i, [_p, _p, _sz]),
    "jsnoop_pipeline_create": (_p, [_i]),
    "jsnoop_pipeline_destroy": (None, [_p]),
    "jsnoop_pipeline_slot": (_p, [_p, _i]),
    "jsnoop_pipeline_run": (_i, [_p, _i, _i, C.POINTER(C.c_double)]),
    "jsnoop_batch_pixels": (C.c_uint64, [_p]),
}

_lib = None


def load(require_device: bool = True) -> C.CDLL:
    """Loads the shared library and types every entry point.  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make -C jpegsnoop_amd/csrc` (or __graft_entry__.build()). "
                "jpegsnoop_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError here = header/library mismatch
            fn.restype, fn.argtypes = res, args
        _lib = lib
    if require_device and _lib.jsnoop_device_count() <= 0:
        raise RuntimeError("no HIP device visible: jpegsnoop_amd decodes on an AMD GPU only (no CPU fallback)")
    return _lib


def last_error() -> str:
    return (load(False).jsnoop_last_error() or b"").decode()
