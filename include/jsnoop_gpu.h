/* jsnoop_gpu.h -- C ABI of libjsnoop_gpu.so, the MI355X-native scan-decode stage.
 *
 * This is the drop-in boundary for the reference's scan decoder.  The reference
 * has no FFI; its boundary is the public C++ method set of `CimgDecode`
 * (reference source/ImgDecode.h:286-356,384-385,407-425) as driven by
 * `CjfifDecode` and re-exported by `CJPEGsnoopCore::I_*` (source/JPEGsnoopCore.h:79-117).
 * One `JsnoopDecoder` handle == one `CimgDecode` object; each entry point cites the
 * method it replaces.  All pointers are plain host or device addresses, no C++ or
 * torch types cross this line.  INTEGRATION.md shows the `CimgDecode`-shaped C++
 * wrapper (jpegsnoop_amd/csrc/ImgDecodeGpu.h) a reference maintainer would bind.
 *
 * Threading: like the reference (single-threaded, non-re-entrant objects,
 * source/JPEGsnoopCore.cpp:46) a handle may be used from one host thread at a
 * time; different handles are independent.  Work is issued on the handle's own HIP
 * stream (or the caller's, see jsnoop_batch_create).
 *
 * Errors: setters return 0 / 1 like the reference's bool setters; decode entry
 * points return void like DecodeScanImg and report through jsnoop_is_preview_ready,
 * the status words and the log callback (the reference's CDocLog sink).  If the HIP
 * runtime or device is unavailable, jsnoop_create returns NULL and
 * jsnoop_last_error() says why -- there is no CPU fallback in this library.
 */
#ifndef JSNOOP_GPU_H
#define JSNOOP_GPU_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define JSNOOP_ABI_VERSION 1

typedef struct JsnoopDecoder JsnoopDecoder;
typedef struct JsnoopBatch   JsnoopBatch;

/* ---- library / device ------------------------------------------------------ */
int         jsnoop_abi_version(void);
/* Host-only self test (no device needed): `rounds` random canonical Huffman table sets through the builders of the parallel
 * path's decode tables (two-level tables, state-only pair entries of the sync pass, value-pair entries of the write pass), every
 * first-level window checked against a plain search through the code list.  Returns the number of disagreements (0 = pass). */
int         jsnoop_selftest_tables(unsigned seed, unsigned rounds);
/* Host-only self test: the marker searches of the staging code (scan end, next FF: sixteen bytes per step) against byte-at-a-time loops on
 * `rounds` random buffers dense in FF / 00 / RSTn / other markers, every alignment.  Returns the number of disagreements (0 = pass). */
int         jsnoop_selftest_bytes(unsigned seed, unsigned rounds);
const char* jsnoop_last_error(void);                /* thread-local text of the last failure      */
int         jsnoop_device_count(void);              /* number of visible HIP devices (0 = none)   */
int         jsnoop_set_device(int device);          /* device used by objects created afterwards  */

/* log sink: replaces CDocLog::AddLine / AddLineWarn / AddLineErr (source/DocLog.cpp:102-194).
 * level: 0 = info, 1 = warning, 2 = error.  NULL disables logging (default). */
typedef void (*jsnoop_log_fn)(void* user, int level, const char* text);

/* ---- lifecycle: CimgDecode ctor :142, dtor :239, Reset :49, ResetState :286 ---- */
JsnoopDecoder* jsnoop_create(void);
void           jsnoop_destroy(JsnoopDecoder*);
void           jsnoop_reset(JsnoopDecoder*);
void           jsnoop_reset_state(JsnoopDecoder*);
/* the two halves of ResetState the class also has on their own: ResetDqtTables :343 (table selection, coefficients, number of SOF components)
 * and ResetDhtLookup :373 (code lists, fast look-up, table selection, the code-length histogram)                                          */
void           jsnoop_reset_dqt_tables(JsnoopDecoder*);
void           jsnoop_reset_dht_lookup(JsnoopDecoder*);
void           jsnoop_set_log_callback(JsnoopDecoder*, jsnoop_log_fn fn, void* user);

/* ---- options: the CSnoopConfig fields read at ImgDecode.cpp:2730-2741 ---------
 * decode_ac = bDecodeScanImgAc ("Full IDCT"), histo_en = bHistoEn,
 * stat_clip_en = bStatClipEn, err_max = nErrMaxDecodeScan.                      */
void jsnoop_set_options(JsnoopDecoder*, int decode_ac, int histo_en, int stat_clip_en, unsigned err_max);

/* ---- tables (SetDqtEntry :424, SetDqtTables :505, GetDqtEntry :466, SetDhtEntry :748,
 *      SetDhtSize :834, SetDhtTables :536) -- same argument meaning and range checks */
int      jsnoop_set_dqt_entry(JsnoopDecoder*, unsigned tbl_dest_id, unsigned coeff_ind, unsigned coeff_ind_zz, unsigned value);
int      jsnoop_set_dqt_tables(JsnoopDecoder*, unsigned comp_ind, unsigned tbl);
unsigned jsnoop_get_dqt_entry(JsnoopDecoder*, unsigned tbl_dest_id, unsigned coeff_ind);
int      jsnoop_set_dht_entry(JsnoopDecoder*, unsigned dest_id, unsigned cls, unsigned ind, unsigned len,
                              unsigned bits_left_just, unsigned mask_left_just, unsigned code);
int      jsnoop_set_dht_size(JsnoopDecoder*, unsigned dest_id, unsigned cls, unsigned size);
int      jsnoop_set_dht_tables(JsnoopDecoder*, unsigned comp_ind, unsigned tbl_dc, unsigned tbl_ac);

/* ---- geometry (SetSofSampFactors :619, SetPrecision :564, SetImageDetails :590) */
void jsnoop_set_sof_samp_factors(JsnoopDecoder*, unsigned comp_ind, unsigned samp_h, unsigned samp_v);
void jsnoop_set_precision(JsnoopDecoder*, unsigned precision);
void jsnoop_set_image_details(JsnoopDecoder*, unsigned dim_x, unsigned dim_y, unsigned comps_sof, unsigned comps_sos,
                              int rst_en, unsigned rst_interval);

/* SetImageDimensions :2706 (the PSD path's way of sizing the preview, source/JfifDecode.cpp:7374): the base rectangle of the preview, which
 * DecodeScanImg sets to the MCU-rounded image size itself (:2874).  jsnoop_get_image_dimensions reads it back.                            */
void jsnoop_set_image_dimensions(JsnoopDecoder*, unsigned width, unsigned height);
void jsnoop_get_image_dimensions(JsnoopDecoder*, unsigned* width, unsigned* height);
/* The public members CjfifDecode pokes when the preview does NOT come from the scan decoder (source/ImgDecode.h:508-510, used at
 * source/JfifDecode.cpp:7369-7373: a Photoshop file's image decoded into m_pDibTemp): jsnoop_dib_temp_create is m_pDibTemp.Kill() +
 * CreateDIB(width, height, 32) + GetDIBBitArray() -- a zeroed bottom-up BGRA buffer owned by the decoder that the caller fills and
 * jsnoop_get_bitmap_ptr then hands out (GetBitmapPtr :4940 returns m_pDibTemp's bits whatever filled them) --, the two setters are the
 * members m_bDibTempReady and m_bPreviewIsJpeg (IsPreviewReady :3753 returns the latter).  Reset() drops the buffer (:80-83).             */
uint8_t* jsnoop_dib_temp_create(JsnoopDecoder*, unsigned width, unsigned height);
void     jsnoop_set_dib_temp_ready(JsnoopDecoder*, int ready);
int      jsnoop_get_dib_temp_ready(JsnoopDecoder*);
void     jsnoop_set_preview_is_jpeg(JsnoopDecoder*, int is_jpeg);

/* ---- minimal JFIF front end: walks SOI/DQT/SOF0-1/DHT/DRI up to the first SOS and
 *      issues the setter calls above exactly as CjfifDecode::DecodeMarker does (source/JfifDecode.cpp:3581,
 *      :3600, :4648, :5008-5025, :5161, :5291).  On success *scan_start is the nStart to pass to
 *      jsnoop_decode_scan_img.  Returns 0, or -1 with jsnoop_last_error() (e.g. SOF2: the reference
 *      refuses progressive files, :4827-4833).                                                    */
int  jsnoop_jfif_walk(JsnoopDecoder*, const uint8_t* file, size_t len, unsigned* scan_start);

/* ---- decode: DecodeScanImg(nStart,bDisplay,bQuiet) :2723 ------------------------
 * `file`/`len` is the whole file image that the reference reads through
 * CwindowBuf::Buf (source/WindowBuf.cpp:639; bytes past `len` read as 0).  The bytes are
 * staged through pinned host memory with hipMemcpyAsync, with the overlays installed through
 * jsnoop_overlay_install applied to the staged copy.  Blocks until the DIB is resident in HBM. */
void jsnoop_decode_scan_img(JsnoopDecoder*, const uint8_t* file, size_t len, unsigned start, int display, int quiet);

/* ---- results: IsPreviewReady :3753, GetImageSize :4929, GetBitmapPtr :4940,
 *      GetPixMapPtrs :4913, LookupFilePosMcu :5020, LookupFilePosPix :5001, LookupBlkYCC :5037.
 * Host pointers are owned by the decoder and stay valid until the next
 * Reset / DecodeScanImg / destroy (same ownership rule as the reference); they are
 * filled by a D2H copy on first request.  The *_dev variants return the HBM copies. */
int            jsnoop_is_preview_ready(JsnoopDecoder*);
void           jsnoop_get_image_size(JsnoopDecoder*, unsigned* x, unsigned* y);
const uint8_t* jsnoop_get_bitmap_ptr(JsnoopDecoder*);
const void*    jsnoop_get_bitmap_dev(JsnoopDecoder*);
void           jsnoop_get_pixmap_ptrs(JsnoopDecoder*, const int16_t** y, const int16_t** cb, const int16_t** cr);
void           jsnoop_lookup_file_pos_mcu(JsnoopDecoder*, unsigned mcu_x, unsigned mcu_y, unsigned* byte, unsigned* bit);
void           jsnoop_lookup_file_pos_pix(JsnoopDecoder*, unsigned pix_x, unsigned pix_y, unsigned* byte, unsigned* bit);
void           jsnoop_lookup_blk_ycc(JsnoopDecoder*, unsigned blk_x, unsigned blk_y, int* y, int* cb, int* cr);

/* PixelToMcu :5056, PixelToBlk :5071, McuXyToLinear :5088 (the hover / status-bar helpers of source/JPEGsnoopViewImg.cpp:294-334) */
void     jsnoop_pixel_to_mcu(JsnoopDecoder*, unsigned pix_x, unsigned pix_y, unsigned* mcu_x, unsigned* mcu_y);
void     jsnoop_pixel_to_blk(JsnoopDecoder*, unsigned pix_x, unsigned pix_y, unsigned* blk_x, unsigned* blk_y);
unsigned jsnoop_mcu_xy_to_linear(JsnoopDecoder*, unsigned mcu_x, unsigned mcu_y);

/* bDumpHistoY (CSnoopConfig, read at ImgDecode.cpp:2730): with bHistoEn and bDisplay the decode log ends with
 * ReportHistogramY's 256 lines of the 2048-bin Y histogram (:3740-3741, :3845-3868).                          */
void jsnoop_set_dump_histo_y(JsnoopDecoder*, int on);

/* ---- byte overlays: CwindowBuf::OverlayInstall / OverlayRemoveAll / OverlayGet / OverlayGetNum
 *      (source/WindowBuf.cpp:516-620).  The reference's fault-injection tool: Buf() (:639-660) returns an overlay's
 *      byte instead of the file's wherever an enabled overlay covers the offset, the LAST installed one winning.
 *      jsnoop_decode_scan_img applies the installed overlays to its staged copy of the file image (offsets inside the
 *      file), so a re-decode after an install shows the patched stream exactly as the reference's would.
 *      At most 500 overlays of fewer than 500 bytes (NUM_OVERLAYS / MAX_OVERLAY, WindowBuf.h:42-43).           */
int      jsnoop_overlay_install(JsnoopDecoder*, const uint8_t* data, unsigned len, unsigned begin);   /* 1 = installed, 0 = refused */
void     jsnoop_overlay_remove_all(JsnoopDecoder*);
unsigned jsnoop_overlay_get_num(JsnoopDecoder*);
int      jsnoop_overlay_get(JsnoopDecoder*, unsigned ind, const uint8_t** data, unsigned* len, unsigned* begin);

/* ---- preview re-render on the retained planes: SetPreviewMode :633,
 *      SetPreviewYccOffset :650 (each re-runs the colour kernel only)            */
void     jsnoop_set_preview_mode(JsnoopDecoder*, unsigned mode);
unsigned jsnoop_get_preview_mode(JsnoopDecoder*);
void     jsnoop_set_preview_ycc_offset(JsnoopDecoder*, unsigned mcu_x, unsigned mcu_y, int y, int cb, int cr);
void     jsnoop_get_preview_ycc_offset(JsnoopDecoder*, unsigned* mcu_x, unsigned* mcu_y, int* y, int* cb, int* cr);   /* :670 */
/* SetPreviewMcuInsert :682 / GetPreviewMcuInsert :693 ("UNUSED" in the reference: stored, triggers a re-render, no pixel effect) */
void     jsnoop_set_preview_mcu_insert(JsnoopDecoder*, unsigned mcu_x, unsigned mcu_y, int len);
void     jsnoop_get_preview_mcu_insert(JsnoopDecoder*, unsigned* mcu_x, unsigned* mcu_y, unsigned* len);

/* ---- Progressive (SOF2) files -- beyond the reference, which refuses them (source/JfifDecode.cpp:4827-4833; the two
 *      entry points above refuse them the same way).  Walks ALL scans of the file (spectral selection and successive
 *      approximation, T.81 Annex G; tables may change between scans), decodes every restart interval of every scan as an
 *      independent lane, and hands the coefficient arena to the same back end as the baseline path, so a progressive
 *      file carrying the coefficients of a baseline file yields that file's DIB.  Results through the getters above
 *      (jsnoop_last_path() == 3; no MCU file map / block-DC maps).  Returns the number of scans, or -1.            */
int jsnoop_decode_progressive(JsnoopDecoder*, const uint8_t* file, size_t len);

/* ---- Export to TIFF (CJPEGsnoopDoc::OnToolsExporttiff, source/JPEGsnoopDoc.cpp:2008-2190, FileTiff::WriteFile
 *      source/FileTiff.cpp:436): mode 0 = RGB 8 bit, 1 = RGB 16 bit, 2 = YCC 8 bit (three-component images); byte-identical
 *      to the reference's file.  The pixel strip is arranged on the device.  0 on success, -1 + jsnoop_last_error().  */
int jsnoop_export_tiff(JsnoopDecoder*, const char* path, int mode);

/* ---- decoder internals that the reference keeps in public/inspectable members and
 *      that the log / hover UI consume (side outputs).
 *      Layouts match oracle/ref_shim/ref_driver.cpp so the same parity script runs
 *      against reference, oracle and this library.                                  */
void            jsnoop_get_geometry(JsnoopDecoder*, unsigned* out8);   /* McuW,McuH,McuXMax,McuYMax,BlkXMax,BlkYMax,ImgSizeX,ImgSizeY */
const uint32_t* jsnoop_mcu_file_map(JsnoopDecoder*);                   /* m_pMcuFileMap [McuYMax][McuXMax], PackFileOffset :5104 */
void            jsnoop_blk_dc_ptrs(JsnoopDecoder*, const int16_t** y, const int16_t** cb, const int16_t** cr);
const uint32_t* jsnoop_dht_histo(JsnoopDecoder*);                      /* m_anDhtHisto [2][4][17] */
void            jsnoop_scan_status(JsnoopDecoder*, unsigned* out8);    /* scan_bad, scan_end, #RST, num_pixels, pos0, align, warn_bad, first */
void            jsnoop_bright_avg(JsnoopDecoder*, int* out10);         /* brightest pixel + average Y (:4722-4730, :4802-4819) */
/* Statistics of the bHistoEn / bStatClipEn colour path (ConvertYCCtoRGB :4229, CapYccRange :4341, CapRgbRange :4495;
 * enable with jsnoop_set_options before DecodeScanImg).  As in the reference they are cleared by DecodeScanImg
 * (:3145-3155) and accumulated by every CalcChannelPreview, i.e. also by the preview re-renders below.  Layout,
 * JSNOOP_STATS_WORDS 32-bit words: [0..35] PixelCcHisto (ImgDecode.h:238-279: min,max,sum as int for PreclipY/Cb/Cr,
 * ClipY/Cb/Cr, ClipR/G/B, PreclipR/G/B), [36] nCount, [37..49] PixelCcClip (ImgDecode.h:220-234), [50..433]
 * m_anCcHisto_r/g/b[128], [434..2481] m_anHistoYFull[2048].                                                      */
#define JSNOOP_STATS_WORDS 2482
void            jsnoop_get_color_stats(JsnoopDecoder*, uint32_t* out);
const float*    jsnoop_idct_lut(JsnoopDecoder*);                       /* m_afIdctLookup [64][64] as uploaded to the device */
const uint32_t* jsnoop_dht_lookupfast(JsnoopDecoder*);                 /* m_anDhtLookupfast [2][4][1024] */
void            jsnoop_idct_block(JsnoopDecoder*, const int16_t* coef64, float* out64);   /* one block through the device IDCT */
/* every (Y, Cb, Cr) in [-128,127]^3 through the device ConvertYCCtoRGBFastFloat (ImgDecode.cpp:4086):
 * out_bgra[(Y+128)<<16 | (Cb+128)<<8 | (Cr+128)] = B | G<<8 | R<<16; 2^24 words; 0 on success */
int             jsnoop_color_sweep(JsnoopDecoder*, uint32_t* out_bgra);
/* which device path decoded the last image: 1 = parallel (self-synchronising) entropy
 * decode, 2 = sequential exact-mirror entropy kernel (taken for streams the parallel
 * path flags as malformed), 0 = nothing decoded */
int             jsnoop_last_path(JsnoopDecoder*);
uint32_t        jsnoop_last_flags(JsnoopDecoder*);                     /* JSNOOP_FLAG_* raised by the parallel path */
/* who produced the side outputs (MCU file map, histogram, status words, messages: what GetPixMapPtrs / LookupFilePosMcu and the log of
 * DecodeScanImg, ImgDecode.cpp:2723, show) of the last image: 0 = not asked for yet, 1 = the parallel side pass, 2 = the sequential
 * exact-mirror reader, 3 = the parallel side pass plus exact readers on chunks of a few MCUs (flagged files, in milliseconds) */
int             jsnoop_last_side_mode(JsnoopDecoder*);

#define JSNOOP_FLAG_BAD_CODE      0x0001u  /* no Huffman code matches (alone: decoded the reference's way by the parallel path) */
#define JSNOOP_FLAG_OVERRUN       0x0002u  /* code or extra bits run past the interval / scan end       */
#define JSNOOP_FLAG_COEF_OVERFLOW 0x0004u  /* coefficient index > 63                                    */
#define JSNOOP_FLAG_RST_MISALIGN  0x0008u  /* RSTn not on an MCU boundary (alone: followed the reference's way) */
#define JSNOOP_FLAG_SHORT         0x0010u  /* entropy data ends before all MCUs are decoded             */
#define JSNOOP_FLAG_TABLES        0x0020u  /* tables not expressible in the parallel path's LUT form    */
#define JSNOOP_FLAG_MARKER        0x0040u  /* non-RST marker or FFFF inside the scan                    */
#define JSNOOP_FLAG_NOSYNC        0x0080u  /* sub-sequence chain failed to converge                     */
#define JSNOOP_FLAG_BAD_EDGE      0x0100u  /* an anomaly whose outcome depends on the reader's look-ahead: mirror */
#define JSNOOP_FLAG_FORCED        0x8000u  /* caller forced the exact path                              */

/* ---- batch submit: N files -> N DIBs, all resident in HBM --------------------------
 * The batched analogue of CJPEGsnoopCore::DoBatchFileProcess (source/JPEGsnoopCore.cpp:765),
 * whose per-file semantics are preserved: every image is decoded exactly as a fresh
 * CimgDecode would.  `stream` is a hipStream_t (NULL = the batch creates its own).  */
JsnoopBatch* jsnoop_batch_create(void* stream);
void         jsnoop_batch_destroy(JsnoopBatch*);
void         jsnoop_batch_clear(JsnoopBatch*);
void         jsnoop_batch_set_options(JsnoopBatch*, int decode_ac, int want_planes, int force_exact_path);
/* Adds one image using the table/geometry state currently held by `tables`
 * (i.e. after the SetDqt/SetDht/SetSof/SetImageDetails calls of its header).
 * The file bytes are copied into the batch's pinned staging arena.
 * Returns the image index, or -1 with jsnoop_last_error() set.                      */
int          jsnoop_batch_add(JsnoopBatch*, const JsnoopDecoder* tables, const uint8_t* file, size_t len, unsigned scan_start);
/* Adds one JPEG file image, walking its header with the built-in minimal JFIF front
 * end (the subset of CjfifDecode::DecodeMarker that feeds CimgDecode).               */
int          jsnoop_batch_add_jpeg(JsnoopBatch*, const uint8_t* file, size_t len);
/* A progressive (SOF2) file into a batch -- jsnoop_batch_add_jpeg routes such files here by itself.  A batch holds either
 * baseline or progressive files; all scans of all its images then decode together, one launch per dependency level
 * (every restart interval of every scan of every image is one lane), then one finalize pass and the common back end.      */
int          jsnoop_batch_add_progressive(JsnoopBatch*, const uint8_t* file, size_t len);
/* Tiles already-added images so the batch holds `total` images (image i reuses the
 * bytes of image i % n): the bench's "N distinct seeds tiled to 1024".               */
int          jsnoop_batch_tile(JsnoopBatch*, int total);
/* Beyond the reference: the decodes of a batch may run its two halves on two streams side by side (same arenas, same results; the
   kernels of one half fill the thinly populated phases of the other: about 4 % more throughput on 1024 x 1080p).  parts = 0: the
   library decides (two streams from 8 MB of scan data in the batch -- the default), 1: one stream (profiling: per-kernel timings
   are then those of whole-batch launches), 2: two streams whenever the batch has two images.  0 / -1.                            */
int          jsnoop_batch_set_split(JsnoopBatch*, int parts);
int          jsnoop_batch_split_parts(const JsnoopBatch*);              /* what the setting comes to for the images the batch holds now: 1 or 2 */
/* ---- tuning: how the library decodes, never what it produces.  Every field has an automatic setting (0); the environment variables
 *      of tools/README.md are read ONCE per process and only supply the defaults jsnoop_tuning_defaults returns -- nothing on the
 *      decode path reads the environment.  A batch takes a copy at jsnoop_batch_set_tuning (call before upload; -1 + last_error on a
 *      value out of range); jsnoop_set_tuning does the same for the private batch behind a single-image decoder.                      */
typedef struct JsnoopTuning {
    uint32_t struct_size;     /* sizeof(JsnoopTuning) of the caller: a shorter (older) struct is accepted, the fields it lacks are automatic, and
                                 jsnoop_batch_get_tuning writes no byte past it; a longer one than the library knows is refused              */
    int32_t  sub_wl;          /* log2(32-bit words) of a sub-sequence: 4..8 = 64 B .. 1 KiB; 0 = by job size (4 / 5 / 6 / 7)              */
    int32_t  cand_rounds;     /* synchronisation form: -1 = rounds of k_sync only, n > 0 = candidates with at most n walk rounds (<= 64),
                                 0 = automatic (candidates with 16 rounds while the job is small enough, see cand_max_walks)          */
    uint64_t cand_max_walks;  /* largest job (64-byte pieces x blocks per MCU) that synchronises by candidates; 0 = 2 600 000         */
    int32_t  sync_launches;   /* synchronisation by rounds: 0 = one cut launch of k_sync, then list rounds over the whole job
                                 (k_sync_links / k_sync_round); n > 0 = n plain launches of k_sync (the form before round 6)           */
    int32_t  write_lanes;     /* lanes per sub-sequence in the write pass of the smallest jobs: 1, 2; 0 = automatic (2 up to 40 960 pieces) */
    int32_t  split;           /* as jsnoop_batch_set_split: 0 automatic, 1 one stream, 2 two streams                                  */
    int32_t  mcus_per_wave;   /* MCUs per back-end wave; 0 = one round of workgroups over the chip, at most 64                        */
    int32_t  pg_lanes;        /* progressive: restart intervals per wave 1, 2, 4, 8, 16 or 64 (lane-per-interval kernel); 0 = by batch size */
    uint32_t cross_checks;    /* JSNOOP_XC_* bits: alternative code paths kept for cross-checking, same results                       */
    uint32_t debug;           /* JSNOOP_DBG_* bits: diagnostics on stderr                                                             */
} JsnoopTuning;
#define JSNOOP_XC_BACKEND_GENERIC 0x01u  /* the all-layouts back-end kernel for every launch                                          */
#define JSNOOP_XC_WRITE_V1        0x02u  /* the first form of the write pass                                                          */
#define JSNOOP_XC_NO_TAIL         0x04u  /* damaged files: no tail take-over / second attempt, the whole image through the mirror     */
#define JSNOOP_XC_SIDE_EXACT      0x08u  /* side outputs always from the exact-mirror reader                                          */
#define JSNOOP_XC_CAND_VERIFY     0x10u  /* k_sync's verification mode behind every candidate chain                                   */
#define JSNOOP_XC_UNSTUFF_3PASS   0x20u  /* un-stuffing as count / scan / write passes instead of the fused look-back pass            */
#define JSNOOP_DBG_CAND           0x01u  /* candidate chain: rounds, queued walks                                                     */
#define JSNOOP_DBG_CAND_LINKS     0x02u  /* ... and the links left open per image (stops the stream)                                  */
#define JSNOOP_DBG_TAIL           0x04u  /* damaged files: tail take-over decisions                                                   */
#define JSNOOP_DBG_TIMING         0x08u  /* single-image calls: where the wall time goes                                              */
void         jsnoop_tuning_defaults(JsnoopTuning* out);                 /* struct_size set, everything else the process defaults: writes
                                                                          * sizeof(JsnoopTuning) of THIS header                       */
/* ... for a caller built against an older (shorter) JsnoopTuning: at most struct_size bytes are written, struct_size says how many.
 * jsnoop_batch_get_tuning reads out->struct_size the same way: set it to sizeof(your JsnoopTuning) (or 0: this header's) before the call. */
void         jsnoop_tuning_defaults_sized(JsnoopTuning* out, uint32_t struct_size);
int          jsnoop_batch_set_tuning(JsnoopBatch*, const JsnoopTuning*);
void         jsnoop_batch_get_tuning(const JsnoopBatch*, JsnoopTuning* out);
int          jsnoop_set_tuning(JsnoopDecoder*, const JsnoopTuning*);
int          jsnoop_batch_count(const JsnoopBatch*);
int          jsnoop_batch_upload(JsnoopBatch*);      /* pinned host -> HBM (async), builds device descriptors */
int          jsnoop_batch_decode(JsnoopBatch*);      /* HBM -> HBM, asynchronous on the batch stream          */
int          jsnoop_batch_sync(JsnoopBatch*);        /* waits; then re-decodes flagged images on the exact path */
/* Timed decode: `reps` decodes bracketed by hipEvents on the batch stream; per-stage
 * average milliseconds into stage_ms[JSNOOP_NUM_STAGES] (may be NULL).  Returns the
 * average milliseconds per whole decode, < 0 on error.                              */
#define JSNOOP_NUM_STAGES 8
double       jsnoop_batch_decode_timed(JsnoopBatch*, int reps, double* stage_ms);
const char*  jsnoop_stage_name(int stage);
/* per-image results */
int          jsnoop_batch_image_info(const JsnoopBatch*, int i, unsigned* out16);  /* dim_x,dim_y,img_x,img_y,mcu_w,mcu_h,mcu_xmax,mcu_ymax,
                                                                                     blk_xmax,blk_ymax,scan_bytes,flags,path,ns,file_len,0 */
const void*  jsnoop_batch_dib_dev(const JsnoopBatch*, int i);                      /* bottom-up BGRA, img_x*img_y*4 bytes in HBM */
int          jsnoop_batch_read_dib(JsnoopBatch*, int i, uint8_t* host_dst);        /* D2H copy of one DIB */
int          jsnoop_batch_read_planes(JsnoopBatch*, int i, int16_t* y, int16_t* cb, int16_t* cr);
int          jsnoop_batch_read_coefs(JsnoopBatch*, int i, int16_t* dst, size_t max_blocks); /* dequantised blocks, decode order */
/* 64-bit FNV-1a of every DIB computed on the device (one word per image), for
 * whole-batch parity checks without moving 8 GB over PCIe.                          */
/* the same statistics for image i of a decoded batch (needs want_planes): one fresh pass, as a DecodeScanImg with
 * bHistoEn (histo_en != 0) or only bStatClipEn (histo_en == 0) would leave them */
int          jsnoop_batch_color_stats(JsnoopBatch*, int i, int histo_en, uint32_t* out);
int          jsnoop_batch_dib_hashes(JsnoopBatch*, uint64_t* host_dst);
/* ---- everything else DecodeScanImg leaves behind, per image of a decoded batch: what the per-file pass of the reference's batch loop
 *      produces (CJPEGsnoopCore::DoBatchFileProcess -> AnalyzeFile -> DoLogSave, source/JPEGsnoopCore.cpp:765-845; the log body of this
 *      path is source/ImgDecode.cpp:3021-3025, :3126-3135 and :3630-3745).  Same code as the single-image API, addressed at image i.
 *  jsnoop_batch_side_outputs: any pointer may be NULL.  mcu_map [mcu_ymax*mcu_xmax] (m_pMcuFileMap :3229), dc_* [blk_ymax*blk_xmax]
 *      (m_pBlkDcValY/Cb/Cr :3524-3608; Cb / Cr untouched for a one-component image), dht_histo [2][4][17] (m_anDhtHisto :1190), status8 as
 *      jsnoop_scan_status, bright_avg10 as jsnoop_bright_avg (three-component images: needs want_planes).  Produced on request by the
 *      parallel side pass (about 1 ms per image), without touching coefficients or pixels.
 *  jsnoop_batch_enable_log: keep the decoder's event records of every image (24 KiB of HBM each); call before upload.
 *  jsnoop_batch_log: the text DecodeScanImg(nStart, bDisplay = TRUE, bQuiet = quiet) of a fresh CimgDecode writes to CDocLog for image i
 *      under bHistoEn = histo_en / bStatClipEn = stat_clip_en, through `fn` (needs jsnoop_batch_enable_log, and want_planes for
 *      three-component images and the statistics).
 *  jsnoop_batch_export_tiff: jsnoop_export_tiff for image i.   All four: 0 on success, -1 + jsnoop_last_error().                    */
int          jsnoop_batch_enable_log(JsnoopBatch*, int on);
int          jsnoop_batch_side_outputs(JsnoopBatch*, int i, uint32_t* mcu_map, int16_t* dc_y, int16_t* dc_cb, int16_t* dc_cr, uint32_t* dht_histo,
                                       unsigned* status8, int* bright_avg10);
int          jsnoop_batch_log(JsnoopBatch*, int i, int histo_en, int stat_clip_en, int quiet, jsnoop_log_fn fn, void* user);
int          jsnoop_batch_export_tiff(JsnoopBatch*, int i, const char* path, int mode);
uint64_t     jsnoop_batch_algorithmic_bytes(const JsnoopBatch*);                   /* sum(scan bytes + DIB bytes), SURVEY.md 8(d) */
uint64_t     jsnoop_batch_pixels(const JsnoopBatch*);                              /* sum(SOF X*Y) */

/* ---- staging pipeline: the CwindowBuf replacement at batch scale (source/WindowBuf.cpp:351-416 BufLoadWindow, :639-714 Buf) ----
 * `slots` batch slots, each with its own pinned staging area, HBM arenas and stream (fill them through jsnoop_pipeline_slot and
 * the jsnoop_batch_add* calls).  jsnoop_pipeline_run cycles `batches` batches through the slots: while one slot decodes, the next
 * slot's compressed bytes cross PCIe, and with d2h != 0 the previous slot's DIBs are copied back to pinned host memory meanwhile.
 * out_ms6: [0] wall ms per batch in steady state (timing scope T2 with d2h = 0, T3 with d2h = 1), then the pieces on their own:
 * [1] H2D ms, [2] decode ms (T1), [3] D2H ms (0 without d2h), [4] compressed bytes per batch, [5] DIB bytes per batch.       */
typedef struct JsnoopPipeline JsnoopPipeline;
JsnoopPipeline* jsnoop_pipeline_create(int slots);
void            jsnoop_pipeline_destroy(JsnoopPipeline*);
JsnoopBatch*    jsnoop_pipeline_slot(JsnoopPipeline*, int i);
int             jsnoop_pipeline_run(JsnoopPipeline*, int batches, int d2h, double* out_ms6);

#ifdef __cplusplus
}
#endif
#endif
