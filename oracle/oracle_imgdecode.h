/* oracle_imgdecode.h -- TEST INFRASTRUCTURE ONLY: the parity checker.
 *
 * A CPU restatement, in plain C, of the reference's scan-decode hot path
 * (JPEGsnoop CimgDecode::DecodeScanImg and callees, reference
 * source/ImgDecode.cpp).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library -- the product path (libjsnoop_gpu.so)
 * never links, loads or calls it.
 *
 * PARITY PINNING: the reference has no tests, fixtures or golden vectors of its
 * own, so this restatement is pinned against the reference
 * ITSELF: oracle/_ref/libjsnoop_ref.so is the unmodified reference source
 * compiled in place (oracle/Makefile, target `ref`) and tests/test_oracle_vs_ref.py
 * compares DIB, int16 planes, MCU map, block-DC maps, Huffman histogram and
 * status words byte-for-byte over well-formed and corrupted streams; the
 * resulting hashes are committed under tests/golden/ so the check travels to
 * machines without /root/reference.
 *
 * The entry points mirror include/jsnoop_gpu.h (the product C ABI) one-for-one.
 */
#ifndef ORACLE_IMGDECODE_H
#define ORACLE_IMGDECODE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct OrcDecoder OrcDecoder;

OrcDecoder* orc_create(void);                       /* CimgDecode ctor   ImgDecode.cpp:142 */
void        orc_destroy(OrcDecoder*);               /* dtor              :239 */
void        orc_reset(OrcDecoder*);                 /* Reset()           :49  */
void        orc_reset_state(OrcDecoder*);           /* ResetState()      :286 */
/* the CSnoopConfig fields read at ImgDecode.cpp:2730-2741 */
void        orc_set_options(OrcDecoder*, int decode_ac, int histo_en, int stat_clip_en, unsigned err_max);

int      orc_set_dqt_entry(OrcDecoder*, unsigned tbl, unsigned nat, unsigned zz, unsigned val);      /* :424 */
int      orc_set_dqt_tables(OrcDecoder*, unsigned comp, unsigned tbl);                               /* :505 */
unsigned orc_get_dqt_entry(OrcDecoder*, unsigned tbl, unsigned nat);                                 /* :466 */
int      orc_set_dht_entry(OrcDecoder*, unsigned dest, unsigned cls, unsigned ind, unsigned len,
                           unsigned bits, unsigned mask, unsigned code);                             /* :748 */
int      orc_set_dht_size(OrcDecoder*, unsigned dest, unsigned cls, unsigned n);                     /* :834 */
int      orc_set_dht_tables(OrcDecoder*, unsigned comp, unsigned dc, unsigned ac);                   /* :536 */
void     orc_set_sof_samp_factors(OrcDecoder*, unsigned comp, unsigned h, unsigned v);               /* :619 */
void     orc_set_precision(OrcDecoder*, unsigned p);                                                 /* :564 */
void     orc_set_image_details(OrcDecoder*, unsigned x, unsigned y, unsigned nf, unsigned ns,
                               int rst_en, unsigned rst_interval);                                   /* :590 */

/* DecodeScanImg(nStart,bDisplay,bQuiet) :2723 -- bytes are read through the
 * equivalent of CwindowBuf::Buf (0 past end of file, WindowBuf.cpp:639-714). */
void     orc_decode_scan_img(OrcDecoder*, const uint8_t* file, size_t len, unsigned start, int display, int quiet);

int      orc_is_preview_ready(OrcDecoder*);                                                          /* :3753 */
void     orc_get_image_size(OrcDecoder*, unsigned* x, unsigned* y);                                  /* :4929 */
const uint8_t* orc_get_bitmap_ptr(OrcDecoder*);                                                      /* :4940 */
void     orc_get_pixmap_ptrs(OrcDecoder*, const int16_t** y, const int16_t** cb, const int16_t** cr);/* :4913 */
void     orc_lookup_file_pos_mcu(OrcDecoder*, unsigned mx, unsigned my, unsigned* byte, unsigned* bit); /* :5020 */
void     orc_lookup_blk_ycc(OrcDecoder*, unsigned bx, unsigned by, int* y, int* cb, int* cr);        /* :5037 */

/* internals exposed for parity of the side outputs (same layout as ref_driver) */
void     orc_get_geometry(OrcDecoder*, unsigned* out8);
const uint32_t* orc_mcu_file_map(OrcDecoder*);
void     orc_blk_dc_ptrs(OrcDecoder*, const int16_t** y, const int16_t** cb, const int16_t** cr);
const uint32_t* orc_dht_histo(OrcDecoder*);               /* [2][4][17] */
void     orc_scan_status(OrcDecoder*, unsigned* out8);
void     orc_bright_avg(OrcDecoder*, int* out10);
void     orc_set_preview_mode(OrcDecoder*, unsigned mode);                                          /* :633 */
unsigned orc_get_preview_mode(OrcDecoder*);
void     orc_set_preview_ycc_offset(OrcDecoder*, unsigned mcu_x, unsigned mcu_y, int y, int cb, int cr); /* :650 */
int      orc_export_tiff(OrcDecoder*, const char* path, int mode);   /* 0 RGB8, 1 RGB16, 2 YCC8; JPEGsnoopDoc.cpp:2110-2180 + FileTiff.cpp:436 */
void     orc_color_stats(OrcDecoder*, unsigned* out2482);   /* bHistoEn / bStatClipEn statistics, layout of jsnoop_get_color_stats */
const float* orc_idct_lut(OrcDecoder*);                   /* [64][64] */
const uint32_t* orc_dht_lookupfast(OrcDecoder*);          /* [2][4][1024] */
const int16_t* orc_coef_ptr(OrcDecoder*);                 /* dequantised natural-order coefficients, decode order, 64/block */
size_t   orc_coef_blocks(OrcDecoder*);
void     orc_idct_block(OrcDecoder*, const int16_t* coef64, float* out64);
void     orc_color_fast(const int* ycc, uint8_t* rgb, size_t n);
uint64_t orc_color_exhaustive_fnv(void);

#ifdef __cplusplus
}
#endif
#endif
