// jsnoop_types.h -- device-visible descriptors shared by the host code and the HIP kernels.
//
// HBM layout of one batch (all arenas are single hipMalloc blocks; offsets are per image):
//
//   raw    u8   file bytes of every image, 16-byte aligned starts     (read once by the entropy front end)
//   ustr   u8   un-stuffed entropy segment of every image (FF00->FF, RSTn removed) + interval table
//   coef   i16  [total_blocks][64] dequantised coefficients, NATURAL order, blocks in decode order
//               (mcu * blocks_per_mcu + block_in_mcu); slot 0 = dequantised DC difference
//   dccum  i16  [total_blocks] cumulative (predicted) DC per block = CimgDecode's m_nDcLum/Cb/Cr at that block
//   dib    u8   [img_y][img_x][4] bottom-up BGRA per image                (written once)
//   planes i16  3 x [blk_ymax*8][blk_xmax*8] per image (optional; m_pPixValY/Cb/Cr)
//   side   u32  per image: MCU file map, block-DC maps, Huffman histogram, status words
//
// Names follow the reference (MCU, block, restart interval, DIB), see SURVEY.md section 8.
#pragma once
#include <stdint.h>

#define JS_MAX_BLK_PER_MCU 48      // 3 components x (4 x 4) sampling factors (MAX_SAMP_FACT_H/V, ImgDecode.h:79-80)
#define JS_DHT_CODES       260     // MAX_DHT_CODES, ImgDecode.h:68
#define JS_FAST_BITS       9       // DHT_FAST_SIZE, ImgDecode.h:96
#define JS_CODE_UNUSED     0xFFFFFFFFu
#define JS_L1_BITS         9       // first-level index width of the parallel path's decode tables (1 KiB rows: four workgroups of the write pass fit a CU)
#define JS_LUT2_MAX        2048    // second-level entries (all tables together) in the parallel path's LUT form
#define JS_MAX_DEVICES     64      // devices one process may drive (per-device state of the launch wrappers)
#define JS_SUBSEQ_BYTES    128     // bytes of un-stuffed stream per sub-sequence (parallel entropy path)

// One distinct set of Huffman + quantisation tables, resolved per scan component
// (index t = (comp-1)*2 + class, class 0 = DC, 1 = AC).  De-duplicated across a batch.
struct JsTableSet {
    // --- exact-mirror form: the arrays CimgDecode::SetDhtEntry fills (ImgDecode.cpp:748-820)
    uint32_t fast[6][1 << JS_FAST_BITS];    // m_anDhtLookupfast: (len<<8)+code, 0xFFFFFFFF = miss
    uint32_t size[6];                       // m_anDhtLookupSize
    uint32_t bitlen[6][JS_DHT_CODES], bits[6][JS_DHT_CODES], mask[6][JS_DHT_CODES], code[6][JS_DHT_CODES];
    uint32_t dest_id[6];                    // DHT destination id behind each slot (histogram index)
    uint16_t qzz[3][64];                    // m_anDqtCoeffZz of the table selected for each component
    // --- parallel-path form: JS_L1_BITS-bit first level + second level for the longer codes (a few % of symbols).
    //     Identical tables share one row (Cb and Cr normally do): slot_row[slot] says which.
    //     entry: bit15 = 0 : [12:8] = code length (0 = invalid), [7:0] = symbol (run<<4 | size)
    //            bit15 = 1 : [14:12] = extra index bits nb (1..5), [11:0] = base into lut2
    uint16_t lut1[6][1 << JS_L1_BITS];
    uint16_t lut2[JS_LUT2_MAX];
    // --- state-only form for the synchronisation walks (k_sync needs positions and coefficient indices, not values): one entry
    //     covers the symbol at the window AND, for AC rows, the next one when its code also lies inside the JS_L1_BITS window.
    //     byte 0 = bits of symbol 1 (code + value), byte 1 = its advance of the coefficient index (DC rows: 1; EOB: 64, which ends
    //     the block from any index), byte 2 = bits of both symbols (0 = no second symbol described), byte 3 = index advance of both
    //     (a second symbol that is EOB adds 64).  Bit 31 = escape: the code is longer than the window, [14:12] extra index bits and
    //     [11:0] base of its second level (as in lut1), whose entries (lut2p, parallel to lut2) are single-symbol entries of the
    //     same form; bits 31 and 30: no code starts with these bits.
    uint32_t lutp[6][1 << JS_L1_BITS];
    uint32_t lut2p[JS_LUT2_MAX];
    // --- value form for the write pass (DC tables: single-symbol entries, [3:0] code length, [7:4] size): the symbol at the window and, when its
    //     code is visible in the same window, the AC symbol behind it -- [3:0] code length, [7:4] size, [11:8] run of symbol 1,
    //     [15:12] / [19:16] / [23:20] the same of symbol 2, [24] a second symbol is described.  Bit 31 = escape as in lutp.
    uint32_t lutw[6][1 << JS_L1_BITS];
    uint32_t row_sub[6];                    // index of a lut1 row among the rows of its class (DC tables / AC tables)
    uint32_t n_dc_rows, n_ac_rows;
    uint32_t slot_row[6];                   // row of lut1 holding the table of slot (comp-1)*2 + class
    uint32_t n_rows, lut2_used;
    uint32_t lut_ok;                        // 1 when every table fits the LUT form and is a canonical prefix code
};

struct JsImage {
    // frame / scan description (SetImageDetails :590, SetSofSampFactors :619, SetPrecision :564)
    uint32_t dim_x, dim_y, ncomp, precision, rst_en, rst_interval, decode_ac, want_planes;
    // geometry (DecodeScanImg :2831-2872)
    uint32_t mcu_w, mcu_h, mcu_xmax, mcu_ymax, blk_xmax, blk_ymax, img_x, img_y;
    uint32_t samp_h[4], samp_v[4], expand_h[4], expand_v[4];       // index 1..3 = Y, Cb, Cr
    uint32_t blk_per_mcu, total_blocks;
    uint8_t  blk_comp[JS_MAX_BLK_PER_MCU], blk_ch[JS_MAX_BLK_PER_MCU], blk_cv[JS_MAX_BLK_PER_MCU];
    // inputs
    uint64_t file_off;  uint32_t file_len, scan_start, scan_len;    // scan_len = bytes up to the terminating marker
    uint64_t ustr_off;  uint32_t ustr_cap;                          // un-stuffed stream capacity (= scan_len + slack)
    uint32_t tableset;
    // outputs
    uint64_t coef_off;      // in blocks
    uint64_t dib_off;       // in bytes
    uint64_t plane_off;     // in samples (3 planes of blk_xmax*8 x blk_ymax*8 follow each other)
    uint64_t side_off;      // in u32 words, see JS_SIDE_*
    uint64_t subseq_off;    // first sub-sequence slot of this image
    uint32_t n_subseq;      // capacity in sub-sequences
    uint64_t seg_off;       // first entry of this image's restart-interval table (u32 start bytes + end sentinel)
    uint32_t seg_cap;       // entries available
    uint64_t mcu_off;       // first byte of this image's per-MCU restart flags
    // preview controls (SetPreviewMode :633, SetPreviewYccOffset :650)
    uint32_t preview_mode; int32_t shift_y, shift_cb, shift_cr; uint32_t shift_mcu_x, shift_mcu_y;
    uint32_t err_max;
    // event log of the exact-mirror reader (what the reference writes to CDocLog while it decodes); 0 capacity = off
    uint64_t ev_off; uint32_t ev_cap;
    uint32_t rec_off;       // first word of this image's MCU-top positions in an arena that holds every image's (number of MCUs + 2 each; the batched side pass)
};

// Event records (6 u32 each, preceded by one count word per image): what the reference logs during the scan decode.
#define JS_ANOM_MAX 1024           // coefficient-index overflows the side pass records per image (beyond that: the mirror's side-only pass)
#define JS_EV_WORDS 6
#define JS_EV_MAX   1024
enum { JS_EV_OVERREAD_BEFORE = 1, JS_EV_OVERREAD_CODE, JS_EV_OVERREAD_BITS, JS_EV_CANT_FIND, JS_EV_RST_INDEX, JS_EV_MARKER,
       JS_EV_BAD_MARKER, JS_EV_BAD_HUFF, JS_EV_NUMCOEF, JS_EV_BAD_SCAN_MCU, JS_EV_RST_NOT_DETECTED };

// Per-image side block (u32 words, in this order):
//   [0..15]   status: 0 scan_bad, 1 scan_end, 2 #RST read, 3 num_pixels, 4 pos0, 5 align, 6 warn_bad, 7 first,
//             8 flags (JSNOOP_FLAG_*), 9 path, 10 un-stuffed length, 11 #intervals,
//             12-13 brightest-pixel key (64-bit: Y+32768 high, ~raster index low), 14 blocks decoded, 15 sum of final Y
//   [16..151] Huffman code-length histogram [2][4][17]
//   [152..]   MCU file map [mcu_ymax*mcu_xmax], then three block-DC maps (i16 packed as u16 pairs, each
//             padded to a whole word count), sized by the image
#define JS_SIDE_STATUS 0
#define JS_SIDE_HISTO  16
#define JS_SIDE_MCUMAP 152

// Layouts the back end converts without a replicated LDS tile (k_idct_color, mcu_to_dib_fast): three components, Y un-expanded,
// Cb and Cr one block each and both expanded eh x ev with eh, ev in {1, 2} (4:4:4, 4:2:2, 4:4:0, 4:2:0), default preview, no YCC shift.
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline bool js_fast_layout(const JsImage& im)
{
    const uint32_t eh = im.expand_h[2], ev = im.expand_v[2];
    return im.ncomp == 3 && im.preview_mode == 1 && (im.shift_y | im.shift_cb | im.shift_cr) == 0 &&
           im.expand_h[1] == 1 && im.expand_v[1] == 1 && im.samp_h[1] == eh && im.samp_v[1] == ev &&
           im.samp_h[2] == 1 && im.samp_v[2] == 1 && im.samp_h[3] == 1 && im.samp_v[3] == 1 && im.expand_h[3] == eh && im.expand_v[3] == ev &&
           eh >= 1 && eh <= 2 && ev >= 1 && ev <= 2;
}
// bytes of the back end's per-wave LDS tile for this image: Y plane + two bare chroma blocks (fast layouts), else three replicated planes
static inline uint32_t js_tile_bytes(const JsImage& im)
{
    const uint32_t b = js_fast_layout(im) ? im.mcu_h * (im.mcu_w + 8u) * 2u + 256u : 3u * im.mcu_h * (im.mcu_w + 8u) * 2u;
    return (b + 15u) & ~15u;
}

static inline __host__ __device__ uint32_t js_side_words(uint32_t nmcu, uint32_t nblk) { return JS_SIDE_MCUMAP + nmcu + 3 * ((nblk + 1) / 2); }
