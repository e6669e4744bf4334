/* oracle_imgdecode.c -- TEST INFRASTRUCTURE ONLY: the parity checker.
 *
 * Plain-C restatement of the arithmetic and control flow of the reference's
 * scan decoder (JPEGsnoop 1.8.0, source/ImgDecode.cpp; line numbers below refer
 * to that file unless another file is named).  It is deliberately literal about
 * the reference's quirks (marker-driven restart, FF FF handling, one-bit resync
 * on a bad code, dense fp32 IDCT in natural order without FMA, fp32 colour
 * conversion with a true division) because bit-exact DIB parity is the contract.
 * Compile with -ffp-contract=off (oracle/Makefile).  See oracle_imgdecode.h for
 * how it is pinned against the compiled reference.
 */
#include "oracle_imgdecode.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define DHT_CLASSES   2
#define DHT_DESTS     4
#define DHT_CODES     260          /* MAX_DHT_CODES  ImgDecode.h:68 */
#define DHT_FAST_BITS 9            /* DHT_FAST_SIZE  ImgDecode.h:96 */
#define CODE_UNUSED   0xFFFFFFFFu  /* DHT_CODE_UNUSED ImgDecode.h:70 */
#define MAX_SAMP      4

enum { RSV_OK, RSV_EOB, RSV_UNDERFLOW, RSV_RST_TERM };          /* ImgDecode.h:166-171 */
enum { SB_OK, SB_BADMARK, SB_RST };                             /* ImgDecode.h:174-178 */
enum { PV_NONE, PV_RGB, PV_YCC, PV_R, PV_G, PV_B, PV_Y, PV_CB, PV_CR };   /* snoop.h tePreviewMode */

static const uint8_t kZigZag[64] = {       /* ITU-T T.81 Figure A.6; General.cpp:257-267 */
     0, 1, 8,16, 9, 2, 3,10, 17,24,32,25,18,11, 4, 5,
    12,19,26,33,40,48,41,34, 27,20,13, 6, 7,14,21,28,
    35,42,49,56,57,50,43,36, 29,22,15,23,30,37,44,51,
    58,59,52,45,38,31,39,46, 53,60,61,54,47,55,62,63 };

struct OrcDecoder {
    /* options (CSnoopConfig fields read at :2730-2741) */
    int      opt_decode_ac, opt_histo_en, opt_stat_clip_en; unsigned opt_err_max;
    /* tables */
    uint16_t dqt_nat[4][64], dqt_zz[4][64]; int dqt_sel[256];
    int      dht_sel[DHT_CLASSES][5];
    unsigned huff_mask[32];
    unsigned dht_setmax[DHT_CLASSES], dht_size[DHT_CLASSES][DHT_DESTS];
    unsigned dht_bitlen[DHT_CLASSES][DHT_DESTS][DHT_CODES], dht_bits[DHT_CLASSES][DHT_DESTS][DHT_CODES];
    unsigned dht_mask[DHT_CLASSES][DHT_DESTS][DHT_CODES], dht_code[DHT_CLASSES][DHT_DESTS][DHT_CODES];
    uint32_t dht_fast[DHT_CLASSES][DHT_DESTS][2 << DHT_FAST_BITS];
    uint32_t dht_histo[DHT_CLASSES][DHT_DESTS][17];
    /* frame / scan description */
    int      details_set; unsigned dim_x, dim_y, num_sof, num_sos, precision;
    unsigned samp_h[256], samp_v[256], samp_hmax, samp_vmax, samp_hmin, samp_vmin;
    unsigned per_mcu_h[256], per_mcu_v[256], expand_h[256], expand_v[256];
    int      rst_en; unsigned rst_interval;
    /* geometry */
    unsigned mcu_w, mcu_h, mcu_xmax, mcu_ymax, blk_xmax, blk_ymax, img_x, img_y;
    /* outputs */
    uint32_t* mcu_map; int16_t* blk_dc[3]; int16_t* pix[3]; uint8_t* dib; int dib_ready, preview_is_jpeg;
    /* preview controls (:633-690) */
    unsigned preview_mode; int shift_y, shift_cb, shift_cr; unsigned shift_mcu_x, shift_mcu_y;
    /* DC predictors (:496-501) */
    int16_t  dc_y, dc_cb, dc_cr, dc_y_css[16], dc_cb_css[16], dc_cr_css[16];
    /* bit reader (:618-636) */
    const uint8_t* file; size_t flen;
    unsigned buff, vacant; unsigned long ptr, ptr_first;
    unsigned pos[4], err[4], latch_err, num, align;
    int      scan_end, scan_bad, cur_err, restart_read;
    unsigned restart_count, restart_last, restart_expect, mcus_left, warn_bad;
    unsigned used1, used2, num_pixels;
    int      decode_ac;                         /* m_bDecodeScanAc */
    /* per-block scratch (:596-601) */
    float    lut[64][64]; int16_t dct[64]; float idct[64];
    /* stats (:579-590) */
    int      bright_valid, bright_y, bright_cb, bright_cr; unsigned bright_r, bright_g, bright_b;
    int      bright_mx, bright_my; long avg_y; int avg_valid;
    /* bHistoEn / bStatClipEn colour path (m_sHisto, m_sStatClip, m_anCcHisto_*, m_anHistoYFull, m_nWarnYccClipNum) */
    int      hist_en, stat_clip_en; unsigned warn_ycc_clip;
    int32_t  cc_histo[36]; unsigned cc_count, cc_clip[13], cc_rgb[3][128], cc_yfull[2048];
    /* test instrumentation: every decoded coefficient block, in decode order */
    int16_t* coef; size_t coef_blocks, coef_cap;
};

/* --------------------------------------------------------------- constants */
static void precalc_idct(OrcDecoder* d)                                   /* PrecalcIdct :2313-2351 */
{
    const float pi = (float)3.141592654, rh = (float)0.707106781;
    for (unsigned y = 0; y < 8; y++) for (unsigned x = 0; x < 8; x++)
        for (unsigned v = 0; v < 8; v++) for (unsigned u = 0; u < 8; u++) {
            float cu = (u == 0) ? rh : 1, cv = (v == 0) ? rh : 1;
            /* float-argument cos() resolves to the float overload in the C++ reference */
            float cp = cosf((2 * x + 1) * u * pi / 16) * cosf((2 * y + 1) * v * pi / 16);
            d->lut[y * 8 + x][v * 8 + u] = cu * cv * cp;
        }
}
static void gen_huff_mask(OrcDecoder* d)                                  /* GenLookupHuffMask :874-883 */
{
    for (unsigned len = 0; len < 32; len++) {
        unsigned m = (1u << len) - 1; m = len ? m << (32 - len) : 0; d->huff_mask[len] = m;
    }
}

/* ------------------------------------------------------------ state resets */
static void restart_dc_state(OrcDecoder* d)                               /* DecodeRestartDcState :2693-2703 */
{
    d->dc_y = d->dc_cb = d->dc_cr = 0;
    memset(d->dc_y_css, 0, sizeof d->dc_y_css); memset(d->dc_cb_css, 0, sizeof d->dc_cb_css);
    memset(d->dc_cr_css, 0, sizeof d->dc_cr_css);
}
static void restart_scan_buf(OrcDecoder* d, unsigned file_pos, int restart) /* DecodeRestartScanBuf :4038-4075 */
{
    d->scan_end = 0; d->scan_bad = 0; d->buff = 0; d->ptr = file_pos;
    if (!restart) d->ptr_first = file_pos;
    d->align = 0;
    for (int i = 0; i < 4; i++) { d->pos[i] = 0; d->err[i] = SB_OK; }
    d->latch_err = SB_OK; d->num = 0; d->vacant = 32; d->cur_err = 0;
    d->restart_read = 0; d->mcus_left = d->rst_interval;
}
static void free_outputs(OrcDecoder* d)
{
    free(d->mcu_map); d->mcu_map = NULL;
    for (int c = 0; c < 3; c++) { free(d->blk_dc[c]); d->blk_dc[c] = NULL; free(d->pix[c]); d->pix[c] = NULL; }
}
void orc_reset(OrcDecoder* d)                                             /* Reset :49-138 */
{
    restart_scan_buf(d, 0, 0); restart_dc_state(d);
    d->restart_read = 0; d->restart_count = 0;
    d->img_x = d->img_y = d->mcu_xmax = d->mcu_ymax = d->blk_xmax = d->blk_ymax = 0;
    d->bright_valid = 0; d->bright_y = d->bright_cb = d->bright_cr = -32768;
    d->bright_r = d->bright_g = d->bright_b = 0; d->bright_mx = d->bright_my = 0;
    d->avg_valid = 0; d->avg_y = 0;
    if (d->dib_ready) { free(d->dib); d->dib = NULL; d->dib_ready = 0; }
    free_outputs(d);
    d->warn_bad = 0; d->warn_ycc_clip = 0;                                /* :130 */
}
static void reset_dqt(OrcDecoder* d)                                      /* ResetDqtTables :343-360 */
{
    for (int i = 0; i < 256; i++) d->dqt_sel[i] = -1;
    memset(d->dqt_nat, 0, sizeof d->dqt_nat); memset(d->dqt_zz, 0, sizeof d->dqt_zz);
    d->num_sof = 0;
}
static void reset_dht(OrcDecoder* d)                                      /* ResetDhtLookup :373-406 */
{
    memset(d->dht_histo, 0, sizeof d->dht_histo);
    memset(d->dht_setmax, 0, sizeof d->dht_setmax); memset(d->dht_size, 0, sizeof d->dht_size);
    memset(d->dht_bitlen, 0, sizeof d->dht_bitlen); memset(d->dht_bits, 0, sizeof d->dht_bits);
    memset(d->dht_mask, 0, sizeof d->dht_mask); memset(d->dht_code, 0, sizeof d->dht_code);
    memset(d->dht_fast, 0xFF, sizeof d->dht_fast);
    for (int c = 0; c < DHT_CLASSES; c++) for (int i = 0; i < 5; i++) d->dht_sel[c][i] = -1;
    d->num_sos = 0;
}
void orc_reset_state(OrcDecoder* d)                                       /* ResetState :286-306 */
{
    reset_dht(d); reset_dqt(d);
    memset(d->samp_h, 0, sizeof d->samp_h); memset(d->samp_v, 0, sizeof d->samp_v);
    d->details_set = 0; d->num_sof = 0; d->precision = 0;
}
OrcDecoder* orc_create(void)                                              /* ctor :142-234 */
{
    OrcDecoder* d = (OrcDecoder*)calloc(1, sizeof *d);
    d->opt_decode_ac = 1; d->opt_err_max = 20;     /* north-star mode: Full IDCT, histogram off */
    orc_reset(d);
    d->mcu_w = d->mcu_h = 1; d->decode_ac = 1;
    precalc_idct(d); gen_huff_mask(d);
    orc_reset_state(d);
    d->preview_mode = PV_RGB;
    return d;
}
void orc_destroy(OrcDecoder* d) { if (!d) return; free_outputs(d); free(d->dib); free(d->coef); free(d); }
void orc_set_options(OrcDecoder* d, int ac, int histo, int clip, unsigned err_max)
{ d->opt_decode_ac = ac; d->opt_histo_en = histo; d->opt_stat_clip_en = clip; d->opt_err_max = err_max; }

/* ------------------------------------------------------------------ setters */
int orc_set_dqt_entry(OrcDecoder* d, unsigned tbl, unsigned nat, unsigned zz, unsigned val)   /* :424-453 */
{
    if (tbl < 4 && nat < 64) { d->dqt_nat[tbl][nat] = (uint16_t)val; d->dqt_zz[tbl][zz & 63] = (uint16_t)val; return 1; }
    return 0;
}
unsigned orc_get_dqt_entry(OrcDecoder* d, unsigned tbl, unsigned nat) { return (tbl < 4 && nat < 64) ? d->dqt_nat[tbl][nat] : 0; }
int orc_set_dqt_tables(OrcDecoder* d, unsigned comp, unsigned tbl)                             /* :505-520 */
{ if (comp < 256 && tbl < 4) { d->dqt_sel[comp] = (int)tbl; return 1; } return 0; }
int orc_set_dht_tables(OrcDecoder* d, unsigned comp, unsigned dc, unsigned ac)                 /* :536-553 */
{ if (comp >= 1 && comp < 5 && dc < 4 && ac < 4) { d->dht_sel[0][comp] = (int)dc; d->dht_sel[1][comp] = (int)ac; return 1; } return 0; }
int orc_set_dht_entry(OrcDecoder* d, unsigned dest, unsigned cls, unsigned ind, unsigned len,
                      unsigned bits, unsigned mask, unsigned code)                             /* :748-820 */
{
    if (dest >= DHT_DESTS || cls >= DHT_CLASSES || ind >= DHT_CODES) return 0;
    d->dht_bitlen[cls][dest][ind] = len; d->dht_bits[cls][dest][ind] = bits;
    d->dht_mask[cls][dest][ind] = mask;  d->dht_code[cls][dest][ind] = code;
    if (dest > d->dht_setmax[cls]) d->dht_setmax[cls] = dest;
    if (len <= DHT_FAST_BITS) {                   /* every 9-bit prefix extension maps to (len<<8)+code */
        unsigned lo = (bits & mask) >> (32 - DHT_FAST_BITS), hi = lo + ((1u << (DHT_FAST_BITS - len)) - 1);
        for (unsigned i = lo; i <= hi; i++) d->dht_fast[cls][dest][i] = code + (len << 8);
    }
    return 1;
}
int orc_set_dht_size(OrcDecoder* d, unsigned dest, unsigned cls, unsigned n)                   /* :834-847 */
{ if (dest >= DHT_DESTS || cls >= DHT_CLASSES || n >= DHT_CODES) return 0; d->dht_size[cls][dest] = n; return 1; }
void orc_set_precision(OrcDecoder* d, unsigned p) { d->precision = p; }                        /* :564 */
void orc_set_sof_samp_factors(OrcDecoder* d, unsigned comp, unsigned h, unsigned v)            /* :619-624 */
{ if (comp < 256) { d->samp_h[comp] = h; d->samp_v[comp] = v; } }
void orc_set_image_details(OrcDecoder* d, unsigned x, unsigned y, unsigned nf, unsigned ns, int rst_en, unsigned rst_int) /* :590-599 */
{ d->details_set = 1; d->dim_x = x; d->dim_y = y; d->num_sof = nf; d->num_sos = ns; d->rst_en = rst_en != 0; d->rst_interval = rst_int; }

/* --------------------------------------------------------------- bit reader */
static unsigned file_byte(const OrcDecoder* d, unsigned long off)         /* CwindowBuf::Buf, WindowBuf.cpp:639-714 */
{ return off < d->flen ? d->file[off] : 0; }

static void buf_consume(OrcDecoder* d, unsigned nbits)                    /* ScanBuffConsume :921-955 */
{
    d->buff = nbits >= 32 ? 0 : d->buff << nbits; d->vacant += nbits;
    unsigned nbytes = (d->align + nbits) / 8;
    for (unsigned i = 0; i < nbytes; i++) {
        d->pos[0] = d->pos[1]; d->pos[1] = d->pos[2]; d->pos[2] = d->pos[3];
        d->err[0] = d->err[1]; d->err[1] = d->err[2]; d->err[2] = d->err[3]; d->err[3] = SB_OK;
        if (d->err[0] != SB_OK) d->latch_err = d->err[0];
        d->num--;
    }
    d->align = (d->align + nbits) % 8;
}
static void buf_add(OrcDecoder* d, unsigned byte, unsigned ptr)           /* ScanBuffAdd :974-988 */
{
    d->buff += byte << (d->vacant - 8); d->vacant -= 8;
    if (d->num >= 4) return;
    d->err[d->num] = SB_OK; d->pos[d->num++] = ptr;
}
static void buf_add_err(OrcDecoder* d, unsigned byte, unsigned ptr, unsigned e) /* ScanBuffAddErr :999-1004 */
{ buf_add(d, byte, ptr); d->err[(d->num - 1) & 3] = e; }

static unsigned buf_add_byte(OrcDecoder* d)                               /* BuffAddByte :1386-1573 */
{
    if (d->restart_read) return 0;
    unsigned b0 = file_byte(d, d->ptr), b1 = file_byte(d, d->ptr + 1), marker = 0;
    if (b0 == 0xFF) {
        marker = b1;
        if (marker >= 0xD0 && marker <= 0xD7) {   /* RSTn: stop feeding until the block loop handles it */
            d->restart_count++; d->restart_last = marker - 0xD0;
            d->restart_expect = (d->restart_last + 1) % 8;
            d->restart_read = 1; return 0;
        }
    }
    if (b0 == 0xFF && b1 == 0x00)      { buf_add(d, b0, (unsigned)d->ptr); d->ptr += 2; }       /* stuffing */
    else if (b0 == 0xFF && b1 == 0xFF) { buf_add(d, b0, (unsigned)d->ptr); d->ptr += 1; }       /* quirk :1486-1525 */
    else if (b0 == 0xFF && marker != 0) {                                                        /* quirk :1527-1561 */
        if (d->warn_bad < d->opt_err_max) d->warn_bad++;
        buf_add_err(d, b0, (unsigned)d->ptr, SB_BADMARK); d->ptr += 1;
    } else                             { buf_add(d, b0, (unsigned)d->ptr); d->ptr += 1; }
    return 0;
}
static void buf_topup(OrcDecoder* d)                                      /* BuffTopup :1292-1323 */
{
    int done = d->vacant < 8;
    if (d->scan_end) done = 1;
    while (!done) {
        unsigned r = buf_add_byte(d);
        if (d->restart_read) done = 1;
        if (d->vacant < 8) done = 1;
        if (r != 0) done = 1;
    }
}
static int huff_extend(unsigned v, unsigned nbits)                        /* HuffmanDc2Signed :859-866 */
{ return v >= (1u << (nbits - 1)) ? (int)v : (int)(v - ((1u << nbits) - 1)); }

static int read_scan_val(OrcDecoder* d, unsigned cls, unsigned tbl, unsigned* zrl, int* val)    /* ReadScanVal :1072-1286 */
{
    unsigned code = CODE_UNUSED, ind = 0; int done, found = 0;
    d->used1 = d->used2 = 0; *zrl = 0; *val = 0;
    if (d->vacant == 32 && d->restart_read) return RSV_RST_TERM;
    if (d->vacant >= 32) {                                       /* overread before code */
        if (d->warn_bad < d->opt_err_max) d->warn_bad++;
        d->scan_end = 1; d->scan_bad = 1; return RSV_UNDERFLOW;
    }
    buf_topup(d);
    done = 0;
    if ((32 - d->vacant) >= DHT_FAST_BITS) {                     /* 9-bit direct lookup :1131-1141 */
        uint32_t f = d->dht_fast[cls][tbl][d->buff >> (32 - DHT_FAST_BITS)];
        if (f != CODE_UNUSED) { d->used1 += f >> 8; code = f & 0xFF; done = 1; found = 1; }
    }
    while (!done) {                                              /* linear search :1145-1164 */
        if ((d->buff & d->dht_mask[cls][tbl][ind]) == d->dht_bits[cls][tbl][ind]) {
            unsigned bl = d->dht_bitlen[cls][tbl][ind];
            if (bl <= 32 - d->vacant) { code = d->dht_code[cls][tbl][ind]; d->used1 += bl; done = 1; found = 1; }
        }
        ind++;
        if (ind >= d->dht_size[cls][tbl]) done = 1;
    }
    if (!found) {
        if (d->restart_read) return RSV_RST_TERM;
        d->used1 = 1; code = CODE_UNUSED;                        /* resync by one bit :1185 */
    }
    if (d->used1 < 17) d->dht_histo[cls][tbl][d->used1]++;
    buf_consume(d, d->used1);
    if (d->vacant > 32) { d->scan_end = 1; d->scan_bad = 1; return RSV_UNDERFLOW; }
    buf_topup(d);
    if (code != CODE_UNUSED) {
        *zrl = (code & 0xF0) >> 4; d->used2 = code & 0x0F;
        if (*zrl == 0 && d->used2 == 0) return RSV_EOB;
        if (d->used2 == 0) { *val = 0; return RSV_OK; }
        unsigned v = (d->buff & d->huff_mask[d->used2]) >> (32 - d->used2);   /* ExtractBits :898-903 */
        *val = huff_extend(v, d->used2);
        if (d->precision >= 8) { int div = 1 << ((d->precision - 8) & 31); *val /= div; }
        buf_consume(d, d->used2);
        if (d->vacant > 32) { d->scan_end = 1; d->scan_bad = 1; return RSV_UNDERFLOW; }
        return RSV_OK;
    }
    if (d->warn_bad < d->opt_err_max) d->warn_bad++;
    d->scan_bad = 1;
    return RSV_UNDERFLOW;
}

/* ----------------------------------------------------------- block decoding */
static void idct_set(OrcDecoder* d, unsigned dqt, unsigned ncoef, unsigned zrl, int16_t val)    /* DecodeIdctSet :2270-2303 */
{
    unsigned ind = ncoef + zrl;
    if (ind >= 64) return;
    d->dct[kZigZag[ind]] = (int16_t)((int)val * (int)d->dqt_zz[dqt & 3][ind]);
}
static void idct_calc_float(OrcDecoder* d)                                                      /* DecodeIdctCalcFloat(64) :2372-2392 */
{
    for (unsigned yx = 0; yx < 64; yx++) {
        float sum = 0;
        for (unsigned vu = 1; vu < 64; vu++) sum += d->lut[yx][vu] * d->dct[vu];
        sum *= 0.25;
        d->idct[yx] = sum;
    }
}
static void coef_record(OrcDecoder* d)
{
    if (d->coef_blocks == d->coef_cap) { d->coef_cap = d->coef_cap ? d->coef_cap * 2 : 4096;
        d->coef = (int16_t*)realloc(d->coef, d->coef_cap * 64 * sizeof(int16_t)); }
    memcpy(d->coef + d->coef_blocks * 64, d->dct, sizeof d->dct); d->coef_blocks++;
}
static int decode_scan_comp(OrcDecoder* d, unsigned tdc, unsigned tac, unsigned tq)              /* DecodeScanComp :1604-1835 */
{
    unsigned zrl, ncoef = 0; int val, done = 0, is_dc = 1, ret = 1;
    memset(d->dct, 0, sizeof d->dct); memset(d->idct, 0, sizeof d->idct);                       /* DecodeIdctClear :2243 */
    while (!done) {
        buf_topup(d);
        unsigned saved_err = d->latch_err;
        int r = read_scan_val(d, is_dc ? 0 : 1, is_dc ? tdc : tac, &zrl, &val);
        if (r == RSV_RST_TERM) {                                 /* marker-driven restart :1644-1680 */
            restart_dc_state(d);
            d->ptr += 2;
            restart_scan_buf(d, (unsigned)d->ptr, 1);
            d->restart_read = 0;
            buf_topup(d);
            r = read_scan_val(d, is_dc ? 0 : 1, is_dc ? tdc : tac, &zrl, &val);
        }
        if (saved_err == SB_BADMARK) {                           /* :1683-1706 */
            d->cur_err = 1; d->scan_bad = 1;
            if (d->warn_bad < d->opt_err_max) d->warn_bad++;
            d->latch_err = SB_OK;
        }
        int16_t v16 = (int16_t)(val & 0xFFFF);
        if (r == RSV_OK) {
            if (is_dc) { idct_set(d, tq, ncoef, zrl, v16); is_dc = 0; }
            else if (d->decode_ac) idct_set(d, tq, ncoef, zrl, v16);
        } else if (r == RSV_EOB) {
            if (is_dc) { idct_set(d, tq, ncoef, zrl, v16); is_dc = 0; }
            else done = 1;
        } else if (r == RSV_UNDERFLOW) {                         /* returns before the IDCT :1737-1757 */
            if (d->warn_bad < d->opt_err_max) d->warn_bad++;
            d->cur_err = 1; ret = 0; goto out;
        }
        ncoef += 1 + zrl;
        if (ncoef == 64) done = 1;
        else if (ncoef > 64) {
            if (d->warn_bad < d->opt_err_max) d->warn_bad++;
            d->cur_err = 1; d->scan_bad = 1; done = 1; ncoef = 64;
        }
    }
    if (d->decode_ac) idct_calc_float(d);
out:
    coef_record(d);
    return ret;
}
static void check_scan_errors(OrcDecoder* d)                                                    /* CheckScanErrors :2605-2660 */
{ if (d->cur_err) { if (d->warn_bad < d->opt_err_max) d->warn_bad++; d->cur_err = 0; } }

static void set_full_res(OrcDecoder* d, unsigned mx, unsigned my, unsigned comp, unsigned cx, unsigned cy, int16_t dcoff) /* SetFullRes :2468-2561 */
{
    if (comp < 1 || comp > 3) return;
    int16_t* plane = d->pix[comp - 1];
    const unsigned w = d->blk_xmax * 8, eh = d->expand_h[comp], ev = d->expand_v[comp];
    unsigned corner = (my * d->mcu_h + cy * 8) * w + (mx * d->mcu_w + cx * 8);
    for (unsigned y = 0; y < 8; y++) {
        for (unsigned x = 0; x < 8; x++) {
            float f = d->idct[y * 8 + x];
            int16_t s = (int16_t)((int16_t)(int32_t)(f * 8) + dcoff);       /* truncate to int, wrap to i16, add, wrap */
            unsigned pc = corner + x * eh;
            for (unsigned j = 0; j < ev; j++) for (unsigned i = 0; i < eh; i++) plane[pc + j * w + i] = s;
        }
        corner += w * ev;
    }
}

/* --------------------------------------------------------- colour conversion */
typedef struct { int pre_y, pre_cb, pre_cr; uint8_t fy, fcb, fcr, r, g, b; } PixCc;
static void ycc_to_rgb_fast_float(PixCc* p)                                                     /* ConvertYCCtoRGBFastFloat :4086-4139 */
{
    int cy = p->pre_y >> 3, ccb = p->pre_cb >> 3, ccr = p->pre_cr >> 3;
    int y = cy < -128 ? -128 : cy > 127 ? 127 : cy, cb = ccb < -128 ? -128 : ccb > 127 ? 127 : ccb,
        cr = ccr < -128 ? -128 : ccr > 127 ? 127 : ccr;
    p->fy = (uint8_t)(y + 128); p->fcb = (uint8_t)(cb + 128); p->fcr = (uint8_t)(cr + 128);
    const float kr = 0.299f, kg = 0.587f, kb = 0.114f;
    float r = cr * (2 - 2 * kr) + y;
    float b = cb * (2 - 2 * kb) + y;
    float g = (y - kb * b - kr * r) / kg;
    r += 128; b += 128; g += 128;
    p->r = (r < 0) ? 0 : (r > 255) ? 255 : (uint8_t)r;
    p->g = (g < 0) ? 0 : (g > 255) ? 255 : (uint8_t)g;
    p->b = (b < 0) ? 0 : (b > 255) ? 255 : (uint8_t)b;
}
/* min / max / sum triplet of PixelCcHisto (ints; the sums wrap like the reference's int adds do in practice) */
static void mms(int32_t* t, int v) { if (v < t[0]) t[0] = v; if (v > t[1]) t[1] = v; t[2] = (int32_t)((uint32_t)t[2] + (uint32_t)v); }
enum { CL_Y_UNDER, CL_Y_OVER, CL_CB_UNDER, CL_CB_OVER, CL_CR_UNDER, CL_CR_OVER, CL_R_UNDER, CL_R_OVER, CL_G_UNDER, CL_G_OVER, CL_B_UNDER, CL_B_OVER, CL_WHITE };
enum { H_PRE_Y = 0, H_PRE_CB = 3, H_PRE_CR = 6, H_CLIP_Y = 9, H_CLIP_CB = 12, H_CLIP_CR = 15, H_CLIP_R = 18, H_CLIP_G = 21, H_CLIP_B = 24,
       H_PRE_R = 27, H_PRE_G = 30, H_PRE_B = 33 };
/* one YCC range check of CapYccRange: the counter only moves while fewer than YCC_CLIP_REPORT_MAX (10) warnings
 * have been issued since Reset() (:4372-4378) -- the clip itself always happens */
static int cap_ycc(OrcDecoder* d, int v, int under_ix, int over_ix)
{
    if (v > 255) { if (d->warn_ycc_clip < 10) { d->warn_ycc_clip++; d->cc_clip[over_ix]++; } v = 255; }
    if (v < 0)   { if (d->warn_ycc_clip < 10) { d->warn_ycc_clip++; d->cc_clip[under_ix]++; } v = 0; }
    return v;
}
static void ycc_to_rgb_histo(OrcDecoder* d, PixCc* p)      /* ConvertYCCtoRGB :4229-4326, CapYccRange :4341-4475, CapRgbRange :4495-4601 */
{
    int32_t* H = d->cc_histo;
    if (d->hist_en) {
        mms(H + H_PRE_Y, p->pre_y); mms(H + H_PRE_CB, p->pre_cb); mms(H + H_PRE_CR, p->pre_cr);
        int hi = p->pre_y; if (hi < -1024) hi = -1024; if (hi > 1023) hi = 1023;
        d->cc_yfull[hi + 1024]++;
    }
    int cy = (p->pre_y + 1024) / 8, ccb = (p->pre_cb + 1024) / 8, ccr = (p->pre_cr + 1024) / 8;   /* C division: truncates toward zero */
    if (d->hist_en) { mms(H + H_CLIP_Y, cy); mms(H + H_CLIP_CB, ccb); mms(H + H_CLIP_CR, ccr); d->cc_count++; }
    cy = cap_ycc(d, cy, CL_Y_UNDER, CL_Y_OVER); ccb = cap_ycc(d, ccb, CL_CB_UNDER, CL_CB_OVER); ccr = cap_ycc(d, ccr, CL_CR_UNDER, CL_CR_OVER);
    p->fy = (uint8_t)cy; p->fcb = (uint8_t)ccb; p->fcr = (uint8_t)ccr;
    const int vy = cy - 128, vcb = ccb - 128, vcr = ccr - 128;
    const float kr = 0.299f, kg = 0.587f, kb = 0.114f;
    float r = vcr * (2 - 2 * kr) + vy;
    float b = vcb * (2 - 2 * kb) + vy;
    float g = (vy - kb * b - kr * r) / kg;
    r += 128; b += 128; g += 128;
    int lr = (int)r, lg = (int)g, lb = (int)b;                                                     /* truncate first, then range-check */
    if (d->hist_en) { mms(H + H_PRE_R, lr); mms(H + H_PRE_G, lg); mms(H + H_PRE_B, lb); }
    if (lr < 0) { d->cc_clip[CL_R_UNDER]++; lr = 0; }
    if (lg < 0) { d->cc_clip[CL_G_UNDER]++; lg = 0; }
    if (lb < 0) { d->cc_clip[CL_B_UNDER]++; lb = 0; }
    if (lr > 255) { d->cc_clip[CL_R_OVER]++; lr = 255; }
    if (lg > 255) { d->cc_clip[CL_G_OVER]++; lg = 255; }
    if (lb > 255) { d->cc_clip[CL_B_OVER]++; lb = 255; }
    if (d->hist_en) { mms(H + H_CLIP_R, lr); mms(H + H_CLIP_G, lg); mms(H + H_CLIP_B, lb); }
    p->r = (uint8_t)lr; p->g = (uint8_t)lg; p->b = (uint8_t)lb;
    if (d->hist_en) { d->cc_rgb[0][p->r / 2]++; d->cc_rgb[1][p->g / 2]++; d->cc_rgb[2][p->b / 2]++; }   /* 256 / HISTO_BINS = 2 */
}
static void channel_extract(unsigned mode, const PixCc* s, uint8_t* r, uint8_t* g, uint8_t* b)  /* ChannelExtract :4832-4872 */
{
    switch (mode) {
    case PV_YCC: *r = s->fcr; *g = s->fy; *b = s->fcb; break;
    case PV_R:   *r = *g = *b = s->r; break;
    case PV_G:   *r = *g = *b = s->g; break;
    case PV_B:   *r = *g = *b = s->b; break;
    case PV_Y:   *r = *g = *b = s->fy; break;
    case PV_CB:  *r = *g = *b = s->fcb; break;
    case PV_CR:  *r = *g = *b = s->fcr; break;
    default:     *r = s->r; *g = s->g; *b = s->b; break;
    }
}
static void calc_channel_preview(OrcDecoder* d)                                                 /* CalcChannelPreview(Full) :4965, :4619-4821 */
{
    if (!d->dib) return;
    const unsigned W = d->img_x, H = d->img_y, pw = d->blk_xmax * 8, row = W * 4;
    const unsigned mcus_across = W / d->mcu_w, shift_ind = d->shift_mcu_y * mcus_across + d->shift_mcu_x;
    unsigned sum_y = 0; unsigned long npix = (unsigned)((H + 1) * (W + 1));
    d->bright_valid = 0; d->bright_y = d->bright_cb = d->bright_cr = -32768; d->avg_valid = 0; d->avg_y = 0;
    for (unsigned py = 0; py < H; py++) {
        unsigned my = py / d->mcu_h, inv = (H - 1) - py;
        for (unsigned px = 0; px < W; px++) {
            unsigned pi = py * pw + px, mx = px / d->mcu_w, mi = my * mcus_across + mx;
            int ty = d->pix[0][pi], tcb = 0, tcr = 0;
            if (d->num_sos == 3) { tcb = d->pix[1][pi]; tcr = d->pix[2][pi]; }
            PixCc s; s.pre_y = ty; s.pre_cb = tcb; s.pre_cr = tcr;
            if (ty > d->bright_y) { d->bright_y = ty; d->bright_cb = tcb; d->bright_cr = tcr; d->bright_mx = (int)mx; d->bright_my = (int)my; }
            if (mi >= shift_ind) { s.pre_y += d->shift_y; s.pre_cb += d->shift_cb; s.pre_cr += d->shift_cr; }
            if (d->hist_en || d->stat_clip_en) ycc_to_rgb_histo(d, &s); else ycc_to_rgb_fast_float(&s);   /* :4742-4747 */
            sum_y += s.fy;
            uint8_t r, g, b; channel_extract(d->preview_mode, &s, &r, &g, &b);
            uint8_t* o = d->dib + (size_t)px * 4 + (size_t)inv * row;
            o[3] = 0; o[2] = r; o[1] = g; o[0] = b;
        }
    }
    d->bright_valid = 1;
    PixCc s; s.pre_y = d->bright_y; s.pre_cb = d->bright_cb; s.pre_cr = d->bright_cr; ycc_to_rgb_fast_float(&s);
    d->bright_r = s.r; d->bright_g = s.g; d->bright_b = s.b;
    if (npix == 0) npix = 1;
    d->avg_y = (long)(sum_y / npix); d->avg_valid = 1;
}

/* ------------------------------------------------------------- scan driver */
void orc_decode_scan_img(OrcDecoder* d, const uint8_t* file, size_t len, unsigned start, int display, int quiet) /* DecodeScanImg :2723-3745 */
{
    (void)quiet;
    d->file = file; d->flen = len; d->coef_blocks = 0;
    int want_ac = display ? d->opt_decode_ac : 0;
    d->hist_en = d->opt_histo_en; d->stat_clip_en = d->opt_stat_clip_en;                          /* :2740-2741 */
    orc_reset(d);
    d->decode_ac = want_ac;
    if (!d->details_set) return;
    if (d->num_sos != 1 && d->num_sos != 3) return;
    d->samp_hmax = d->samp_vmax = 0; d->samp_hmin = d->samp_vmin = 0xFF;
    for (unsigned c = 1; c <= d->num_sos; c++) {
        if (d->samp_h[c] > d->samp_hmax) d->samp_hmax = d->samp_h[c];
        if (d->samp_v[c] > d->samp_vmax) d->samp_vmax = d->samp_v[c];
        if (d->samp_h[c] < d->samp_hmin) d->samp_hmin = d->samp_h[c];
        if (d->samp_v[c] < d->samp_vmin) d->samp_vmin = d->samp_v[c];
    }
    if (d->num_sos == 1) { d->samp_h[1] = d->samp_v[1] = 1; d->samp_hmax = d->samp_vmax = d->samp_hmin = d->samp_vmin = 1; } /* :2805-2817 */
    if (d->samp_hmax == 0 || d->samp_vmax == 0 || d->samp_hmax > MAX_SAMP || d->samp_vmax > MAX_SAMP) return;
    d->mcu_w = d->samp_hmax * 8; d->mcu_h = d->samp_vmax * 8;
    for (unsigned c = 1; c <= d->num_sos; c++) {
        if (d->samp_h[c] == 0 || d->samp_v[c] == 0) return;      /* the reference would divide by zero here (:2837) */
        d->expand_h[c] = d->samp_hmax / d->samp_h[c]; d->expand_v[c] = d->samp_vmax / d->samp_v[c];
        d->per_mcu_h[c] = d->samp_h[c]; d->per_mcu_v[c] = d->samp_v[c];
    }
    d->mcu_xmax = d->dim_x / d->mcu_w; d->mcu_ymax = d->dim_y / d->mcu_h;
    if (d->dim_x % d->mcu_w) d->mcu_xmax++;
    if (d->dim_y % d->mcu_h) d->mcu_ymax++;
    d->blk_xmax = d->mcu_xmax * d->samp_hmax; d->blk_ymax = d->mcu_ymax * d->samp_vmax;
    if (d->blk_xmax == 0 || d->blk_ymax == 0) return;
    d->img_x = d->mcu_xmax * d->mcu_w; d->img_y = d->mcu_ymax * d->mcu_h;

    const size_t nmcu = (size_t)d->mcu_xmax * d->mcu_ymax, nblk = (size_t)d->blk_xmax * d->blk_ymax;
    const unsigned pw = d->blk_xmax * 8, ph = d->blk_ymax * 8;
    d->mcu_map = (uint32_t*)calloc(nmcu, sizeof(uint32_t));
    for (unsigned c = 0; c < (d->num_sos == 3 ? 3u : 1u); c++) {
        d->blk_dc[c] = (int16_t*)calloc(nblk, sizeof(int16_t));
        d->pix[c] = (int16_t*)calloc((size_t)pw * ph, sizeof(int16_t));
    }
    free(d->dib); d->dib = NULL; d->dib_ready = 0; d->preview_is_jpeg = 0;
    if (display) d->dib = (uint8_t*)calloc((size_t)d->img_x * d->img_y, 4);       /* CDIB::CreateDIB, Dib.cpp:53-88 */
    if (display) {                                                                /* :3145-3155 */
        memset(d->cc_histo, 0, sizeof d->cc_histo); d->cc_count = 0; memset(d->cc_clip, 0, sizeof d->cc_clip);
        memset(d->cc_rgb, 0, sizeof d->cc_rgb); memset(d->cc_yfull, 0, sizeof d->cc_yfull);
    }

    restart_dc_state(d);
    restart_scan_buf(d, start, 0);
    d->restart_expect = 0; d->restart_last = 0;
    buf_topup(d);
    if (d->num_sof != 1 && d->num_sof != 3) return;
    for (unsigned i = 1; i <= d->num_sos; i++) if (d->dqt_sel[i] < 0) return;                   /* :3047-3055 */
    const unsigned qy = (unsigned)d->dqt_sel[1], qcb = (unsigned)d->dqt_sel[2], qcr = (unsigned)d->dqt_sel[3];
    int ready = 1;
    for (unsigned cls = 0; cls < 2; cls++) for (unsigned i = 1; i <= d->num_sos; i++) if (d->dht_sel[cls][i] < 0) ready = 0;
    for (unsigned i = 1; i <= d->num_sos; i++) for (unsigned cls = 0; cls < 2; cls++) {
        unsigned sel = (unsigned)d->dht_sel[cls][i];
        if (sel >= DHT_DESTS || d->dht_size[cls][sel] == 0) ready = 0;
    }
    if (!ready) return;                                                                          /* :3098-3103 */
    const unsigned hdc[4] = {0, (unsigned)d->dht_sel[0][1], (unsigned)d->dht_sel[0][2], (unsigned)d->dht_sel[0][3]};
    const unsigned hac[4] = {0, (unsigned)d->dht_sel[1][1], (unsigned)d->dht_sel[1][2], (unsigned)d->dht_sel[1][3]};
    const unsigned tq[4] = {0, qy, qcb, qcr};
    d->num_pixels = 0;

    for (unsigned my = 0; my < d->mcu_ymax; my++) {
        int stop = 0;
        for (unsigned mx = 0; mx < d->mcu_xmax && !stop; mx++) {
            d->decode_ac = want_ac;
            const unsigned mi = my * d->mcu_xmax + mx;
            d->mcu_map[mi] = (d->pos[0] << 4) + d->align;                                       /* PackFileOffset :5104 */
            for (unsigned comp = 1; comp <= d->num_sos; comp++) {                                /* Y, then Cb, then Cr :3263-3405 */
                int16_t* acc = comp == 1 ? &d->dc_y : comp == 2 ? &d->dc_cb : &d->dc_cr;
                int16_t* css = comp == 1 ? d->dc_y_css : comp == 2 ? d->dc_cb_css : d->dc_cr_css;
                for (unsigned cv = 0; cv < d->per_mcu_v[comp]; cv++) for (unsigned ch = 0; ch < d->per_mcu_h[comp]; ch++) {
                    decode_scan_comp(d, hdc[comp], hac[comp], tq[comp]);
                    if (d->cur_err) check_scan_errors(d);
                    *acc = (int16_t)(*acc + d->dct[0]);
                    css[cv * MAX_SAMP + ch] = *acc;
                    if (display) set_full_res(d, mx, my, comp, ch, cv, *acc);
                    if (comp == 1) d->num_pixels += 64;
                }
            }
            /* per-block cumulative DC maps :3524-3608 (note the Y corner uses expand, not samples-per-MCU) */
            {
                unsigned cornx = mx * d->expand_h[1], corny = my * d->expand_v[1], lin = corny * d->blk_xmax + cornx;
                for (unsigned cv = 0; cv < d->per_mcu_v[1]; cv++) for (unsigned ch = 0; ch < d->per_mcu_h[1]; ch++) {
                    unsigned b = lin + cv * d->blk_xmax + ch;
                    if (b < nblk) d->blk_dc[0][b] = d->dc_y_css[cv * MAX_SAMP + ch];
                }
                if (d->num_sos == 3) for (unsigned comp = 2; comp <= 3; comp++) {
                    const int16_t* css = comp == 2 ? d->dc_cb_css : d->dc_cr_css;
                    for (unsigned cv = 0; cv < d->per_mcu_v[comp]; cv++) for (unsigned ch = 0; ch < d->per_mcu_h[comp]; ch++) {
                        unsigned b = (my * d->expand_v[comp] + cv) * d->blk_xmax + (mx * d->expand_h[comp] + ch);
                        if (b < nblk) d->blk_dc[comp - 1][b] = css[cv * MAX_SAMP + ch];
                    }
                }
            }
            if (d->rst_en) d->mcus_left--;
            if (d->scan_end && d->scan_bad) stop = 1;                                            /* :3623-3625 */
        }
    }
    if (display) { calc_channel_preview(d); d->dib_ready = 1; d->preview_is_jpeg = 1; }
}

/* ------------------------------------------------------------------ getters */
int  orc_is_preview_ready(OrcDecoder* d) { return d->preview_is_jpeg; }
void orc_set_preview_mode(OrcDecoder* d, unsigned mode) { d->preview_mode = mode; calc_channel_preview(d); }              /* SetPreviewMode :633-639 */
unsigned orc_get_preview_mode(OrcDecoder* d) { return d->preview_mode; }
void orc_set_preview_ycc_offset(OrcDecoder* d, unsigned mx, unsigned my, int y, int cb, int cr)                        /* SetPreviewYccOffset :650-659 */
{ d->shift_y = y; d->shift_cb = cb; d->shift_cr = cr; d->shift_mcu_x = mx; d->shift_mcu_y = my; calc_channel_preview(d); }
void orc_get_image_size(OrcDecoder* d, unsigned* x, unsigned* y) { *x = d->img_x; *y = d->img_y; }
const uint8_t* orc_get_bitmap_ptr(OrcDecoder* d) { return d->dib; }
void orc_get_pixmap_ptrs(OrcDecoder* d, const int16_t** y, const int16_t** cb, const int16_t** cr)
{ *y = d->pix[0]; *cb = d->pix[1]; *cr = d->pix[2]; }
void orc_lookup_file_pos_mcu(OrcDecoder* d, unsigned mx, unsigned my, unsigned* byte, unsigned* bit)
{ uint32_t p = d->mcu_map[mx + my * d->mcu_xmax]; *bit = p & 7; *byte = p >> 4; }              /* UnpackFileOffset :5123 */
void orc_lookup_blk_ycc(OrcDecoder* d, unsigned bx, unsigned by, int* y, int* cb, int* cr)
{
    size_t i = bx + (size_t)by * d->blk_xmax; *y = d->blk_dc[0][i];
    if (d->num_sos == 3) { *cb = d->blk_dc[1][i]; *cr = d->blk_dc[2][i]; } else { *cb = 0; *cr = 0; }
}
void orc_get_geometry(OrcDecoder* d, unsigned* o)
{ o[0] = d->mcu_w; o[1] = d->mcu_h; o[2] = d->mcu_xmax; o[3] = d->mcu_ymax; o[4] = d->blk_xmax; o[5] = d->blk_ymax; o[6] = d->img_x; o[7] = d->img_y; }
const uint32_t* orc_mcu_file_map(OrcDecoder* d) { return d->mcu_map; }
void orc_blk_dc_ptrs(OrcDecoder* d, const int16_t** y, const int16_t** cb, const int16_t** cr)
{ *y = d->blk_dc[0]; *cb = d->blk_dc[1]; *cr = d->blk_dc[2]; }
const uint32_t* orc_dht_histo(OrcDecoder* d) { return &d->dht_histo[0][0][0]; }
void orc_scan_status(OrcDecoder* d, unsigned* o)
{ o[0] = (unsigned)d->scan_bad; o[1] = (unsigned)d->scan_end; o[2] = d->restart_count; o[3] = d->num_pixels;
  o[4] = d->pos[0]; o[5] = d->align; o[6] = d->warn_bad; o[7] = (unsigned)d->ptr_first; }
void orc_bright_avg(OrcDecoder* d, int* o)
{ o[0] = d->bright_valid; o[1] = d->bright_y; o[2] = d->bright_cb; o[3] = d->bright_cr; o[4] = (int)d->bright_r;
  o[5] = (int)d->bright_g; o[6] = (int)d->bright_b; o[7] = d->bright_mx; o[8] = d->bright_my; o[9] = (int)d->avg_y; }
/* [0..36] PixelCcHisto (36 ints + nCount), [37..49] PixelCcClip, [50..433] R,G,B 128-bin histograms, [434..2481] full Y histogram */
void orc_color_stats(OrcDecoder* d, unsigned* o)
{
    memcpy(o, d->cc_histo, 36 * 4); o[36] = d->cc_count; memcpy(o + 37, d->cc_clip, 13 * 4);
    memcpy(o + 50, d->cc_rgb, 3 * 128 * 4); memcpy(o + 434, d->cc_yfull, 2048 * 4);
}
/* ------------------------------------------------------------------ TIFF export
 * Pixel re-arrangement of CJPEGsnoopDoc::OnToolsExporttiff (source/JPEGsnoopDoc.cpp:2110-2180) and the container of
 * FileTiff::WriteFile / WriteIfd (source/FileTiff.cpp:281-433, :436-538): big-endian TIFF, one strip, tags in the
 * reference's order; values longer than 4 bytes go to an area behind the IFD; width / height / strip offset are
 * written as SHORT like the reference does.  mode 0 = RGB 8 bit, 1 = RGB 16 bit, 2 = YCC 8 bit. */
typedef struct { uint8_t* p; size_t n, cap; } TBuf;
static void tb_put(TBuf* b, const void* src, size_t n)
{ if (b->n + n > b->cap) { b->cap = (b->n + n) * 2 + 256; b->p = (uint8_t*)realloc(b->p, b->cap); } memcpy(b->p + b->n, src, n); b->n += n; }
static void tb8(TBuf* b, unsigned v) { uint8_t c = (uint8_t)v; tb_put(b, &c, 1); }
static void tb16(TBuf* b, unsigned v) { tb8(b, (v >> 8) & 0xFF); tb8(b, v & 0xFF); }
static void tb32(TBuf* b, unsigned v) { tb16(b, (v >> 16) & 0xFFFF); tb16(b, v & 0xFFFF); }
enum { TT_SHORT = 3, TT_LONG = 4, TT_RATIONAL = 5 };
static void tiff_single(TBuf* ifd, unsigned* count, unsigned tag, unsigned type, unsigned v)               /* WriteIfdEntrySingle :120 */
{ tb16(ifd, tag); tb16(ifd, type); tb32(ifd, 1); if (type == TT_SHORT) { tb16(ifd, v & 0xFFFF); tb16(ifd, 0); } else tb32(ifd, v); (*count)++; }
static void tiff_mult(TBuf* ifd, TBuf* extra, unsigned extra_ptr, unsigned* count, unsigned tag, unsigned type, unsigned n, const unsigned* vals) /* WriteIfdEntryMult :190 */
{
    const unsigned tlen = n * (type == TT_SHORT ? 2u : 4u); const int in_extra = tlen > 4;
    tb16(ifd, tag); tb16(ifd, type); tb32(ifd, type != TT_RATIONAL ? n : n / 2);
    if (in_extra) tb32(ifd, extra_ptr + (unsigned)extra->n);
    (*count)++;
    for (unsigned i = 0; i < n; i++) { TBuf* dst = in_extra ? extra : ifd; if (type == TT_SHORT) tb16(dst, vals[i] & 0xFFFF); else tb32(dst, vals[i]); }
    if (!in_extra && tlen < 4) for (unsigned k = 0; k < 4 - tlen; k++) tb8(ifd, 0);
}
static void tiff_ifd(TBuf* out, unsigned w, unsigned h, int ycc, int b16, unsigned ptr_img_in, unsigned num_in, unsigned extra_ptr, unsigned* num_out, unsigned* extra_ptr_out)
{   /* one pass of WriteIfd :281-433: entries, terminator, then the extra area */
    TBuf ifd = {0, 0, 0}, extra = {0, 0, 0}; unsigned count = 0, v[16];
    tb16(&ifd, num_in);
    tiff_single(&ifd, &count, 0x0100, TT_SHORT, w); tiff_single(&ifd, &count, 0x0101, TT_SHORT, h);
    v[0] = v[1] = v[2] = b16 ? 16 : 8; tiff_mult(&ifd, &extra, extra_ptr, &count, 0x0102, TT_SHORT, 3, v);
    tiff_single(&ifd, &count, 0x0103, TT_SHORT, 1);
    tiff_single(&ifd, &count, 0x0106, TT_SHORT, ycc ? 6 : 2);
    tiff_single(&ifd, &count, 0x0111, TT_SHORT, ptr_img_in);
    tiff_single(&ifd, &count, 0x0112, TT_SHORT, 1);
    tiff_single(&ifd, &count, 0x0115, TT_SHORT, 3);
    tiff_single(&ifd, &count, 0x0116, TT_SHORT, h);
    tiff_single(&ifd, &count, 0x0117, TT_LONG, h * w * (b16 ? 6u : 3u));
    v[0] = 72; v[1] = 1; tiff_mult(&ifd, &extra, extra_ptr, &count, 0x011A, TT_RATIONAL, 2, v); tiff_mult(&ifd, &extra, extra_ptr, &count, 0x011B, TT_RATIONAL, 2, v);
    tiff_single(&ifd, &count, 0x011C, TT_SHORT, 1); tiff_single(&ifd, &count, 0x0128, TT_SHORT, 2);
    if (ycc) {
        const unsigned c[6] = {299, 1000, 587, 1000, 114, 1000}; tiff_mult(&ifd, &extra, extra_ptr, &count, 0x0211, TT_RATIONAL, 6, c);
        v[0] = v[1] = 1; tiff_mult(&ifd, &extra, extra_ptr, &count, 0x0212, TT_SHORT, 2, v);
        tiff_single(&ifd, &count, 0x0213, TT_SHORT, 1);
    }
    { const unsigned bw[12] = {0, 1, 0xFF, 1, 0, 1, 0xFF, 1, 0, 1, 0xFF, 1}; tiff_mult(&ifd, &extra, extra_ptr, &count, 0x0214, TT_RATIONAL, 12, bw); }
    tb32(&ifd, 0);
    *num_out = count; *extra_ptr_out = 8 + (unsigned)ifd.n;
    tb_put(out, ifd.p, ifd.n); tb_put(out, extra.p, extra.n);
    free(ifd.p); free(extra.p);
}
int orc_export_tiff(OrcDecoder* d, const char* path, int mode)
{
    const int ycc = mode == 2, b16 = mode == 1;
    const unsigned W = d->img_x, Hh = d->img_y;
    if (!d->dib || !W || !Hh || (ycc && !(d->pix[0] && d->pix[1] && d->pix[2]))) return -1;
    TBuf out = {0, 0, 0};
    tb32(&out, 0x4D4D002A); tb32(&out, 8);
    unsigned num = 0, extra_ptr = 0, n1, e1;
    { TBuf pre = {0, 0, 0}; tiff_ifd(&pre, W, Hh, ycc, b16, 0, 0, 0, &n1, &e1); num = n1; extra_ptr = e1; unsigned end = 8 + (unsigned)pre.n; free(pre.p);
      tiff_ifd(&out, W, Hh, ycc, b16, end, num, extra_ptr, &n1, &e1); }          /* pass 2 knows the entry count, the extra area and the strip offset */
    const size_t npx = (size_t)W * Hh;
    for (unsigned y = 0; y < Hh; y++) for (unsigned x = 0; x < W; x++) {
        if (!ycc) {
            const uint8_t* s = d->dib + ((size_t)(Hh - 1 - y) * W + x) * 4;
            if (!b16) { tb8(&out, s[2]); tb8(&out, s[1]); tb8(&out, s[0]); }
            else { tb8(&out, s[2]); tb8(&out, 0); tb8(&out, s[1]); tb8(&out, 0); tb8(&out, s[0]); tb8(&out, 0); }   /* Swap16(v << 8) stored little-endian = bytes (v, 0) */
        } else {
            const size_t i = (size_t)y * W + x;
            for (int c = 0; c < 3; c++) { int v = d->pix[c][i]; if (v < -1024) v = -1024; if (v > 1023) v = 1023; tb8(&out, (unsigned)((0x0400 + v) >> 3)); }
        }
    }
    (void)npx;
    FILE* f = fopen(path, "wb"); if (!f) { free(out.p); return -1; }
    fwrite(out.p, 1, out.n, f); fclose(f); free(out.p);
    return 0;
}
const float* orc_idct_lut(OrcDecoder* d) { return &d->lut[0][0]; }
const uint32_t* orc_dht_lookupfast(OrcDecoder* d) { return &d->dht_fast[0][0][0]; }
const int16_t* orc_coef_ptr(OrcDecoder* d) { return d->coef; }
size_t orc_coef_blocks(OrcDecoder* d) { return d->coef_blocks; }
void orc_idct_block(OrcDecoder* d, const int16_t* c, float* out)
{ memcpy(d->dct, c, sizeof d->dct); idct_calc_float(d); memcpy(out, d->idct, sizeof d->idct); }
void orc_color_fast(const int* ycc, uint8_t* rgb, size_t n)
{
    for (size_t i = 0; i < n; i++) { PixCc p; p.pre_y = ycc[3 * i]; p.pre_cb = ycc[3 * i + 1]; p.pre_cr = ycc[3 * i + 2];
        ycc_to_rgb_fast_float(&p); rgb[3 * i] = p.r; rgb[3 * i + 1] = p.g; rgb[3 * i + 2] = p.b; }
}
uint64_t orc_color_exhaustive_fnv(void)
{
    uint64_t f = 0xcbf29ce484222325ULL;
    for (int y = -128; y < 128; y++) for (int cb = -128; cb < 128; cb++) for (int cr = -128; cr < 128; cr++) {
        PixCc p; p.pre_y = 8 * y; p.pre_cb = 8 * cb; p.pre_cr = 8 * cr; ycc_to_rgb_fast_float(&p);
        f ^= p.r; f *= 0x100000001b3ULL; f ^= p.g; f *= 0x100000001b3ULL; f ^= p.b; f *= 0x100000001b3ULL;
    }
    return f;
}
