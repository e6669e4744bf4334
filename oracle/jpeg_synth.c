/* jpeg_synth.c -- TEST / BENCH INPUT GENERATOR (not part of the product path).
 * See jpeg_synth.h.  Everything here follows ITU-T T.81 (public standard):
 * Annex A (FDCT, zig-zag), Annex B (marker syntax), Annex C (code generation),
 * Annex F (sequential Huffman coding), Annex G (progressive), Annex K (tables).
 */
#include "jpeg_synth.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ tables */
static const uint8_t ZZ[64] = {            /* zig-zag index -> natural index (T.81 Fig. A.6) */
     0, 1, 8,16, 9, 2, 3,10, 17,24,32,25,18,11, 4, 5,
    12,19,26,33,40,48,41,34, 27,20,13, 6, 7,14,21,28,
    35,42,49,56,57,50,43,36, 29,22,15,23,30,37,44,51,
    58,59,52,45,38,31,39,46, 53,60,61,54,47,55,62,63 };
static const uint8_t KQ_LUM[64] = {        /* T.81 Table K.1 (natural order) */
    16,11,10,16,24,40,51,61, 12,12,14,19,26,58,60,55, 14,13,16,24,40,57,69,56, 14,17,22,29,51,87,80,62,
    18,22,37,56,68,109,103,77, 24,35,55,64,81,104,113,92, 49,64,78,87,103,121,120,101, 72,92,95,98,112,100,103,99 };
static const uint8_t KQ_CHR[64] = {        /* T.81 Table K.2 */
    17,18,24,47,99,99,99,99, 18,21,26,66,99,99,99,99, 24,26,56,99,99,99,99,99, 47,66,99,99,99,99,99,99,
    99,99,99,99,99,99,99,99, 99,99,99,99,99,99,99,99, 99,99,99,99,99,99,99,99, 99,99,99,99,99,99,99,99 };
/* T.81 Tables K.3 - K.6: BITS[1..16] then HUFFVAL */
static const uint8_t K_DC_LUM_BITS[16] = {0,1,5,1,1,1,1,1,1,0,0,0,0,0,0,0};
static const uint8_t K_DC_LUM_VAL[12]  = {0,1,2,3,4,5,6,7,8,9,10,11};
static const uint8_t K_DC_CHR_BITS[16] = {0,3,1,1,1,1,1,1,1,1,1,0,0,0,0,0};
static const uint8_t K_DC_CHR_VAL[12]  = {0,1,2,3,4,5,6,7,8,9,10,11};
static const uint8_t K_AC_LUM_BITS[16] = {0,2,1,3,3,2,4,3,5,5,4,4,0,0,1,0x7d};
static const uint8_t K_AC_LUM_VAL[162] = {
    0x01,0x02,0x03,0x00,0x04,0x11,0x05,0x12,0x21,0x31,0x41,0x06,0x13,0x51,0x61,0x07,0x22,0x71,0x14,0x32,0x81,0x91,0xa1,0x08,
    0x23,0x42,0xb1,0xc1,0x15,0x52,0xd1,0xf0,0x24,0x33,0x62,0x72,0x82,0x09,0x0a,0x16,0x17,0x18,0x19,0x1a,0x25,0x26,0x27,0x28,
    0x29,0x2a,0x34,0x35,0x36,0x37,0x38,0x39,0x3a,0x43,0x44,0x45,0x46,0x47,0x48,0x49,0x4a,0x53,0x54,0x55,0x56,0x57,0x58,0x59,
    0x5a,0x63,0x64,0x65,0x66,0x67,0x68,0x69,0x6a,0x73,0x74,0x75,0x76,0x77,0x78,0x79,0x7a,0x83,0x84,0x85,0x86,0x87,0x88,0x89,
    0x8a,0x92,0x93,0x94,0x95,0x96,0x97,0x98,0x99,0x9a,0xa2,0xa3,0xa4,0xa5,0xa6,0xa7,0xa8,0xa9,0xaa,0xb2,0xb3,0xb4,0xb5,0xb6,
    0xb7,0xb8,0xb9,0xba,0xc2,0xc3,0xc4,0xc5,0xc6,0xc7,0xc8,0xc9,0xca,0xd2,0xd3,0xd4,0xd5,0xd6,0xd7,0xd8,0xd9,0xda,0xe1,0xe2,
    0xe3,0xe4,0xe5,0xe6,0xe7,0xe8,0xe9,0xea,0xf1,0xf2,0xf3,0xf4,0xf5,0xf6,0xf7,0xf8,0xf9,0xfa };
static const uint8_t K_AC_CHR_BITS[16] = {0,2,1,2,4,4,3,4,7,5,4,4,0,1,2,0x77};
static const uint8_t K_AC_CHR_VAL[162] = {
    0x00,0x01,0x02,0x03,0x11,0x04,0x05,0x21,0x31,0x06,0x12,0x41,0x51,0x07,0x61,0x71,0x13,0x22,0x32,0x81,0x08,0x14,0x42,0x91,
    0xa1,0xb1,0xc1,0x09,0x23,0x33,0x52,0xf0,0x15,0x62,0x72,0xd1,0x0a,0x16,0x24,0x34,0xe1,0x25,0xf1,0x17,0x18,0x19,0x1a,0x26,
    0x27,0x28,0x29,0x2a,0x35,0x36,0x37,0x38,0x39,0x3a,0x43,0x44,0x45,0x46,0x47,0x48,0x49,0x4a,0x53,0x54,0x55,0x56,0x57,0x58,
    0x59,0x5a,0x63,0x64,0x65,0x66,0x67,0x68,0x69,0x6a,0x73,0x74,0x75,0x76,0x77,0x78,0x79,0x7a,0x82,0x83,0x84,0x85,0x86,0x87,
    0x88,0x89,0x8a,0x92,0x93,0x94,0x95,0x96,0x97,0x98,0x99,0x9a,0xa2,0xa3,0xa4,0xa5,0xa6,0xa7,0xa8,0xa9,0xaa,0xb2,0xb3,0xb4,
    0xb5,0xb6,0xb7,0xb8,0xb9,0xba,0xc2,0xc3,0xc4,0xc5,0xc6,0xc7,0xc8,0xc9,0xca,0xd2,0xd3,0xd4,0xd5,0xd6,0xd7,0xd8,0xd9,0xda,
    0xe2,0xe3,0xe4,0xe5,0xe6,0xe7,0xe8,0xe9,0xea,0xf2,0xf3,0xf4,0xf5,0xf6,0xf7,0xf8,0xf9,0xfa };

typedef struct { uint8_t bits[16]; uint8_t val[256]; int nval; uint16_t code[256]; uint8_t size[256]; } HuffTab;

static void huff_derive(HuffTab* t)               /* T.81 Annex C: canonical codes */
{
    memset(t->code, 0, sizeof t->code); memset(t->size, 0, sizeof t->size);
    unsigned code = 0; int k = 0;
    for (int len = 1; len <= 16; len++) {
        for (int i = 0; i < t->bits[len - 1]; i++, k++) { t->code[t->val[k]] = (uint16_t)code++; t->size[t->val[k]] = (uint8_t)len; }
        code <<= 1;
    }
    t->nval = k;
}
static void huff_std(HuffTab* t, const uint8_t* bits, const uint8_t* val, int n)
{ memcpy(t->bits, bits, 16); memset(t->val, 0, sizeof t->val); memcpy(t->val, val, (size_t)n); huff_derive(t); }

/* T.81 K.2: optimal code lengths from symbol frequencies, limited to 16 bits,
 * with the reserved all-ones code point (pseudo-symbol 256). */
static void huff_optimal(HuffTab* t, const long* freq_in)
{
    long freq[257]; int codesize[257], others[257]; uint8_t bits[33];
    memcpy(freq, freq_in, 256 * sizeof(long)); freq[256] = 1;
    memset(codesize, 0, sizeof codesize); memset(bits, 0, sizeof bits);
    for (int i = 0; i < 257; i++) others[i] = -1;
    for (;;) {
        int c1 = -1, c2 = -1; long v = 1000000000L;
        for (int i = 0; i <= 256; i++) if (freq[i] && freq[i] <= v) { v = freq[i]; c1 = i; }
        v = 1000000000L;
        for (int i = 0; i <= 256; i++) if (freq[i] && freq[i] <= v && i != c1) { v = freq[i]; c2 = i; }
        if (c2 < 0) break;
        freq[c1] += freq[c2]; freq[c2] = 0;
        codesize[c1]++; while (others[c1] >= 0) { c1 = others[c1]; codesize[c1]++; }
        others[c1] = c2;
        codesize[c2]++; while (others[c2] >= 0) { c2 = others[c2]; codesize[c2]++; }
    }
    for (int i = 0; i <= 256; i++) if (codesize[i]) bits[codesize[i]]++;
    for (int i = 32; i > 16; i--) while (bits[i] > 0) {
        int j = i - 2; while (bits[j] == 0) j--;
        bits[i] -= 2; bits[i - 1]++; bits[j + 1] += 2; bits[j]--;
    }
    { int i = 16; while (bits[i] == 0) i--; bits[i]--; }     /* drop the reserved code point */
    memcpy(t->bits, bits + 1, 16);
    int k = 0; memset(t->val, 0, sizeof t->val);
    for (int len = 1; len <= 32; len++) for (int s = 0; s < 256; s++) if (codesize[s] == len) t->val[k++] = (uint8_t)s;
    huff_derive(t);
}

/* ------------------------------------------------------------- bit writer */
typedef struct { uint8_t* out; size_t cap, n; uint32_t acc; int nbits; } BitW;
static void put_byte(BitW* w, unsigned b) { if (w->n < w->cap) w->out[w->n] = (uint8_t)b; w->n++; }
static void put_u16(BitW* w, unsigned v) { put_byte(w, v >> 8); put_byte(w, v & 255); }
static void put_bits(BitW* w, unsigned v, int n)
{
    if (!n) return;
    w->acc = (w->acc << n) | (v & ((1u << n) - 1)); w->nbits += n;
    while (w->nbits >= 8) {
        unsigned b = (w->acc >> (w->nbits - 8)) & 255; put_byte(w, b); if (b == 255) put_byte(w, 0);
        w->nbits -= 8;
    }
}
static void flush_bits(BitW* w) { if (w->nbits) put_bits(w, (1u << (8 - w->nbits)) - 1, 8 - w->nbits); w->acc = 0; w->nbits = 0; }
static int  bit_size(int v) { int a = v < 0 ? -v : v, n = 0; while (a) { n++; a >>= 1; } return n; }
static unsigned bit_mag(int v, int n) { return (unsigned)(v < 0 ? v + (1 << n) - 1 : v); }

/* ---------------------------------------------------------- synthetic image */
static uint32_t xs32(uint32_t* s) { uint32_t x = *s; x ^= x << 13; x ^= x >> 17; x ^= x << 5; return *s = x; }
void jsynth_image_rgb(const JsynthParams* p, uint8_t* rgb)
{
    const int W = p->width, H = p->height;
    uint32_t s = p->seed * 2654435761u + 0x9e3779b9u; if (!s) s = 1;
    for (int i = 0; i < 8; i++) xs32(&s);
    double fx[3], fy[3], ph[3], amp[3], base[3];
    for (int c = 0; c < 3; c++) {
        fx[c] = (1.0 + (xs32(&s) % 700) / 100.0) * 6.283185307179586 / (W > 1 ? W : 1);
        fy[c] = (1.0 + (xs32(&s) % 700) / 100.0) * 6.283185307179586 / (H > 1 ? H : 1);
        ph[c] = (xs32(&s) % 6283) / 1000.0; amp[c] = 40.0 + xs32(&s) % 50; base[c] = 100.0 + xs32(&s) % 56;
    }
    double* sx = (double*)malloc(sizeof(double) * 3 * (size_t)W);
    double* cx = (double*)malloc(sizeof(double) * 3 * (size_t)W);
    for (int c = 0; c < 3; c++) for (int x = 0; x < W; x++) { sx[c * W + x] = sin(fx[c] * x + ph[c]); cx[c * W + x] = cos(fx[c] * x + ph[c]); }
    const double sig = p->noise_sigma;
    for (int y = 0; y < H; y++) {
        double sy[3], cy[3];
        for (int c = 0; c < 3; c++) { sy[c] = sin(fy[c] * y); cy[c] = cos(fy[c] * y); }
        uint8_t* row = rgb + (size_t)y * W * 3;
        for (int x = 0; x < W; x++) for (int c = 0; c < 3; c++) {
            /* sin(a+b) = sin a cos b + cos a sin b : smooth 2-D sinusoid */
            double v = base[c] + amp[c] * (sx[c * W + x] * cy[c] + cx[c * W + x] * sy[c]);
            if (sig > 0) {  /* sum of four uniforms ~ N(0,1) after scaling (Irwin-Hall) */
                uint32_t r = xs32(&s), q = xs32(&s);
                double u = ((r & 0xffff) + (r >> 16) + (q & 0xffff) + (q >> 16)) / 65536.0 - 2.0;
                v += sig * u * 1.7320508075688772;
            }
            int iv = (int)floor(v + 0.5); row[x * 3 + c] = (uint8_t)(iv < 0 ? 0 : iv > 255 ? 255 : iv);
        }
    }
    free(sx); free(cx);
}

/* ------------------------------------------------------------------ encoder */
typedef struct { int hs, vs, bw, bh, qsel; int16_t* coef; } Comp;   /* bw,bh = blocks incl. MCU padding */

static void fdct_quant(const double* px /*8x8, level shifted*/, const uint16_t* q /*natural*/, int16_t* out /*zig-zag*/)
{
    static double C[8][8]; static int init = 0;
    if (!init) { for (int u = 0; u < 8; u++) for (int x = 0; x < 8; x++)
        C[u][x] = (u ? 0.5 : 0.35355339059327373) * cos((2 * x + 1) * u * 3.141592653589793 / 16.0); init = 1; }
    double t[64], f[64];
    for (int y = 0; y < 8; y++) for (int u = 0; u < 8; u++) { double a = 0; for (int x = 0; x < 8; x++) a += C[u][x] * px[y * 8 + x]; t[y * 8 + u] = a; }
    for (int v = 0; v < 8; v++) for (int u = 0; u < 8; u++) { double a = 0; for (int y = 0; y < 8; y++) a += C[v][y] * t[y * 8 + u]; f[v * 8 + u] = a; }
    for (int k = 0; k < 64; k++) { double r = f[ZZ[k]] / q[ZZ[k]]; out[k] = (int16_t)(r < 0 ? -floor(-r + 0.5) : floor(r + 0.5)); }
}

static void emit_dqt(BitW* w, int id, const uint16_t* q)
{ put_u16(w, 0xFFDB); put_u16(w, 67); put_byte(w, (unsigned)id); for (int k = 0; k < 64; k++) put_byte(w, q[ZZ[k]]); }
static void emit_dht(BitW* w, int cls, int id, const HuffTab* t)
{ put_u16(w, 0xFFC4); put_u16(w, (unsigned)(19 + t->nval)); put_byte(w, (unsigned)(cls << 4 | id));
  for (int i = 0; i < 16; i++) put_byte(w, t->bits[i]); for (int i = 0; i < t->nval; i++) put_byte(w, t->val[i]); }

/* One sequential block (T.81 F.1.2): DC difference then AC run/size pairs.
 * With freq != NULL only gathers statistics. */
static void code_block_seq(BitW* w, const int16_t* zz, int* pred, const HuffTab* dc, const HuffTab* ac, long* fdc, long* fac)
{
    int d = zz[0] - *pred; *pred = zz[0];
    int n = bit_size(d);
    if (fdc) fdc[n]++; else { put_bits(w, dc->code[n], dc->size[n]); put_bits(w, bit_mag(d, n), n); }
    int run = 0;
    for (int k = 1; k < 64; k++) {
        int v = zz[k];
        if (!v) { run++; continue; }
        while (run > 15) { if (fac) fac[0xF0]++; else put_bits(w, ac->code[0xF0], ac->size[0xF0]); run -= 16; }
        n = bit_size(v); int sym = run << 4 | n;
        if (fac) fac[sym]++; else { put_bits(w, ac->code[sym], ac->size[sym]); put_bits(w, bit_mag(v, n), n); }
        run = 0;
    }
    if (run) { if (fac) fac[0]++; else put_bits(w, ac->code[0], ac->size[0]); }
}

/* Walks the interleaved MCU order of a sequential scan. pass 0 = statistics. */
static void scan_sequential(BitW* w, const JsynthParams* p, Comp* comp, int ncomp, int mcux, int mcuy,
                            HuffTab* dc, HuffTab* ac, long fdc[2][256], long fac[2][256], int stats)
{
    int pred[3] = {0, 0, 0}, left = p->restart_interval, rstn = 0;
    for (int my = 0; my < mcuy; my++) for (int mx = 0; mx < mcux; mx++) {
        if (p->restart_interval && left == 0) {
            if (!stats) { flush_bits(w); put_u16(w, 0xFFD0 + (rstn & 7)); }
            rstn++; pred[0] = pred[1] = pred[2] = 0; left = p->restart_interval;
        }
        for (int c = 0; c < ncomp; c++) {
            int t = c ? 1 : 0;
            for (int v = 0; v < comp[c].vs; v++) for (int h = 0; h < comp[c].hs; h++) {
                int bx = mx * comp[c].hs + h, by = my * comp[c].vs + v;
                const int16_t* zz = comp[c].coef + ((size_t)by * comp[c].bw + bx) * 64;
                code_block_seq(w, zz, &pred[c], &dc[t], &ac[t], stats ? fdc[t] : NULL, stats ? fac[t] : NULL);
            }
        }
        if (p->restart_interval) left--;
    }
    if (!stats) flush_bits(w);
}

/* Progressive (T.81 Annex G): first and refinement scans of the DC coefficients and of AC bands. */
static void prog_dc_scan(BitW* w, const JsynthParams* p, Comp* comp, int ncomp, int mcux, int mcuy, HuffTab* dc, long fdc[2][256], int stats, int ah, int al)
{
    int pred[3] = {0, 0, 0}, left = p->restart_interval, rstn = 0;
    for (int my = 0; my < mcuy; my++) for (int mx = 0; mx < mcux; mx++) {
        if (p->restart_interval && left == 0) {
            if (!stats) { flush_bits(w); put_u16(w, 0xFFD0 + (rstn & 7)); }
            rstn++; pred[0] = pred[1] = pred[2] = 0; left = p->restart_interval;
        }
        for (int c = 0; c < ncomp; c++) for (int v = 0; v < comp[c].vs; v++) for (int h = 0; h < comp[c].hs; h++) {
            int bx = mx * comp[c].hs + h, by = my * comp[c].vs + v, t = c ? 1 : 0;
            int z = comp[c].coef[((size_t)by * comp[c].bw + bx) * 64];
            if (ah) { if (!stats) put_bits(w, (unsigned)(z >> al) & 1u, 1); continue; }     /* G.1.2.1 refinement: one bit per block */
            z >>= al;                                                                      /* point transform: arithmetic shift */
            int d = z - pred[c]; pred[c] = z; int n = bit_size(d);
            if (stats) fdc[t][n]++; else { put_bits(w, dc[t].code[n], dc[t].size[n]); put_bits(w, bit_mag(d, n), n); }
        }
        if (p->restart_interval) left--;
    }
    if (!stats) flush_bits(w);
}
static void prog_flush_eobrun(BitW* w, int* eobrun, const HuffTab* ac, long* fac)
{
    if (!*eobrun) return;
    int n = 0, t = *eobrun; while (t > 1) { n++; t >>= 1; }
    if (fac) fac[n << 4]++; else { put_bits(w, ac->code[n << 4], ac->size[n << 4]); if (n) put_bits(w, (unsigned)*eobrun & ((1u << n) - 1), n); }
    *eobrun = 0;
}
/* Non-interleaved AC band scan of one component over its un-padded block grid
 * (T.81 A.2.3: ceil(X*Hi/Hmax / 8) x ceil(Y*Vi/Vmax / 8) blocks). */
static int pt_ac(int v, int al) { return v < 0 ? -((-v) >> al) : v >> al; }   /* AC point transform: divide by 2^Al toward zero */
static void prog_ac_scan(BitW* w, const JsynthParams* p, const Comp* c, int nbx, int nby, int ss, int se, const HuffTab* ac, long* fac, int al)
{
    int eobrun = 0, left = p->restart_interval, rstn = 0;
    for (int by = 0; by < nby; by++) for (int bx = 0; bx < nbx; bx++) {
        if (p->restart_interval && left == 0) {
            prog_flush_eobrun(w, &eobrun, ac, fac);
            if (!fac) { flush_bits(w); put_u16(w, 0xFFD0 + (rstn & 7)); }
            rstn++; left = p->restart_interval;
        }
        const int16_t* zz = c->coef + ((size_t)by * c->bw + bx) * 64;
        int run = 0;
        for (int k = ss; k <= se; k++) {
            int v = pt_ac(zz[k], al);
            if (!v) { run++; continue; }
            prog_flush_eobrun(w, &eobrun, ac, fac);
            while (run > 15) { if (fac) fac[0xF0]++; else put_bits(w, ac->code[0xF0], ac->size[0xF0]); run -= 16; }
            int n = bit_size(v), sym = run << 4 | n;
            if (fac) fac[sym]++; else { put_bits(w, ac->code[sym], ac->size[sym]); put_bits(w, bit_mag(v, n), n); }
            run = 0;
        }
        if (run) { if (++eobrun == 0x7FFF) prog_flush_eobrun(w, &eobrun, ac, fac); }
        if (p->restart_interval) left--;
    }
    prog_flush_eobrun(w, &eobrun, ac, fac);
    if (!fac) flush_bits(w);
}

/* AC refinement scan (T.81 G.1.2.3): coefficients that become non-zero at this bit position are coded as run/1 + sign,
 * the already non-zero ones contribute one correction bit each, buffered until the next code is written. */
typedef struct { unsigned char b[2048]; int n; } BitQ;
static void refine_flush(BitW* w, int* eobrun, BitQ* be, const HuffTab* ac, long* fac)
{
    if (*eobrun) {
        int n = 0, t = *eobrun; while (t > 1) { n++; t >>= 1; }
        if (fac) fac[n << 4]++; else { put_bits(w, ac->code[n << 4], ac->size[n << 4]); if (n) put_bits(w, (unsigned)*eobrun & ((1u << n) - 1), n); }
        *eobrun = 0;
    }
    if (!fac) for (int i = 0; i < be->n; i++) put_bits(w, be->b[i], 1);
    be->n = 0;
}
static void prog_ac_refine_scan(BitW* w, const JsynthParams* p, const Comp* c, int nbx, int nby, int ss, int se, const HuffTab* ac, long* fac, int al)
{
    int eobrun = 0, left = p->restart_interval, rstn = 0; BitQ be; be.n = 0;
    for (int by = 0; by < nby; by++) for (int bx = 0; bx < nbx; bx++) {
        if (p->restart_interval && left == 0) {
            refine_flush(w, &eobrun, &be, ac, fac);
            if (!fac) { flush_bits(w); put_u16(w, 0xFFD0 + (rstn & 7)); }
            rstn++; left = p->restart_interval;
        }
        const int16_t* zz = c->coef + ((size_t)by * c->bw + bx) * 64;
        int av[64], eob = 0;
        for (int k = ss; k <= se; k++) { int v = zz[k]; av[k] = (v < 0 ? -v : v) >> al; if (av[k] == 1) eob = k; }
        int run = 0; BitQ br; br.n = 0;
        for (int k = ss; k <= se; k++) {
            if (av[k] == 0) { run++; continue; }
            while (run > 15 && k <= eob) {                       /* ZRL only while a new coefficient is still to come */
                refine_flush(w, &eobrun, &be, ac, fac);
                if (fac) fac[0xF0]++; else put_bits(w, ac->code[0xF0], ac->size[0xF0]);
                run -= 16;
                if (!fac) for (int i = 0; i < br.n; i++) put_bits(w, br.b[i], 1);
                br.n = 0;
            }
            if (av[k] > 1) { br.b[br.n++] = (unsigned char)(av[k] & 1); continue; }   /* correction bit of a known coefficient */
            refine_flush(w, &eobrun, &be, ac, fac);
            if (fac) fac[(run << 4) + 1]++; else { put_bits(w, ac->code[(run << 4) + 1], ac->size[(run << 4) + 1]); put_bits(w, zz[k] < 0 ? 0u : 1u, 1); }
            if (!fac) for (int i = 0; i < br.n; i++) put_bits(w, br.b[i], 1);
            br.n = 0; run = 0;
        }
        if (run > 0 || br.n > 0) {                                 /* the rest of the block rides on an end-of-band run */
            eobrun++;
            for (int i = 0; i < br.n; i++) be.b[be.n++] = br.b[i];
            if (eobrun == 0x7FFF || be.n > 1900) refine_flush(w, &eobrun, &be, ac, fac);
        }
        if (p->restart_interval) left--;
    }
    refine_flush(w, &eobrun, &be, ac, fac);
    if (!fac) flush_bits(w);
}

size_t jsynth_encode_rgb(const JsynthParams* p, const uint8_t* rgb, uint8_t* out, size_t cap)
{
    const int W = p->width, H = p->height, ncomp = p->gray ? 1 : 3;
    const int hmax = p->gray ? 1 : p->hs, vmax = p->gray ? 1 : p->vs;
    const int mcuw = 8 * hmax, mcuh = 8 * vmax, mcux = (W + mcuw - 1) / mcuw, mcuy = (H + mcuh - 1) / mcuh;
    const int PW = mcux * mcuw, PH = mcuy * mcuh;

    /* quality scaling as in the IJG library: scale = q<50 ? 5000/q : 200-2q */
    int q = p->quality < 1 ? 1 : p->quality > 100 ? 100 : p->quality;
    int scale = q < 50 ? 5000 / q : 200 - 2 * q;
    uint16_t qt[2][64];
    for (int k = 0; k < 64; k++) {
        int a = (KQ_LUM[k] * scale + 50) / 100, b = (KQ_CHR[k] * scale + 50) / 100;
        qt[0][k] = (uint16_t)(a < 1 ? 1 : a > 255 ? 255 : a); qt[1][k] = (uint16_t)(b < 1 ? 1 : b > 255 ? 255 : b);
    }

    /* colour transform (JFIF) into MCU-padded planes, edge replication */
    float* pl[3] = {0, 0, 0};
    for (int c = 0; c < ncomp; c++) pl[c] = (float*)malloc(sizeof(float) * (size_t)PW * PH);
    for (int y = 0; y < PH; y++) { int sy = y < H ? y : H - 1;
        for (int x = 0; x < PW; x++) { int sx = x < W ? x : W - 1;
            const uint8_t* s = rgb + ((size_t)sy * W + sx) * 3; double r = s[0], g = s[1], b = s[2];
            pl[0][(size_t)y * PW + x] = (float)(0.299 * r + 0.587 * g + 0.114 * b);
            if (ncomp == 3) { pl[1][(size_t)y * PW + x] = (float)(-0.168735892 * r - 0.331264108 * g + 0.5 * b + 128.0);
                              pl[2][(size_t)y * PW + x] = (float)(0.5 * r - 0.418687589 * g - 0.081312411 * b + 128.0); }
        } }

    Comp comp[3]; memset(comp, 0, sizeof comp);
    for (int c = 0; c < ncomp; c++) {
        comp[c].hs = c ? 1 : hmax; comp[c].vs = c ? 1 : vmax; comp[c].qsel = c ? 1 : 0;
        comp[c].bw = mcux * comp[c].hs; comp[c].bh = mcuy * comp[c].vs;
        comp[c].coef = (int16_t*)malloc(sizeof(int16_t) * 64 * (size_t)comp[c].bw * comp[c].bh);
        const int ex = hmax / comp[c].hs, ey = vmax / comp[c].vs;     /* box-filter down-sampling */
        for (int by = 0; by < comp[c].bh; by++) for (int bx = 0; bx < comp[c].bw; bx++) {
            double px[64];
            for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) {
                double a = 0; for (int j = 0; j < ey; j++) for (int i = 0; i < ex; i++)
                    a += pl[c][(size_t)((by * 8 + y) * ey + j) * PW + (bx * 8 + x) * ex + i];
                px[y * 8 + x] = floor(a / (ex * ey) + 0.5) - 128.0;
            }
            fdct_quant(px, qt[comp[c].qsel], comp[c].coef + ((size_t)by * comp[c].bw + bx) * 64);
        }
    }
    for (int c = 0; c < ncomp; c++) free(pl[c]);

    HuffTab dc[2], ac[2];
    huff_std(&dc[0], K_DC_LUM_BITS, K_DC_LUM_VAL, 12); huff_std(&dc[1], K_DC_CHR_BITS, K_DC_CHR_VAL, 12);
    huff_std(&ac[0], K_AC_LUM_BITS, K_AC_LUM_VAL, 162); huff_std(&ac[1], K_AC_CHR_BITS, K_AC_CHR_VAL, 162);

    BitW w; w.out = out; w.cap = cap; w.n = 0; w.acc = 0; w.nbits = 0;
    static const uint8_t app0[] = {0xFF,0xE0,0,16,'J','F','I','F',0,1,1,0,0,1,0,1,0,0};
    put_u16(&w, 0xFFD8); for (size_t i = 0; i < sizeof app0; i++) put_byte(&w, app0[i]);
    emit_dqt(&w, 0, qt[0]); if (ncomp == 3) emit_dqt(&w, 1, qt[1]);
    put_u16(&w, p->progressive ? 0xFFC2 : 0xFFC0); put_u16(&w, (unsigned)(8 + 3 * ncomp)); put_byte(&w, 8);
    put_u16(&w, (unsigned)H); put_u16(&w, (unsigned)W); put_byte(&w, (unsigned)ncomp);
    for (int c = 0; c < ncomp; c++) { put_byte(&w, (unsigned)(c + 1)); put_byte(&w, (unsigned)(comp[c].hs << 4 | comp[c].vs)); put_byte(&w, (unsigned)comp[c].qsel); }
    if (p->restart_interval) { put_u16(&w, 0xFFDD); put_u16(&w, 4); put_u16(&w, (unsigned)p->restart_interval); }

    if (!p->progressive) {
        if (p->optimize_huffman) {
            long fdc[2][256], fac[2][256]; memset(fdc, 0, sizeof fdc); memset(fac, 0, sizeof fac);
            scan_sequential(&w, p, comp, ncomp, mcux, mcuy, dc, ac, fdc, fac, 1);
            for (int t = 0; t < (ncomp == 3 ? 2 : 1); t++) { huff_optimal(&dc[t], fdc[t]); huff_optimal(&ac[t], fac[t]); }
        }
        for (int t = 0; t < (ncomp == 3 ? 2 : 1); t++) { emit_dht(&w, 0, t, &dc[t]); emit_dht(&w, 1, t, &ac[t]); }
        put_u16(&w, 0xFFDA); put_u16(&w, (unsigned)(6 + 2 * ncomp)); put_byte(&w, (unsigned)ncomp);
        for (int c = 0; c < ncomp; c++) { put_byte(&w, (unsigned)(c + 1)); put_byte(&w, c ? 0x11 : 0x00); }
        put_byte(&w, 0); put_byte(&w, 63); put_byte(&w, 0);
        scan_sequential(&w, p, comp, ncomp, mcux, mcuy, dc, ac, NULL, NULL, 0);
    } else {
        /* progressive == 1: spectral selection only (DC, then two AC bands per component);
         * progressive == 2: spectral selection + successive approximation, a script in the style of the IJG default */
        typedef struct { int comp, ss, se, ah, al; } ScanDef;     /* comp < 0: all components interleaved (DC scans only) */
        ScanDef script[16]; int ns = 0;
        if (p->progressive == 1) {
            script[ns++] = (ScanDef){-1, 0, 0, 0, 0};
            for (int c = 0; c < ncomp; c++) { script[ns++] = (ScanDef){c, 1, 5, 0, 0}; script[ns++] = (ScanDef){c, 6, 63, 0, 0}; }
        } else {
            script[ns++] = (ScanDef){-1, 0, 0, 0, 1};
            script[ns++] = (ScanDef){0, 1, 5, 0, 2};
            if (ncomp == 3) { script[ns++] = (ScanDef){2, 1, 63, 0, 1}; script[ns++] = (ScanDef){1, 1, 63, 0, 1}; }
            script[ns++] = (ScanDef){0, 6, 63, 0, 2};
            script[ns++] = (ScanDef){0, 1, 63, 2, 1};
            script[ns++] = (ScanDef){-1, 0, 0, 1, 0};
            if (ncomp == 3) { script[ns++] = (ScanDef){2, 1, 63, 1, 0}; script[ns++] = (ScanDef){1, 1, 63, 1, 0}; }
            script[ns++] = (ScanDef){0, 1, 63, 1, 0};
        }
        for (int si = 0; si < ns; si++) {
            const ScanDef sd = script[si];
            if (sd.comp < 0) {                                     /* DC scan, interleaved */
                if (!sd.ah) {
                    if (p->optimize_huffman) { long fdc[2][256]; memset(fdc, 0, sizeof fdc);
                        prog_dc_scan(&w, p, comp, ncomp, mcux, mcuy, dc, fdc, 1, 0, sd.al);
                        for (int t = 0; t < (ncomp == 3 ? 2 : 1); t++) huff_optimal(&dc[t], fdc[t]); }
                    for (int t = 0; t < (ncomp == 3 ? 2 : 1); t++) emit_dht(&w, 0, t, &dc[t]);
                }
                put_u16(&w, 0xFFDA); put_u16(&w, (unsigned)(6 + 2 * ncomp)); put_byte(&w, (unsigned)ncomp);
                for (int c = 0; c < ncomp; c++) { put_byte(&w, (unsigned)(c + 1)); put_byte(&w, c ? 0x10 : 0x00); }
                put_byte(&w, 0); put_byte(&w, 0); put_byte(&w, (unsigned)(sd.ah << 4 | sd.al));
                prog_dc_scan(&w, p, comp, ncomp, mcux, mcuy, dc, NULL, 0, sd.ah, sd.al);
                continue;
            }
            const int c = sd.comp, t = c ? 1 : 0;
            int nbx = ncomp == 1 ? (W + 7) / 8 : ((W * comp[c].hs + hmax - 1) / hmax + 7) / 8;
            int nby = ncomp == 1 ? (H + 7) / 8 : ((H * comp[c].vs + vmax - 1) / vmax + 7) / 8;
            HuffTab a = ac[t];
            /* EOBn symbols (0x10..0xE0) are absent from Annex K: AC scan tables are always per-scan optimal */
            { long fac[256]; memset(fac, 0, sizeof fac);
                if (sd.ah) prog_ac_refine_scan(&w, p, &comp[c], nbx, nby, sd.ss, sd.se, &a, fac, sd.al);
                else prog_ac_scan(&w, p, &comp[c], nbx, nby, sd.ss, sd.se, &a, fac, sd.al);
                huff_optimal(&a, fac); }
            emit_dht(&w, 1, t, &a);
            put_u16(&w, 0xFFDA); put_u16(&w, 8); put_byte(&w, 1); put_byte(&w, (unsigned)(c + 1)); put_byte(&w, (unsigned)t);
            put_byte(&w, (unsigned)sd.ss); put_byte(&w, (unsigned)sd.se); put_byte(&w, (unsigned)(sd.ah << 4 | sd.al));
            if (sd.ah) prog_ac_refine_scan(&w, p, &comp[c], nbx, nby, sd.ss, sd.se, &a, NULL, sd.al);
            else prog_ac_scan(&w, p, &comp[c], nbx, nby, sd.ss, sd.se, &a, NULL, sd.al);
        }
    }
    put_u16(&w, 0xFFD9);
    for (int c = 0; c < ncomp; c++) free(comp[c].coef);
    return w.n;
}

size_t jsynth_encode(const JsynthParams* p, uint8_t* out, size_t cap)
{
    uint8_t* rgb = (uint8_t*)malloc((size_t)p->width * p->height * 3);
    jsynth_image_rgb(p, rgb);
    size_t n = jsynth_encode_rgb(p, rgb, out, cap);
    free(rgb);
    return n;
}
