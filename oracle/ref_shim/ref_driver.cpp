// ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A flat C interface over the UNMODIFIED reference classes (CimgDecode,
// CwindowBuf, CDocLog compiled in place from /root/reference/source), with the
// same call sequence CjfifDecode drives them with (reference
// source/JfifDecode.cpp:3581,3600,4648,5008-5025,5161,5291,5299).  The entry
// points mirror include/jsnoop_gpu.h one-for-one so that the parity tests can run
// the same script against (a) this library, (b) oracle/oracle_imgdecode.c and
// (c) the HIP path.  `#define private public` exposes the decoder's internal maps
// and per-function probes (IDCT, colour) for known-answer tests only.
#include "stdafx.h"
#include <cstring>
#define private public
#include "ImgDecode.h"
#undef private
#include "JPEGsnoop.h"
#include "FileTiff.h"
#include "General.h"

std::vector<std::string>& ShimLog();
CSnoopConfig* ShimConfig();

struct JsRef {
    CDocLog     log;
    CwindowBuf  wbuf;
    CimgDecode* dec;
    CFile*      file;
    std::vector<BYTE> bytes;
    JsRef() : dec(nullptr), file(nullptr) { (void)AfxGetApp(); dec = new CimgDecode(&log, &wbuf); }
    ~JsRef() { delete dec; wbuf.BufFileUnset(); delete file; }
};

extern "C" {

void* jsref_create(void) { return new JsRef(); }
void  jsref_destroy(void* h) { delete (JsRef*)h; }
void  jsref_reset_state(void* h) { ((JsRef*)h)->dec->ResetState(); }
void  jsref_reset(void* h) { ((JsRef*)h)->dec->Reset(); }

void jsref_set_options(int decode_ac, int histo_en, int stat_clip_en, unsigned err_max)
{
    CSnoopConfig* c = ShimConfig();
    c->bDecodeScanImgAc = decode_ac != 0;
    c->bHistoEn = histo_en != 0;
    c->bStatClipEn = stat_clip_en != 0;
    c->nErrMaxDecodeScan = err_max;
}

void jsref_set_dump_histo_y(int on) { ShimConfig()->bDumpHistoY = on != 0; }      // CSnoopConfig::bDumpHistoY, read at ImgDecode.cpp:2730

int jsref_set_dqt_entry(void* h, unsigned tbl, unsigned nat, unsigned zz, unsigned val)
{ return ((JsRef*)h)->dec->SetDqtEntry(tbl, nat, zz, (unsigned short)val); }
int jsref_set_dqt_tables(void* h, unsigned comp, unsigned tbl)
{ return ((JsRef*)h)->dec->SetDqtTables(comp, tbl); }
unsigned jsref_get_dqt_entry(void* h, unsigned tbl, unsigned nat)
{ return ((JsRef*)h)->dec->GetDqtEntry(tbl, nat); }
int jsref_set_dht_entry(void* h, unsigned dest, unsigned cls, unsigned ind, unsigned len,
                        unsigned bits, unsigned mask, unsigned code)
{ return ((JsRef*)h)->dec->SetDhtEntry(dest, cls, ind, len, bits, mask, code); }
int jsref_set_dht_size(void* h, unsigned dest, unsigned cls, unsigned n)
{ return ((JsRef*)h)->dec->SetDhtSize(dest, cls, n); }
int jsref_set_dht_tables(void* h, unsigned comp, unsigned dc, unsigned ac)
{ return ((JsRef*)h)->dec->SetDhtTables(comp, dc, ac); }
void jsref_set_sof_samp_factors(void* h, unsigned comp, unsigned sh, unsigned sv)
{ ((JsRef*)h)->dec->SetSofSampFactors(comp, sh, sv); }
void jsref_set_precision(void* h, unsigned p) { ((JsRef*)h)->dec->SetPrecision(p); }
void jsref_set_image_details(void* h, unsigned x, unsigned y, unsigned nf, unsigned ns,
                             int rst_en, unsigned rst_interval)
{ ((JsRef*)h)->dec->SetImageDetails(x, y, nf, ns, rst_en != 0, rst_interval); }

// Hands the whole file to the reference's own sliding-window reader through a
// memory-backed CFile, then runs the scan decode from byte offset `start`.
void jsref_decode_scan_img(void* hv, const unsigned char* file, size_t len, unsigned start,
                           int display, int quiet)
{
    JsRef* h = (JsRef*)hv;
    ShimLog().clear();
    h->wbuf.BufFileUnset();
    delete h->file;
    h->bytes.assign(file, file + len);
    h->file = new CFile(h->bytes.data(), h->bytes.size());
    h->wbuf.BufFileSet(h->file);
    h->wbuf.BufLoadWindow(0);
    h->dec->DecodeScanImg(start, display != 0, quiet != 0);
}

int  jsref_is_preview_ready(void* h) { return ((JsRef*)h)->dec->IsPreviewReady(); }
void jsref_get_image_size(void* h, unsigned* x, unsigned* y)
{ unsigned a = 0, b = 0; ((JsRef*)h)->dec->GetImageSize(a, b); *x = a; *y = b; }
const unsigned char* jsref_get_bitmap_ptr(void* h)
{ unsigned char* p = nullptr; ((JsRef*)h)->dec->GetBitmapPtr(p); return p; }
void jsref_get_pixmap_ptrs(void* h, const short** y, const short** cb, const short** cr)
{ CimgDecode* d = ((JsRef*)h)->dec; *y = d->m_pPixValY; *cb = d->m_pPixValCb; *cr = d->m_pPixValCr; }
void jsref_lookup_file_pos_mcu(void* h, unsigned mx, unsigned my, unsigned* byte, unsigned* bit)
{ unsigned a = 0, b = 0; ((JsRef*)h)->dec->LookupFilePosMcu(mx, my, a, b); *byte = a; *bit = b; }
void jsref_lookup_blk_ycc(void* h, unsigned bx, unsigned by, int* y, int* cb, int* cr)
{ int a = 0, b = 0, c = 0; ((JsRef*)h)->dec->LookupBlkYCC(bx, by, a, b, c); *y = a; *cb = b; *cr = c; }

// ---- internals, for side-output parity and KATs -----------------------------
void jsref_get_geometry(void* h, unsigned* out8)
{
    CimgDecode* d = ((JsRef*)h)->dec;
    out8[0] = d->m_nMcuWidth; out8[1] = d->m_nMcuHeight; out8[2] = d->m_nMcuXMax; out8[3] = d->m_nMcuYMax;
    out8[4] = d->m_nBlkXMax;  out8[5] = d->m_nBlkYMax;   out8[6] = d->m_nImgSizeX; out8[7] = d->m_nImgSizeY;
}
const unsigned* jsref_mcu_file_map(void* h) { return ((JsRef*)h)->dec->m_pMcuFileMap; }
void jsref_blk_dc_ptrs(void* h, const short** y, const short** cb, const short** cr)
{ CimgDecode* d = ((JsRef*)h)->dec; *y = d->m_pBlkDcValY; *cb = d->m_pBlkDcValCb; *cr = d->m_pBlkDcValCr; }
const unsigned* jsref_dht_histo(void* h) { return &((JsRef*)h)->dec->m_anDhtHisto[0][0][0]; }  // [2][4][17]
void jsref_scan_status(void* h, unsigned* out8)
{
    CimgDecode* d = ((JsRef*)h)->dec;
    out8[0] = d->m_bScanBad; out8[1] = d->m_bScanEnd; out8[2] = d->m_nRestartRead; out8[3] = d->m_nNumPixels;
    out8[4] = d->m_anScanBuffPtr_pos[0]; out8[5] = d->m_nScanBuffPtr_align; out8[6] = d->m_nWarnBadScanNum;
    out8[7] = d->m_nScanBuffPtr_first;
}
void jsref_bright_avg(void* h, int* out10)
{
    CimgDecode* d = ((JsRef*)h)->dec;
    out10[0] = d->m_bBrightValid; out10[1] = d->m_nBrightY; out10[2] = d->m_nBrightCb; out10[3] = d->m_nBrightCr;
    out10[4] = (int)d->m_nBrightR; out10[5] = (int)d->m_nBrightG; out10[6] = (int)d->m_nBrightB;
    out10[7] = d->m_ptBrightMcu.x; out10[8] = d->m_ptBrightMcu.y; out10[9] = (int)d->m_nAvgY;
}
void jsref_set_preview_mode(void* h, unsigned mode) { ((JsRef*)h)->dec->SetPreviewMode(mode); }                       // :633
unsigned jsref_get_preview_mode(void* h) { return ((JsRef*)h)->dec->GetPreviewMode(); }
void jsref_set_preview_ycc_offset(void* h, unsigned mx, unsigned my, int y, int cb, int cr)                                // :650
{ ((JsRef*)h)->dec->SetPreviewYccOffset(mx, my, y, cb, cr); }
// TIFF export: the container is written by the reference's own FileTiff::WriteFile (source/FileTiff.cpp:436); the pixel
// re-arrangement in front of it lives in a GUI handler (CJPEGsnoopDoc::OnToolsExporttiff, source/JPEGsnoopDoc.cpp:2110-2180:
// dialogs + loop) and is restated here.  mode 0 = RGB 8 bit, 1 = RGB 16 bit, 2 = YCC 8 bit.  Returns 0 on success.
int jsref_export_tiff(void* h, const char* path, int mode)
{
    CimgDecode* d = ((JsRef*)h)->dec;
    unsigned nSizeX = 0, nSizeY = 0; unsigned char* pRgb = NULL; short *pY = NULL, *pCb = NULL, *pCr = NULL;
    d->GetImageSize(nSizeX, nSizeY); d->GetBitmapPtr(pRgb); d->GetPixMapPtrs(pY, pCb, pCr);
    const bool bModeYcc = mode == 2, bMode16b = mode == 1;
    if (!pRgb || !nSizeX || !nSizeY || (bModeYcc && !(pY && pCb && pCr))) return -1;
    std::vector<unsigned char> sel8; std::vector<unsigned short> sel16;
    if (bMode16b) sel16.resize((size_t)nSizeX * nSizeY * 3); else sel8.resize((size_t)nSizeX * nSizeY * 3);
    for (unsigned y = 0; y < nSizeY; y++) for (unsigned x = 0; x < nSizeX; x++) {
        const size_t dst = ((size_t)y * nSizeX + x) * 3;
        if (!bModeYcc) {
            const size_t src = ((size_t)(nSizeY - 1 - y) * nSizeX + x) * 4;            // the DIB is bottom-up
            const unsigned short r = pRgb[src + 2], g = pRgb[src + 1], b = pRgb[src + 0];
            if (!bMode16b) { sel8[dst] = r & 0xFF; sel8[dst + 1] = g & 0xFF; sel8[dst + 2] = b & 0xFF; }
            else { sel16[dst] = Swap16(r << 8); sel16[dst + 1] = Swap16(g << 8); sel16[dst + 2] = Swap16(b << 8); }
        } else {
            const size_t src = (size_t)y * nSizeX + x;
            short vy = pY[src], vcb = pCb[src], vcr = pCr[src];
            if (vy < -1024) vy = -1024; if (vy > 1023) vy = 1023;
            if (vcb < -1024) vcb = -1024; if (vcb > 1023) vcb = 1023;
            if (vcr < -1024) vcr = -1024; if (vcr > 1023) vcr = 1023;
            sel8[dst] = (unsigned char)((0x0400 + vy) >> 3); sel8[dst + 1] = (unsigned char)((0x0400 + vcb) >> 3); sel8[dst + 2] = (unsigned char)((0x0400 + vcr) >> 3);
        }
    }
    FileTiff t;
    t.WriteFile(CString(path), bModeYcc, bMode16b, bMode16b ? (void*)sel16.data() : (void*)sel8.data(), nSizeX, nSizeY);
    return 0;
}
// Colour statistics of the bHistoEn / bStatClipEn path (ConvertYCCtoRGB :4229, CapYccRange :4341, CapRgbRange :4495):
// out[0..36] PixelCcHisto, [37..49] PixelCcClip, [50..433] m_anCcHisto_r/g/b[128], [434..2481] m_anHistoYFull[2048]
void jsref_color_stats(void* h, unsigned* out2482)
{
    CimgDecode* d = ((JsRef*)h)->dec;
    memcpy(out2482, &d->m_sHisto, 37 * 4); memcpy(out2482 + 37, &d->m_sStatClip, 13 * 4);
    memcpy(out2482 + 50, d->m_anCcHisto_r, 128 * 4); memcpy(out2482 + 178, d->m_anCcHisto_g, 128 * 4); memcpy(out2482 + 306, d->m_anCcHisto_b, 128 * 4);
    memcpy(out2482 + 434, d->m_anHistoYFull, 2048 * 4);
}
const float* jsref_idct_lut(void* h) { return &((JsRef*)h)->dec->m_afIdctLookup[0][0]; }       // [64][64]
const unsigned* jsref_dht_lookupfast(void* h) { return &((JsRef*)h)->dec->m_anDhtLookupfast[0][0][0]; } // [2][4][1024]

// Runs the reference's own IDCT on one natural-order coefficient block.
void jsref_idct_block(void* h, const short* coef64, float* out64)
{
    CimgDecode* d = ((JsRef*)h)->dec;
    for (int i = 0; i < 64; i++) d->m_anDctBlock[i] = coef64[i];
    d->DecodeIdctCalcFloat(64);
    for (int i = 0; i < 64; i++) out64[i] = d->m_afIdctBlock[i];
}
// Runs the reference's own fast-float colour conversion on n pre-range triples.
void jsref_color_fast(void* h, const int* ycc, unsigned char* rgb, size_t n)
{
    CimgDecode* d = ((JsRef*)h)->dec;
    for (size_t i = 0; i < n; i++) {
        PixelCc px; memset(&px, 0, sizeof px);
        px.nPrerangeY = ycc[3 * i]; px.nPrerangeCb = ycc[3 * i + 1]; px.nPrerangeCr = ycc[3 * i + 2];
        d->ConvertYCCtoRGBFastFloat(px);
        rgb[3 * i] = px.nFinalR; rgb[3 * i + 1] = px.nFinalG; rgb[3 * i + 2] = px.nFinalB;
    }
}
// FNV-1a-64 over all 2^24 (Y,Cb,Cr) triples, R,G,B byte order.
unsigned long long jsref_color_exhaustive_fnv(void* h)
{
    CimgDecode* d = ((JsRef*)h)->dec;
    unsigned long long f = 0xcbf29ce484222325ULL;
    for (int y = -128; y < 128; y++) for (int cb = -128; cb < 128; cb++) for (int cr = -128; cr < 128; cr++) {
        PixelCc px; memset(&px, 0, sizeof px);
        px.nPrerangeY = 8 * y; px.nPrerangeCb = 8 * cb; px.nPrerangeCr = 8 * cr;
        d->ConvertYCCtoRGBFastFloat(px);
        f ^= px.nFinalR; f *= 0x100000001b3ULL; f ^= px.nFinalG; f *= 0x100000001b3ULL; f ^= px.nFinalB; f *= 0x100000001b3ULL;
    }
    return f;
}

size_t jsref_log_count(void) { return ShimLog().size(); }
const char* jsref_log_line(size_t i) { return i < ShimLog().size() ? ShimLog()[i].c_str() : ""; }

} // extern "C"
