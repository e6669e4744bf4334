// jfif_front.cpp -- minimal JFIF header front end.
//
// The subset of CjfifDecode::DecodeMarker (reference source/JfifDecode.cpp:3759) that feeds
// CimgDecode: DQT (:4576-4650), SOF0/SOF1 (:4802-5039), DHT (:3401-3612), DRI (:5310-5324) and the
// first SOS (:5105-5182, :5291), issuing the same setter calls with the same arguments.  Everything
// else (EXIF, makernotes, signatures ...) is out of scope and skipped by segment length.
#include <string.h>
#include "jsnoop_host.h"

static const uint8_t kZigZag[64] = {
     0, 1, 8,16, 9, 2, 3,10, 17,24,32,25,18,11, 4, 5, 12,19,26,33,40,48,41,34, 27,20,13, 6, 7,14,21,28,
    35,42,49,56,57,50,43,36, 29,22,15,23,30,37,44,51, 58,59,52,45,38,31,39,46, 53,60,61,54,47,55,62,63 };

int js_jfif_walk(JsnoopDecoder* d, const uint8_t* f, size_t n, unsigned* scan_start)
{
    auto B = [&](size_t i) -> unsigned { return i < n ? f[i] : 0u; };      // CwindowBuf::Buf: 0 past EOF
    uint8_t unzz[64]; for (int k = 0; k < 64; k++) unzz[kZigZag[k]] = (uint8_t)k;
    jsnoop_reset_state(d);
    if (n < 4 || f[0] != 0xFF || f[1] != 0xD8) { js_set_error("not a JPEG stream (no SOI)"); return -1; }
    size_t pos = 2;
    unsigned nf = 0, sof_x = 0, sof_y = 0; bool have_sof = false; int rst_en = 0; unsigned rst_interval = 0;
    while (pos + 4 <= n) {
        if (f[pos] != 0xFF) { pos++; continue; }
        while (pos < n && f[pos] == 0xFF) pos++;                             // marker padding (:3777-3790)
        const unsigned m = B(pos++);
        if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (m == 0xD9) break;
        const unsigned len = B(pos) * 256 + B(pos + 1);
        const size_t seg = pos + 2, end = pos + len;
        if (len < 2 || end > n) { js_set_error("truncated marker segment 0xFF%02X", m); return -1; }
        if (m == 0xDB) {                                                    // DQT
            size_t p = seg;
            while (p < end) {
                const unsigned pq = B(p) >> 4, tq = B(p) & 15; p++;
                unsigned tbl[64];
                for (int k = 0; k < 64; k++) { unsigned v = B(p++); if (pq) v = (v << 8) + B(p++); tbl[kZigZag[k]] = v; }
                for (unsigned nat = 0; nat < 64; nat++) if (!jsnoop_set_dqt_entry(d, tq, nat, unzz[nat], tbl[nat])) { js_set_error("DQT destination out of range"); return -1; }
            }
        } else if (m == 0xC0 || m == 0xC1) {                                // SOF0 / SOF1
            const unsigned prec = B(seg); sof_y = B(seg + 1) * 256 + B(seg + 2); sof_x = B(seg + 3) * 256 + B(seg + 4); nf = B(seg + 5);
            for (unsigned c = 1; c <= nf; c++) {
                const unsigned tq = B(seg + 6 + 3 * (c - 1) + 2);
                if (!jsnoop_set_dqt_tables(d, c, tq)) { js_set_error("SOF table selector out of range"); return -1; }
                jsnoop_set_precision(d, prec);
            }
            for (unsigned c = 1; c <= nf; c++) { const unsigned hv = B(seg + 6 + 3 * (c - 1) + 1); jsnoop_set_sof_samp_factors(d, c, hv >> 4, hv & 15); }
            have_sof = true;
        } else if (m == 0xC2 || m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
            // the reference refuses every other SOF mode (m_bImgSofUnsupported, :4827-4833, :5272-5274)
            js_set_error("SOF mode 0xFF%02X is not supported by the scan decoder", m); return -1;
        } else if (m == 0xC4) {                                             // DHT
            size_t p = seg;
            while (p < end) {
                const unsigned tc = B(p) >> 4, th = B(p) & 15; p++;
                if (tc >= 2 || th >= 4) { js_set_error("DHT class/destination out of range"); return -1; }
                unsigned counts[17]; for (int i = 1; i <= 16; i++) counts[i] = B(p++);
                unsigned code = 0, ind = 0;
                for (unsigned bl = 1; bl <= 16; bl++) {
                    for (unsigned i = 0; i < counts[bl]; i++) {
                        const unsigned mask = ((1u << bl) - 1) << (32 - bl);
                        if (!jsnoop_set_dht_entry(d, th, tc, ind, bl, code << (32 - bl), mask, B(p++))) { js_set_error("too many DHT codes"); return -1; }
                        ind++; code++;
                    }
                    code <<= 1;
                }
                if (!jsnoop_set_dht_size(d, th, tc, ind)) { js_set_error("DHT size out of range"); return -1; }
            }
        } else if (m == 0xDD) {                                             // DRI
            rst_interval = B(seg) * 256 + B(seg + 1); rst_en = rst_interval != 0;
        } else if (m == 0xDA) {                                             // SOS (first one only, ImgDecode.h:23)
            if (!have_sof) { js_set_error("SOS before valid SOF defined"); return -1; }
            const unsigned ns = B(seg);
            if (ns > 4) { js_set_error("Scan decode does not support > 4 components"); return -1; }
            for (unsigned c = 1; c <= ns; c++) { const unsigned tt = B(seg + 1 + 2 * (c - 1) + 1); if (!jsnoop_set_dht_tables(d, c, tt >> 4, tt & 15)) { js_set_error("SOS table selector out of range"); return -1; } }
            jsnoop_set_image_details(d, sof_x, sof_y, nf, ns, rst_en, rst_interval);
            *scan_start = (unsigned)end;
            return 0;
        }
        pos = end;
    }
    js_set_error("no SOS marker found");
    return -1;
}
