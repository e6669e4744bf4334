// jfif_front.cpp -- minimal JFIF header front end.
//
// The subset of CjfifDecode::DecodeMarker (reference source/JfifDecode.cpp:3759) that feeds
// CimgDecode: DQT (:4576-4650), SOF0/SOF1 (:4802-5039), DHT (:3401-3612), DRI (:5310-5324) and the
// first SOS (:5105-5182, :5291), issuing the same setter calls with the same arguments.  Everything
// else (EXIF, makernotes, signatures ...) is out of scope and skipped by segment length.
#include <string.h>
#include <vector>
#include "jsnoop_host.h"

static const uint8_t kZigZag[64] = {
     0, 1, 8,16, 9, 2, 3,10, 17,24,32,25,18,11, 4, 5, 12,19,26,33,40,48,41,34, 27,20,13, 6, 7,14,21,28,
    35,42,49,56,57,50,43,36, 29,22,15,23,30,37,44,51, 58,59,52,45,38,31,39,46, 53,60,61,54,47,55,62,63 };

// ITU-T T.81 Annex K.3.3 "typical" Huffman tables (Tables K.3 - K.6): code counts per length and symbol values
static const uint8_t kStdDcLumBits[16] = { 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0 };
static const uint8_t kStdDcChrBits[16] = { 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0 };
static const uint8_t kStdDcVals[12]    = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11 };
static const uint8_t kStdAcLumBits[16] = { 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7D };
static const uint8_t kStdAcChrBits[16] = { 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77 };
static const uint8_t kStdAcLumVals[162] = {
    0x01,0x02,0x03,0x00,0x04,0x11,0x05,0x12,0x21,0x31,0x41,0x06,0x13,0x51,0x61,0x07,0x22,0x71,0x14,0x32,0x81,0x91,0xA1,0x08,0x23,0x42,0xB1,0xC1,0x15,0x52,0xD1,0xF0,
    0x24,0x33,0x62,0x72,0x82,0x09,0x0A,0x16,0x17,0x18,0x19,0x1A,0x25,0x26,0x27,0x28,0x29,0x2A,0x34,0x35,0x36,0x37,0x38,0x39,0x3A,0x43,0x44,0x45,0x46,0x47,0x48,0x49,
    0x4A,0x53,0x54,0x55,0x56,0x57,0x58,0x59,0x5A,0x63,0x64,0x65,0x66,0x67,0x68,0x69,0x6A,0x73,0x74,0x75,0x76,0x77,0x78,0x79,0x7A,0x83,0x84,0x85,0x86,0x87,0x88,0x89,
    0x8A,0x92,0x93,0x94,0x95,0x96,0x97,0x98,0x99,0x9A,0xA2,0xA3,0xA4,0xA5,0xA6,0xA7,0xA8,0xA9,0xAA,0xB2,0xB3,0xB4,0xB5,0xB6,0xB7,0xB8,0xB9,0xBA,0xC2,0xC3,0xC4,0xC5,
    0xC6,0xC7,0xC8,0xC9,0xCA,0xD2,0xD3,0xD4,0xD5,0xD6,0xD7,0xD8,0xD9,0xDA,0xE1,0xE2,0xE3,0xE4,0xE5,0xE6,0xE7,0xE8,0xE9,0xEA,0xF1,0xF2,0xF3,0xF4,0xF5,0xF6,0xF7,0xF8,
    0xF9,0xFA };
static const uint8_t kStdAcChrVals[162] = {
    0x00,0x01,0x02,0x03,0x11,0x04,0x05,0x21,0x31,0x06,0x12,0x41,0x51,0x07,0x61,0x71,0x13,0x22,0x32,0x81,0x08,0x14,0x42,0x91,0xA1,0xB1,0xC1,0x09,0x23,0x33,0x52,0xF0,
    0x15,0x62,0x72,0xD1,0x0A,0x16,0x24,0x34,0xE1,0x25,0xF1,0x17,0x18,0x19,0x1A,0x26,0x27,0x28,0x29,0x2A,0x35,0x36,0x37,0x38,0x39,0x3A,0x43,0x44,0x45,0x46,0x47,0x48,
    0x49,0x4A,0x53,0x54,0x55,0x56,0x57,0x58,0x59,0x5A,0x63,0x64,0x65,0x66,0x67,0x68,0x69,0x6A,0x73,0x74,0x75,0x76,0x77,0x78,0x79,0x7A,0x82,0x83,0x84,0x85,0x86,0x87,
    0x88,0x89,0x8A,0x92,0x93,0x94,0x95,0x96,0x97,0x98,0x99,0x9A,0xA2,0xA3,0xA4,0xA5,0xA6,0xA7,0xA8,0xA9,0xAA,0xB2,0xB3,0xB4,0xB5,0xB6,0xB7,0xB8,0xB9,0xBA,0xC2,0xC3,
    0xC4,0xC5,0xC6,0xC7,0xC8,0xC9,0xCA,0xD2,0xD3,0xD4,0xD5,0xD6,0xD7,0xD8,0xD9,0xDA,0xE2,0xE3,0xE4,0xE5,0xE6,0xE7,0xE8,0xE9,0xEA,0xF2,0xF3,0xF4,0xF5,0xF6,0xF7,0xF8,
    0xF9,0xFA };

int js_jfif_walk(JsnoopDecoder* d, const uint8_t* f, size_t n, unsigned* scan_start)
{
    auto B = [&](size_t i) -> unsigned { return i < n ? f[i] : 0u; };      // CwindowBuf::Buf: 0 past EOF
    uint8_t unzz[64]; for (int k = 0; k < 64; k++) unzz[kZigZag[k]] = (uint8_t)k;
    jsnoop_reset_state(d);
    if (n < 4 || f[0] != 0xFF || f[1] != 0xD8) { js_set_error("not a JPEG stream (no SOI)"); return -1; }
    size_t pos = 2;
    unsigned nf = 0, sof_x = 0, sof_y = 0; bool have_sof = false; int rst_en = 0; unsigned rst_interval = 0;
    // One DHT segment body (possibly several tables): the SetDhtEntry / SetDhtSize calls of CjfifDecode::DecodeDHT (:3535-3600)
    auto parse_dht = [&](auto&& byte, size_t p, size_t pend) -> bool {
        while (p < pend) {
            const unsigned tc = byte(p) >> 4, th = byte(p) & 15; p++;
            if (tc >= 2 || th >= 4) { js_set_error("DHT class/destination out of range"); return false; }
            unsigned counts[17]; for (int i = 1; i <= 16; i++) counts[i] = byte(p++);
            unsigned code = 0, ind = 0;
            for (unsigned bl = 1; bl <= 16; bl++) {
                for (unsigned i = 0; i < counts[bl]; i++) {
                    const unsigned mask = ((1u << bl) - 1) << (32 - bl);
                    if (!jsnoop_set_dht_entry(d, th, tc, ind, bl, code << (32 - bl), mask, byte(p++))) { js_set_error("too many DHT codes"); return false; }
                    ind++; code++;
                }
                code <<= 1;
            }
            if (!jsnoop_set_dht_size(d, th, tc, ind)) { js_set_error("DHT size out of range"); return false; }
        }
        return true;
    };
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
            if (!parse_dht([&](size_t i) { return B(i); }, seg, end)) return -1;
        } else if (m == 0xE0 && len >= 6 && B(seg) == 'A' && B(seg + 1) == 'V' && B(seg + 2) == 'I' && B(seg + 3) == '1') {
            // Motion JPEG frame from an AVI: no DHT in the stream, the reference imports the standard tables at this point
            // ("Importing standard Huffman table...", source/JfifDecode.cpp:4405-4421, :7987: T.81 Annex K.3-K.6 as DC0, DC1, AC0, AC1)
            std::vector<uint8_t> std_dht;
            auto put = [&](unsigned tcth, const uint8_t* bits, const uint8_t* vals, unsigned nv) { std_dht.push_back((uint8_t)tcth); std_dht.insert(std_dht.end(), bits, bits + 16); std_dht.insert(std_dht.end(), vals, vals + nv); };
            put(0x00, kStdDcLumBits, kStdDcVals, 12); put(0x01, kStdDcChrBits, kStdDcVals, 12); put(0x10, kStdAcLumBits, kStdAcLumVals, 162); put(0x11, kStdAcChrBits, kStdAcChrVals, 162);
            if (!parse_dht([&](size_t i) { return i < std_dht.size() ? (unsigned)std_dht[i] : 0u; }, 0, std_dht.size())) return -1;
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
