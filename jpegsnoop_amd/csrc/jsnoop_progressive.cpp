// jsnoop_progressive.cpp -- host side of the progressive (SOF2) decode: marker walk over ALL scans of the file, canonical
// Huffman tables as they stand at each SOS (tables may be redefined between scans), restart-interval byte ranges of every
// scan, then the scan kernels in file order, the dequantising finalize pass and the unchanged back end.
//
// "Beyond-reference" mode: the reference refuses these files (source/JfifDecode.cpp:4827-4833), the
// drop-in entry points (jsnoop_jfif_walk / jsnoop_decode_scan_img) keep refusing them the same way; this is a separate call.
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <vector>
#include "jsnoop_host.h"
#include "jsnoop_launch.h"
#include "jsnoop_progressive.h"

#include "jsnoop_bytes.h"

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    js_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); return -1; } } while (0)

namespace {

const uint8_t kZz[64] = {
     0, 1, 8,16, 9, 2, 3,10, 17,24,32,25,18,11, 4, 5, 12,19,26,33,40,48,41,34, 27,20,13, 6, 7,14,21,28,
    35,42,49,56,57,50,43,36, 29,22,15,23,30,37,44,51, 58,59,52,45,38,31,39,46, 53,60,61,54,47,55,62,63 };

struct RawDht { uint8_t counts[17]; uint8_t syms[256]; bool set = false; };

bool build_table(const RawDht& h, JsProgTable* t)                // T.81 Annex C: code sizes -> codes, F.2.2.3 decode tables
{
    memset(t, 0, sizeof *t);
    unsigned code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        t->valoff[l] = (int32_t)k - (int32_t)code;
        for (unsigned i = 0; i < h.counts[l]; i++, k++, code++) {
            if (k >= 256 || code >= (1u << l)) return false;    // over-subscribed
            t->sym[k] = h.syms[k];
            if (l <= 8) for (unsigned f = 0; f < (1u << (8 - l)); f++) t->look[(code << (8 - l)) + f] = (uint16_t)((l << 8) | h.syms[k]);
        }
        t->maxcode[l] = h.counts[l] ? (int32_t)code - 1 : -1;
        code <<= 1;
    }
    t->maxcode[17] = 0x7FFFFFFF; t->nsym = k;
    return k > 0;
}

}  // namespace

// One parsed progressive file: frame, every scan with the tables in force at its SOS and its restart intervals, dependency levels.
struct ProgImage { JsImage im; JsProgFrame fr; std::vector<JsProgScan> scans; std::vector<JsProgTable> tabs; std::vector<JsProgSeg> segs; std::vector<int> level; int nlev = 1; };

// Walks the file and fills P.  `d` supplies the preview state and the log sink and receives the geometry (js_geometry), like the
// decoder object of the baseline path; its table state is reset first.
static int prog_parse(JsnoopDecoder* d, const uint8_t* f, size_t n, ProgImage* P)
{
    if (!d || !f) { js_set_error("progressive decode: bad argument"); return -1; }
    auto B = [&](size_t i) -> unsigned { return i < n ? f[i] : 0u; };
    if (n < 4 || f[0] != 0xFF || f[1] != 0xD8) { js_set_error("not a JPEG stream (no SOI)"); return -1; }
    jsnoop_reset_state(d);

    RawDht dht[2][4]; uint16_t dqt[4][64]; bool dqt_set[4] = { false, false, false, false };
    unsigned nf = 0, X = 0, Y = 0, comp_id[3] = { 0, 0, 0 }, comp_h[3] = { 1, 1, 1 }, comp_v[3] = { 1, 1, 1 }, comp_tq[3] = { 0, 0, 0 };
    bool have_sof = false; unsigned rst_interval = 0;
    std::vector<JsProgScan>& scans = P->scans; std::vector<JsProgTable>& tabs = P->tabs; std::vector<JsProgSeg>& segs = P->segs;
    size_t pos = 2;
    while (pos + 4 <= n) {
        if (f[pos] != 0xFF) { pos++; continue; }
        while (pos < n && f[pos] == 0xFF) pos++;
        const unsigned m = B(pos++);
        if (m == 0xD8 || m == 0x01 || m == 0x00 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (m == 0xD9) break;
        const unsigned len = B(pos) * 256 + B(pos + 1);
        const size_t seg = pos + 2, end = pos + len;
        if (len < 2 || end > n) { js_set_error("truncated marker segment 0xFF%02X", m); return -1; }
        if (m == 0xDB) {
            for (size_t p = seg; p < end;) {
                const unsigned pq = B(p) >> 4, tq = B(p) & 15; p++;
                if (tq >= 4) { js_set_error("DQT destination out of range"); return -1; }
                for (int k = 0; k < 64; k++) { unsigned v = B(p++); if (pq) v = (v << 8) + B(p++); dqt[tq][kZz[k]] = (uint16_t)v; }
                dqt_set[tq] = true;
            }
        } else if (m == 0xC2) {
            if (B(seg) != 8) { js_set_error("progressive decode supports 8-bit precision only"); return -1; }
            Y = B(seg + 1) * 256 + B(seg + 2); X = B(seg + 3) * 256 + B(seg + 4); nf = B(seg + 5);
            if ((nf != 1 && nf != 3) || !X || !Y) { js_set_error("progressive decode: %u components / %ux%u not supported", nf, X, Y); return -1; }
            for (unsigned c = 0; c < nf; c++) {
                comp_id[c] = B(seg + 6 + 3 * c); comp_h[c] = B(seg + 7 + 3 * c) >> 4; comp_v[c] = B(seg + 7 + 3 * c) & 15; comp_tq[c] = B(seg + 8 + 3 * c) & 3;
                if (!comp_h[c] || !comp_v[c] || comp_h[c] > 4 || comp_v[c] > 4) { js_set_error("progressive decode: sampling factor out of range"); return -1; }
            }
            have_sof = true;
        } else if (m == 0xC0 || m == 0xC1) { js_set_error("not a progressive file (SOF%u): use jsnoop_jfif_walk + jsnoop_decode_scan_img", m - 0xC0); return -1; }
        else if (m == 0xC4) {
            for (size_t p = seg; p < end;) {
                const unsigned tc = B(p) >> 4, th = B(p) & 15; p++;
                if (tc >= 2 || th >= 4) { js_set_error("DHT class/destination out of range"); return -1; }
                RawDht& h = dht[tc][th]; unsigned tot = 0; h.counts[0] = 0;
                for (int i = 1; i <= 16; i++) { h.counts[i] = (uint8_t)B(p++); tot += h.counts[i]; }
                if (tot > 256) { js_set_error("DHT with more than 256 codes"); return -1; }
                for (unsigned i = 0; i < tot; i++) h.syms[i] = (uint8_t)B(p++);
                h.set = true;
            }
        } else if (m == 0xDD) rst_interval = B(seg) * 256 + B(seg + 1);
        else if (m == 0xDA) {
            if (!have_sof) { js_set_error("SOS before SOF2"); return -1; }
            JsProgScan sc; memset(&sc, 0, sizeof sc);
            sc.ncomp = B(seg);
            if (sc.ncomp < 1 || sc.ncomp > nf) { js_set_error("SOS with %u components", sc.ncomp); return -1; }
            unsigned td[3] = { 0, 0, 0 }, ta[3] = { 0, 0, 0 };
            for (unsigned i = 0; i < sc.ncomp; i++) {
                const unsigned id = B(seg + 1 + 2 * i); unsigned c = 0; while (c < nf && comp_id[c] != id) c++;
                if (c == nf) { js_set_error("SOS names an unknown component %u", id); return -1; }
                sc.comp[i] = c; td[i] = (B(seg + 2 + 2 * i) >> 4) & 3; ta[i] = B(seg + 2 + 2 * i) & 3;
            }
            sc.ss = B(seg + 1 + 2 * sc.ncomp); sc.se = B(seg + 2 + 2 * sc.ncomp); sc.ah = B(seg + 3 + 2 * sc.ncomp) >> 4; sc.al = B(seg + 3 + 2 * sc.ncomp) & 15;
            if (sc.ss > sc.se || sc.se > 63 || sc.al > 13 || (sc.ss == 0 && sc.se != 0) || (sc.ss > 0 && sc.ncomp != 1)) {
                js_set_error("illegal progressive scan parameters Ss=%u Se=%u Ah=%u Al=%u Ns=%u", sc.ss, sc.se, sc.ah, sc.al, sc.ncomp); return -1; }
            // the tables in force at this SOS
            for (unsigned i = 0; i < sc.ncomp; i++) {
                const bool need_dc = sc.ss == 0 && sc.ah == 0, need_ac = sc.ss > 0;
                if (need_dc || need_ac) {
                    const RawDht& h = need_dc ? dht[0][td[i]] : dht[1][ta[i]];
                    JsProgTable t;
                    if (!h.set || !build_table(h, &t)) { js_set_error("scan uses an undefined or malformed Huffman table"); return -1; }
                    uint32_t slot = sc.ntabs;
                    for (uint32_t q = 0; q < sc.ntabs; q++) if (!memcmp(&tabs[sc.tab[q]], &t, sizeof t)) slot = q;
                    if (slot == sc.ntabs) { sc.tab[sc.ntabs++] = (uint32_t)tabs.size(); tabs.push_back(t); }
                    if (need_dc) sc.dc_slot[i] = slot; else sc.ac_slot[i] = slot;
                }
            }
            // entropy data: up to the next marker that is neither stuffing nor RSTn; split at the RSTn markers
            sc.seg_first = (uint32_t)segs.size(); sc.rst_interval = rst_interval;
            size_t q = end, s0 = end;
            while (q < n) {
                q = js_next_ff(f, q, n);                                     // (sixteen bytes per step: every byte of every scan passes here)
                if (q >= n) break;
                if (f[q] == 0xFF && q + 1 < n && f[q + 1] != 0x00) {
                    if (f[q + 1] >= 0xD0 && f[q + 1] <= 0xD7) { segs.push_back({ (uint32_t)s0, (uint32_t)q }); q += 2; s0 = q; continue; }
                    if (f[q + 1] == 0xFF) { q++; continue; }                 // fill byte
                    break;
                }
                q++;
            }
            segs.push_back({ (uint32_t)s0, (uint32_t)q });
            sc.nseg = (uint32_t)segs.size() - sc.seg_first;
            scans.push_back(sc);
            pos = q; continue;
        }
        pos = end;
    }
    if (!have_sof || scans.empty()) { js_set_error("no SOF2 / no scans in the stream"); return -1; }

    // frame geometry through the same code as the baseline path (SetImageDetails / SetSofSampFactors semantics)
    JsTables& t = d->t;
    for (unsigned c = 0; c < nf; c++) {
        if (!dqt_set[comp_tq[c]]) { js_set_error("component %u selects an undefined quantisation table", c + 1); return -1; }
        jsnoop_set_sof_samp_factors(d, c + 1, comp_h[c], comp_v[c]);
        for (unsigned nat = 0; nat < 64; nat++) t.dqt_nat[comp_tq[c]][nat] = dqt[comp_tq[c]][nat];
        jsnoop_set_dqt_tables(d, c + 1, comp_tq[c]);
    }
    jsnoop_set_precision(d, 8);
    jsnoop_set_image_details(d, X, Y, nf, nf, rst_interval != 0, rst_interval);
    JsImage& im = P->im;
    if (!js_geometry(d, &im)) { js_set_error("image geometry not decodable (see log callback)"); return -1; }
    im.precision = 8; im.decode_ac = 1; im.err_max = d->opt_err_max; im.file_len = (uint32_t)n;
    im.rst_en = rst_interval != 0; im.rst_interval = rst_interval;
    im.scan_start = segs[0].start; im.scan_len = 0;
    im.preview_mode = d->preview_mode; im.shift_y = d->shift_y; im.shift_cb = d->shift_cb; im.shift_cr = d->shift_cr;
    im.shift_mcu_x = d->shift_mcu_x; im.shift_mcu_y = d->shift_mcu_y;

    JsProgFrame& fr = P->fr; memset(&fr, 0, sizeof fr);
    fr.ncomp = im.ncomp;
    for (unsigned c = 0, fb = 0; c < im.ncomp; c++) {
        fr.hs[c] = im.samp_h[c + 1]; fr.vs[c] = im.samp_v[c + 1]; fr.first_blk[c] = fb; fb += fr.hs[c] * fr.vs[c];
        for (int k = 0; k < 64; k++) fr.qnat[c][k] = dqt[comp_tq[c]][k];
    }
    const unsigned hmax = im.mcu_w / 8, vmax = im.mcu_h / 8;
    for (JsProgScan& sc : scans) {
        if (sc.ncomp == 1) {                                      // A.2.3: a non-interleaved scan covers ceil(X * Hi / Hmax / 8) x ceil(Y * Vi / Vmax / 8) blocks
            const unsigned c = sc.comp[0];
            sc.nbx = ((X * fr.hs[c] + hmax - 1) / hmax + 7) / 8; sc.nby = ((Y * fr.vs[c] + vmax - 1) / vmax + 7) / 8;
        }
        const uint32_t units = sc.ncomp > 1 ? im.mcu_xmax * im.mcu_ymax : sc.nbx * sc.nby;
        const uint32_t want = sc.rst_interval ? (units + sc.rst_interval - 1) / sc.rst_interval : 1;
        if (sc.nseg > want) sc.nseg = want;                        // surplus RSTn: ignore what follows the last expected interval
    }

    // Scans that touch different coefficients are independent (a DC scan: slot 0 of its components; an AC scan, first or
    // refinement: its band of one component, read and written by position): dependency levels, the scans of a level decode together.
    const size_t ns = scans.size();
    P->level.assign(ns, 0); P->nlev = 1;
    auto band = [](const JsProgScan& q, unsigned& lo, unsigned& hi) { if (q.ss == 0) { lo = hi = 0; } else { lo = q.ss; hi = q.se; } };
    for (size_t i = 0; i < ns; i++) {
        unsigned li, hi; band(scans[i], li, hi);
        for (size_t j = 0; j < i; j++) {
            unsigned lj, hj; band(scans[j], lj, hj);
            bool share = false;
            for (unsigned a = 0; a < scans[i].ncomp; a++) for (unsigned c = 0; c < scans[j].ncomp; c++) share = share || scans[i].comp[a] == scans[j].comp[c];
            if (share && li <= hj && lj <= hi) P->level[i] = std::max(P->level[i], P->level[j] + 1);
        }
        P->nlev = std::max(P->nlev, P->level[i] + 1);
    }
    return 0;
}

bool js_is_progressive(const uint8_t* f, size_t n)                  // the first frame header of the stream: SOF2?
{
    size_t pos = 2;
    while (pos + 4 <= n) {
        if (f[pos] != 0xFF) { pos++; continue; }
        while (pos < n && f[pos] == 0xFF) pos++;
        if (pos >= n) break;
        const unsigned m = f[pos++];
        if (m == 0xD8 || m == 0x01 || m == 0x00 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (m == 0xC2) return true;
        if ((m >= 0xC0 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) || m == 0xDA || m == 0xD9) return false;
        if (pos + 2 > n) break;
        pos += (size_t)f[pos] * 256 + f[pos + 1];
    }
    return false;
}

// ---- batch state of the progressive path (behind JsnoopBatch::prog) --------------------------------------------------------------
struct JsProgBatch {
    std::vector<JsProgFrame> frames; std::vector<JsProgScan> scans; std::vector<JsProgTable> tabs; std::vector<JsProgSeg> segs;
    std::vector<int> level; int nlev = 0;                         // per scan; levels are per image, level L of every image decodes in launch L
    std::vector<uint32_t> first_scan;                              // per image: its first scan (+ end sentinel)
    void* d_buf = nullptr; size_t d_cap = 0; bool dirty = true;
    // device views into d_buf (valid after js_prog_upload)
    JsProgFrame* d_frames = nullptr; JsProgScan* d_scans = nullptr; JsProgTable* d_tabs = nullptr; JsProgSeg* d_segs = nullptr;
    uint32_t *d_status = nullptr, *d_lvl_scans = nullptr, *d_lvl_wg = nullptr, *d_blk_base = nullptr;
    std::vector<uint32_t> lvl_first, lvl_count, lvl_wgs;          // per level: slice of d_lvl_scans / d_lvl_wg, total workgroups
    uint32_t pg_lanes = 1;                                         // intervals per wave of the sequential scan kinds (batch-wide choice)
};
size_t js_prog_count(const JsnoopBatch* b) { return b->prog ? b->prog->frames.size() : 0; }
void js_prog_dirty(JsnoopBatch* b) { if (b->prog) b->prog->dirty = true; }   // a tuning change: the work lists are rebuilt at the next decode
void js_prog_clear(JsnoopBatch* b) { if (b->prog) { JsProgBatch* g = b->prog; g->frames.clear(); g->scans.clear(); g->tabs.clear(); g->segs.clear(); g->level.clear(); g->first_scan.clear(); g->nlev = 0; g->dirty = true; } }
void js_prog_free(JsnoopBatch* b) { if (b->prog) { if (b->prog->d_buf) hipFree(b->prog->d_buf); delete b->prog; b->prog = nullptr; } }
// image `src` once more as image `dst` (tile): same scans, tables and intervals (file-relative), its own frame entry
void js_prog_dup(JsnoopBatch* b, uint32_t src, uint32_t dst)
{
    JsProgBatch* g = b->prog;
    g->frames.push_back(g->frames[src]);
    const uint32_t s0 = g->first_scan[src], s1 = g->first_scan[src + 1];
    g->first_scan.back() = (uint32_t)g->scans.size();            // sentinel becomes the first scan of dst
    for (uint32_t q = s0; q < s1; q++) { JsProgScan sc = g->scans[q]; sc.img = dst; g->scans.push_back(sc); g->level.push_back(g->level[q]); }
    g->first_scan.push_back((uint32_t)g->scans.size());
    g->dirty = true;
}

int JsnoopBatch::add_progressive(JsnoopDecoder* d, const uint8_t* f, size_t n)
{
    if (imgs.size() != js_prog_count(this)) { js_set_error("a batch holds either baseline or progressive files, not both"); return -1; }
    ProgImage P;
    if (prog_parse(d, f, n, &P)) return -1;
    if (!prog) prog = new JsProgBatch;
    JsProgBatch* g = prog;
    const int idx = add_described(P.im, f, n);
    if (idx < 0) return -1;
    const uint32_t tab0 = (uint32_t)g->tabs.size(), seg0 = (uint32_t)g->segs.size();
    if (g->first_scan.empty()) g->first_scan.push_back(0);
    g->frames.push_back(P.fr);
    g->tabs.insert(g->tabs.end(), P.tabs.begin(), P.tabs.end());
    g->segs.insert(g->segs.end(), P.segs.begin(), P.segs.end());
    for (size_t q = 0; q < P.scans.size(); q++) {
        JsProgScan sc = P.scans[q]; sc.img = (uint32_t)idx; sc.seg_first += seg0;
        for (uint32_t k = 0; k < sc.ntabs; k++) sc.tab[k] += tab0;
        g->scans.push_back(sc); g->level.push_back(P.level[q]);
    }
    g->first_scan.push_back((uint32_t)g->scans.size());
    g->nlev = std::max(g->nlev, P.nlev);
    g->dirty = true;
    return idx;
}

// scan tables, interval lists, per-level work lists and the per-image status words: one grow-only device buffer
static int js_prog_upload(JsnoopBatch* b)
{
    JsProgBatch* g = b->prog;
    const size_t nimg = g->frames.size(), nsc = g->scans.size();
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_fr = 0, o_sc = o_fr + up(nimg * sizeof(JsProgFrame)), o_tb = o_sc + up(nsc * sizeof(JsProgScan)), o_sg = o_tb + up(g->tabs.size() * sizeof(JsProgTable)),
                 o_st = o_sg + up(g->segs.size() * sizeof(JsProgSeg)), o_ls = o_st + up(nimg * 16), o_lw = o_ls + up(nsc * 4), o_bb = o_lw + up((nsc + (size_t)g->nlev) * 4),
                 total = o_bb + up((nimg + 1) * 4);
    if (g->d_cap < total) {
        if (g->d_buf) { hipStreamSynchronize(b->stream); hipFree(g->d_buf); g->d_buf = nullptr; g->d_cap = 0; }
        if (hipMalloc(&g->d_buf, total * 2) != hipSuccess) { js_set_error("hipMalloc failed (progressive tables)"); return -1; }
        g->d_cap = total * 2; g->dirty = true;
    }
    uint8_t* base = (uint8_t*)g->d_buf;
    g->d_frames = (JsProgFrame*)(base + o_fr); g->d_scans = (JsProgScan*)(base + o_sc); g->d_tabs = (JsProgTable*)(base + o_tb); g->d_segs = (JsProgSeg*)(base + o_sg);
    g->d_status = (uint32_t*)(base + o_st); g->d_lvl_scans = (uint32_t*)(base + o_ls); g->d_lvl_wg = (uint32_t*)(base + o_lw); g->d_blk_base = (uint32_t*)(base + o_bb);
    if (!g->dirty) return 0;
    // per level: the scans of that level over all images, and the exclusive prefix of their workgroup counts
    // a single file leaves the chip mostly idle: one interval per wave; a batch that fills it packs several (JSNOOP_PG_LANES overrides)
    size_t total_iv = 0; for (const JsProgScan& sc : g->scans) total_iv += sc.nseg;
    // Scans cut into many restart intervals decode one interval per LANE (k_prog_scan_lanes: 64 sequential decoders abreast in
    // predicated straight-line code); with few intervals per scan the wave-per-interval kernel keeps the lanes of a wave on one
    // block (AC refinement) or leaves them idle.  64 = lanes; 1 / 8 = intervals per wave of the wave-per-interval kernel.
    // (a lane-per-interval wave is a latency chain of its own: below ~64 x 1080p files with a marker per MCU row the chip is not full of
    //  them and the wave-per-interval kernel is as fast; 128 files: 20 against 12 Gpixel/s, 1024 files: 35 against 13)
    // Round 5 (the wave form reads through LDS and runs scalar: tools/prog_batch.py with JSNOOP_PG_LANES, N x config 5, ms per batch at 1 / 4 / 8 / 64:
    //  N = 4: 2.29 / 2.23 / 2.27 / 9.0, 8: 3.21 / 2.92 / 2.99 / 9.0, 16: 4.88 / 4.36 / 4.39 / 9.2, 32: 8.25 / 7.21 / 7.04 / 9.4, 48: 11.6 / 10.0 / 9.83 / 9.69)
    g->pg_lanes = (nsc && total_iv / nsc >= 16 && total_iv >= 88000) ? 64u : (total_iv > 40000 ? 8u : (total_iv > 7000 ? 4u : 1u));
    if (b->tune.pg_lanes) g->pg_lanes = (uint32_t)b->tune.pg_lanes;
    std::vector<uint32_t> ls, lw; g->lvl_first.clear(); g->lvl_count.clear(); g->lvl_wgs.clear();
    for (int lv = 0; lv < g->nlev; lv++) {
        g->lvl_first.push_back((uint32_t)ls.size()); uint32_t acc = 0; const size_t w0 = lw.size();
        for (size_t q = 0; q < nsc; q++) if (g->level[q] == lv) { ls.push_back((uint32_t)q); lw.push_back(acc); acc += js_prog_wgs_of(g->scans[q], g->pg_lanes); }
        lw.push_back(acc);
        g->lvl_count.push_back((uint32_t)(ls.size() - g->lvl_first.back())); g->lvl_wgs.push_back(acc);
        (void)w0;
    }
    std::vector<uint32_t> bb(nimg + 1, 0);
    for (size_t i = 0; i < nimg; i++) bb[i + 1] = bb[i] + b->imgs[i].total_blocks;
    HIP_TRY(hipMemcpyAsync(g->d_frames, g->frames.data(), nimg * sizeof(JsProgFrame), hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipMemcpyAsync(g->d_scans, g->scans.data(), nsc * sizeof(JsProgScan), hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipMemcpyAsync(g->d_tabs, g->tabs.data(), g->tabs.size() * sizeof(JsProgTable), hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipMemcpyAsync(g->d_segs, g->segs.data(), g->segs.size() * sizeof(JsProgSeg), hipMemcpyHostToDevice, b->stream));
    if (!ls.empty()) HIP_TRY(hipMemcpyAsync(g->d_lvl_scans, ls.data(), ls.size() * 4, hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipMemcpyAsync(g->d_lvl_wg, lw.data(), lw.size() * 4, hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipMemcpyAsync(g->d_blk_base, bb.data(), bb.size() * 4, hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));                      // the host vectors above go out of scope
    g->dirty = false;
    return 0;
}

// Every scan of every image: one launch per dependency level, then the dequantising finalize pass and the unchanged back end.
int JsnoopBatch::decode_progressive(bool timed)
{
    JsProgBatch* g = prog;
    const uint32_t n = (uint32_t)imgs.size();
    if (js_prog_upload(this)) return -1;
    if (timed) HIP_TRY(hipEventRecord(ev[0], stream));
    HIP_TRY(hipMemsetAsync(g->d_status, 0, (size_t)n * 16, stream));
    HIP_TRY(hipMemsetAsync(dev.coef, 0, total_blocks * 128, stream));
    HIP_TRY(hipMemsetAsync(dev.dccum, 0, total_blocks * 2, stream));
    HIP_TRY(hipMemsetAsync(dev.side, 0, side_words * 4, stream));
    if (timed) for (int s = 1; s <= 4; s++) HIP_TRY(hipEventRecord(ev[s], stream));
    size_t wg_off = 0;
    for (int lv = 0; lv < g->nlev; lv++) {
        js_launch_prog_level(stream, dev.imgs, g->d_frames, g->d_scans, g->d_lvl_scans + g->lvl_first[lv], g->d_lvl_wg + wg_off, g->lvl_count[lv], g->lvl_wgs[lv], g->pg_lanes,
                             g->d_tabs, g->d_segs, dev.raw, dev.coef, g->d_status);
        wg_off += g->lvl_count[lv] + 1;
    }
    if (timed) HIP_TRY(hipEventRecord(ev[5], stream));            // reported under "write": the scans are this path's coefficient writers
    js_launch_prog_finalize(stream, dev.imgs, g->d_frames, n, g->d_blk_base, (uint32_t)total_blocks, dev.coef, dev.dccum);
    if (timed) { HIP_TRY(hipEventRecord(ev[6], stream)); HIP_TRY(hipEventRecord(ev[7], stream)); }
    if (launch_back_end(n)) return -1;
    if (timed) HIP_TRY(hipEventRecord(ev[8], stream));
    HIP_TRY(hipGetLastError());
    return 0;
}
int JsnoopBatch::sync_progressive()
{
    const uint32_t n = (uint32_t)imgs.size();
    std::vector<uint32_t> st((size_t)n * 4, 0);
    if (d2h_staged(st.data(), prog->d_status, st.size() * 4)) return -1;            // (through the page-locked landing buffer; it synchronises the stream)
    host_flags.assign(n, 0); host_path.assign(n, 3u);
    for (uint32_t i = 0; i < n; i++) host_flags[i] = st[(size_t)i * 4] ? JSNOOP_FLAG_BAD_CODE : 0u;
    return 0;
}

extern "C" int jsnoop_batch_add_progressive(JsnoopBatch* b, const uint8_t* file, size_t len)
{ JsnoopDecoder tmp; return b->add_progressive(&tmp, file, len); }

extern "C" int jsnoop_decode_progressive(JsnoopDecoder* d, const uint8_t* f, size_t n)
{
    if (!d || !f) { js_set_error("jsnoop_decode_progressive: bad argument"); return -1; }
    jsnoop_reset(d);
    d->preview_is_jpeg = false; d->last_path = 0; d->last_flags = 0;
    JsnoopBatch* b = d->batch;
    b->clear();
    const bool dbg_t = (d->batch->tune.debug & JSNOOP_DBG_TIMING) != 0;   // where a call's wall time goes (stderr, one line per call)
    auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tp[6]; int ntp = 0; if (dbg_t) tp[ntp++] = now_us();
    if (b->add_progressive(d, f, n) < 0) return -1;
    if (dbg_t) tp[ntp++] = now_us();
    if (b->upload()) return -1;
    if (dbg_t) tp[ntp++] = now_us();
    if (b->decode(false)) return -1;
    if (dbg_t) tp[ntp++] = now_us();
    if (b->sync()) return -1;
    if (dbg_t) { tp[ntp++] = now_us(); fprintf(stderr, "[timing] progressive: parse+stage %.0f upload %.0f enqueue %.0f wait %.0f us\n", tp[1] - tp[0], tp[2] - tp[1], tp[3] - tp[2], tp[4] - tp[3]); }
    const uint32_t status0 = b->host_flags[0];
    const int nscans = (int)(b->prog->first_scan[1] - b->prog->first_scan[0]);
    d->have_image = true; d->host_valid = 0; d->preview_is_jpeg = true;
    d->last_path = 3; d->last_flags = b->host_flags[0];
    d->side_ready = true; d->fetch_side();                         // no file map / DC maps for a multi-scan image: zeros, plus the back end's reductions
    d->hist_latched = d->opt_histo_en != 0; d->clip_latched = d->opt_stat_clip_en != 0;
    memset(d->stats, 0, sizeof d->stats); d->pending_log.clear();
    d->stats_pass(); d->flush_pending_log();
    if (status0) d->log(2, "*** ERROR: progressive scan data is malformed; the image is decoded as far as the data goes");
    return nscans;
}
