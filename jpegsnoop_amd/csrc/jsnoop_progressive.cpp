// jsnoop_progressive.cpp -- host side of the progressive (SOF2) decode: marker walk over ALL scans of the file, canonical
// Huffman tables as they stand at each SOS (tables may be redefined between scans), restart-interval byte ranges of every
// scan, then the scan kernels in file order, the dequantising finalize pass and the unchanged back end.
//
// "Beyond-reference" mode: the reference refuses these files (source/JfifDecode.cpp:4827-4833), the
// drop-in entry points (jsnoop_jfif_walk / jsnoop_decode_scan_img) keep refusing them the same way; this is a separate call.
#include <string.h>
#include <algorithm>
#include <vector>
#include "jsnoop_host.h"
#include "jsnoop_launch.h"
#include "jsnoop_progressive.h"

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

extern "C" int jsnoop_decode_progressive(JsnoopDecoder* d, const uint8_t* f, size_t n)
{
    if (!d || !f) { js_set_error("jsnoop_decode_progressive: bad argument"); return -1; }
    auto B = [&](size_t i) -> unsigned { return i < n ? f[i] : 0u; };
    if (n < 4 || f[0] != 0xFF || f[1] != 0xD8) { js_set_error("not a JPEG stream (no SOI)"); return -1; }
    jsnoop_reset(d); jsnoop_reset_state(d);
    d->preview_is_jpeg = false; d->last_path = 0; d->last_flags = 0;

    RawDht dht[2][4]; uint16_t dqt[4][64]; bool dqt_set[4] = { false, false, false, false };
    unsigned nf = 0, X = 0, Y = 0, comp_id[3] = { 0, 0, 0 }, comp_h[3] = { 1, 1, 1 }, comp_v[3] = { 1, 1, 1 }, comp_tq[3] = { 0, 0, 0 };
    bool have_sof = false; unsigned rst_interval = 0;
    std::vector<JsProgScan> scans; std::vector<JsProgTable> tabs; std::vector<JsProgSeg> segs;
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
    JsImage im;
    if (!js_geometry(d, &im)) { js_set_error("image geometry not decodable (see log callback)"); return -1; }
    im.precision = 8; im.decode_ac = 1; im.err_max = d->opt_err_max; im.file_len = (uint32_t)n;
    im.rst_en = rst_interval != 0; im.rst_interval = rst_interval;
    im.scan_start = segs[0].start; im.scan_len = 0;
    im.preview_mode = d->preview_mode; im.shift_y = d->shift_y; im.shift_cb = d->shift_cb; im.shift_cr = d->shift_cr;
    im.shift_mcu_x = d->shift_mcu_x; im.shift_mcu_y = d->shift_mcu_y;

    JsProgFrame fr; memset(&fr, 0, sizeof fr);
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

    // stage through the decoder's private batch: arenas, file bytes, descriptors
    JsnoopBatch* b = d->batch;
    b->clear();
    if (b->add_described(im, f, n) < 0) return -1;
    if (b->upload()) return -1;
    HIP_TRY(hipSetDevice(b->device));
    const JsImage& dim = b->imgs[0];
    // scan tables, interval list and status word live in one grow-only device buffer of the batch (no allocation per call)
    const size_t tab_bytes = (tabs.size() * sizeof(JsProgTable) + 255) & ~(size_t)255, seg_bytes = (segs.size() * sizeof(JsProgSeg) + 255) & ~(size_t)255;
    if (b->prog_cap < tab_bytes + seg_bytes + 256) {
        if (b->prog_buf) { hipStreamSynchronize(b->stream); hipFree(b->prog_buf); b->prog_buf = nullptr; b->prog_cap = 0; }
        const size_t want = (tab_bytes + seg_bytes + 256) * 2;
        if (hipMalloc(&b->prog_buf, want) != hipSuccess) { b->prog_buf = nullptr; js_set_error("hipMalloc failed"); return -1; }
        b->prog_cap = want;
    }
    JsProgTable* d_tabs = (JsProgTable*)b->prog_buf; JsProgSeg* d_segs = (JsProgSeg*)((uint8_t*)b->prog_buf + tab_bytes);
    uint32_t* d_status = (uint32_t*)((uint8_t*)b->prog_buf + tab_bytes + seg_bytes);
    HIP_TRY(hipMemcpyAsync(d_tabs, tabs.data(), tabs.size() * sizeof(JsProgTable), hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipMemcpyAsync(d_segs, segs.data(), segs.size() * sizeof(JsProgSeg), hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipMemsetAsync(d_status, 0, 16, b->stream));
    HIP_TRY(hipMemsetAsync(b->dev.coef + dim.coef_off * 64, 0, (size_t)dim.total_blocks * 128, b->stream));
    HIP_TRY(hipMemsetAsync(b->dev.dccum + dim.coef_off, 0, (size_t)dim.total_blocks * 2, b->stream));
    HIP_TRY(hipMemsetAsync(b->dev.side, 0, b->side_words * 4, b->stream));
    // Scans that touch different coefficients are independent (a DC scan: slot 0 of its components; an AC scan, first or
    // refinement: its band of one component, read and written by position).  Levels of the dependency order run one after the other, the scans of a level side by side on
    // helper streams: a kernel of a few dozen single-lane decoders leaves the chip empty.
    const size_t ns = scans.size();
    std::vector<int> level(ns, 0); int nlev = 1;
    auto band = [](const JsProgScan& q, unsigned& lo, unsigned& hi) { if (q.ss == 0) { lo = hi = 0; } else { lo = q.ss; hi = q.se; } };
    for (size_t i = 0; i < ns; i++) {
        unsigned li, hi; band(scans[i], li, hi);
        for (size_t j = 0; j < i; j++) {
            unsigned lj, hj; band(scans[j], lj, hj);
            bool share = false;
            for (unsigned a = 0; a < scans[i].ncomp; a++) for (unsigned c = 0; c < scans[j].ncomp; c++) share = share || scans[i].comp[a] == scans[j].comp[c];
            if (share && li <= hj && lj <= hi) level[i] = std::max(level[i], level[j] + 1);
        }
        nlev = std::max(nlev, level[i] + 1);
    }
    const bool fork = ns > 1 && nlev < (int)ns && b->ensure_aux() == 0;
    for (int lv = 0; lv < nlev; lv++) {
        unsigned used = 0, k = 0;
        if (fork) hipEventRecord(b->aux_ev[JsnoopBatch::kAux], b->stream);
        for (size_t i = 0; i < ns; i++) {
            if (level[i] != lv) continue;
            hipStream_t st = b->stream;
            if (fork && k % (JsnoopBatch::kAux + 1)) {
                const unsigned a = k % (JsnoopBatch::kAux + 1) - 1; st = b->aux[a];
                if (!(used >> a & 1u)) { hipStreamWaitEvent(st, b->aux_ev[JsnoopBatch::kAux], 0); used |= 1u << a; }
            }
            js_launch_prog_scan(st, b->dev.imgs, fr, scans[i], d_tabs, d_segs, b->dev.raw, b->dev.coef, d_status);
            k++;
        }
        for (unsigned a = 0; a < JsnoopBatch::kAux; a++) if (used >> a & 1u) { hipEventRecord(b->aux_ev[a], b->aux[a]); hipStreamWaitEvent(b->stream, b->aux_ev[a], 0); }
    }
    js_launch_prog_finalize(b->stream, b->dev.imgs, fr, dim.total_blocks, b->dev.coef, b->dev.dccum);
    if (b->launch_back_end(1)) return -1;
    uint32_t status[4] = { 0, 0, 0, 0 };
    hipError_t e = hipMemcpyAsync(status, d_status, 16, hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    if (e != hipSuccess) { js_set_error("progressive decode: device error: %s", hipGetErrorString(e)); return -1; }
    b->host_flags.assign(1, status[0] ? JSNOOP_FLAG_BAD_CODE : 0u); b->host_path.assign(1, 3u);
    d->have_image = true; d->host_valid = 0; d->preview_is_jpeg = true;
    d->last_path = 3; d->last_flags = b->host_flags[0];
    d->side_ready = true; d->fetch_side();                         // no file map / DC maps for a multi-scan image: zeros, plus the back end's reductions
    d->hist_latched = d->opt_histo_en != 0; d->clip_latched = d->opt_stat_clip_en != 0;
    memset(d->stats, 0, sizeof d->stats); d->pending_log.clear();
    d->stats_pass(); d->flush_pending_log();
    if (status[0]) d->log(2, "*** ERROR: progressive scan data is malformed (status 0x%X); the image is decoded as far as the data goes", status[0]);
    return (int)scans.size();
}
