// jsnoop_report.cpp -- the text CimgDecode::DecodeScanImg writes to CDocLog, reproduced through the log callback
// ("log output drop-in").  Everything here is formatting: the numbers come from the side block,
// the colour statistics and the event records the device kernels produced.
//
//   while decoding   : the reader's messages (reference source/ImgDecode.cpp:1102, :1204, :1248, :1269, :1419, :1536,
//                      :1693, :1743, :1782, :2637, :3196) -- event records of the exact-mirror reader; for images the
//                      parallel path decoded they reduce to restart-marker bookkeeping (derived here from the marker bytes
//                      and the MCU file map) and the markers the look-ahead meets at the end of the scan
//   after the decode : compression statistics, Huffman code-length histogram, colour statistics, average luminance,
//                      brightest pixel, closing lines (:3653-3745, ReportColorStats :3765, DrawHistogram :3900)
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "jsnoop_host.h"
#include "jsnoop_launch.h"

namespace {

struct Ev { uint32_t kind, a[5]; uint64_t order; };

void only_reported(JsnoopDecoder* d, unsigned& count)        // the counter every counted message shares (m_nWarnBadScanNum)
{
    count++;
    if (count >= d->opt_err_max) d->log(2, "    Only reported first %u instances of this message...", d->opt_err_max);
}

void emit_event(JsnoopDecoder* d, const JsImage& im, const Ev& e, unsigned& count)
{
    switch (e.kind) {
    case JS_EV_OVERREAD_BEFORE: d->log(2, "*** ERROR: Overread scan segment (before nCode)! @ Offset: 0x%08X.%u", e.a[0], e.a[1]); only_reported(d, count); break;
    case JS_EV_OVERREAD_CODE:   d->log(2, "*** ERROR: Overread scan segment (after nCode)! @ Offset: 0x%08X.%u", e.a[0], e.a[1]); break;
    case JS_EV_OVERREAD_BITS:   d->log(2, "*** ERROR: Overread scan segment (after bitstring)! @ Offset: 0x%08X.%u", e.a[0], e.a[1]); break;
    case JS_EV_CANT_FIND:       d->log(2, "*** ERROR: Can't find huffman bitstring @ 0x%08X.%u, table %u, value [0x%08x]", e.a[0], e.a[1], e.a[2], e.a[3]); only_reported(d, count); break;
    case JS_EV_RST_INDEX:       d->log(2, "  ERROR: Expected RST marker index RST%u got RST%u @ 0x%08X.0", e.a[0], e.a[1], e.a[2]); break;
    case JS_EV_MARKER:
        d->log(0, "  Scan Data encountered marker   0xFF%02X @ 0x%08X.0", e.a[0], e.a[1]);
        if (e.a[0] != 0xD9) d->log(2, "  NOTE: Marker wasn't EOI (0xFFD9)");
        only_reported(d, count); break;
    case JS_EV_BAD_MARKER:      d->log(2, "*** ERROR: Bad marker @ 0x%08X.%u", e.a[0], e.a[1]); only_reported(d, count); break;
    case JS_EV_BAD_HUFF:        d->log(2, "*** ERROR: Bad huffman code @ 0x%08X.%u", e.a[0], e.a[1]); only_reported(d, count); break;
    case JS_EV_NUMCOEF:         d->log(2, "*** ERROR: @ 0x%08X.%u, nNumCoeffs>64 [%u]", e.a[0], e.a[1], e.a[2]); only_reported(d, count); break;
    case JS_EV_BAD_SCAN_MCU: {
        const unsigned mx = e.a[0] & 0xFFFF, my = e.a[0] >> 16, comp = e.a[1] & 0xFF, ch = (e.a[1] >> 8) & 0xFF, cv = (e.a[1] >> 16) & 0xFF;
        char css[48];
        snprintf(css, sizeof css, comp == 1 ? "Lum CSS(%u,%u)" : comp == 2 ? "Chr(Cb) CSS(%u,%u)" : comp == 3 ? "Chr(Cr) CSS(%u,%u)" : "??? CSS(%u,%u)", ch, cv);
        d->log(2, "*** ERROR: Bad scan data in MCU(%u,%u): %s @ Offset 0x%08X.%u", mx, my, css, e.a[2], e.a[3]);
        d->log(2, "           MCU located at pixel=(%u,%u)", im.mcu_w * mx + ch * 8, im.mcu_h * my + cv * 8);
        only_reported(d, count); break; }
    case JS_EV_RST_NOT_DETECTED:
        d->log(0, "  Expect Restart interval elapsed @ 0x%08X.%u", e.a[0], e.a[1]);
        d->log(2, "    ERROR: Restart marker not detected"); break;
    default: break;
    }
}

}  // namespace

// What the reader's very first refill logs (DecodeRestartScanBuf + BuffTopup, :3007-3019 -- BEFORE "*** Decoding SCAN Data ***", :3022): BuffAddByte's
// rules (:1386-1573) over the first bytes of the scan, until the 32-bit register is full or an RSTn is met.  The device readers log the same events
// first in their lists: js_emit_decode_events leaves those out again.
void js_emit_head_events(JsnoopDecoder* d, const uint8_t* file, size_t len, uint32_t scan_start)
{
    JsImage none{};
    unsigned count = 0, n = 0; uint32_t ptr = scan_start, vacant = 32;
    auto byte = [&](uint32_t o) -> uint32_t { return o < len ? file[o] : 0u; };
    while (vacant >= 8) {
        const uint32_t b0 = byte(ptr), b1 = byte(ptr + 1);
        if (b0 == 0xFF && b1 >= 0xD0 && b1 <= 0xD7) {
            if (b1 - 0xD0 != 0) { Ev e{}; e.kind = JS_EV_RST_INDEX; e.a[0] = 0; e.a[1] = b1 - 0xD0; e.a[2] = ptr; emit_event(d, none, e, count); n++; }     // (RST0 is what the first marker is expected to be, :3013)
            break;
        }
        if (b0 == 0xFF && b1 == 0x00) ptr += 2;
        else if (b0 == 0xFF && b1 == 0xFF) ptr += 1;
        else if (b0 == 0xFF) { if (count < d->opt_err_max) { Ev e{}; e.kind = JS_EV_MARKER; e.a[0] = b1; e.a[1] = ptr; emit_event(d, none, e, count); n++; } ptr += 1; }
        else ptr += 1;
        vacant -= 8;
    }
    d->head_events = n; d->head_counted = count;
}

// Messages of the decode loop.  Returns after the last of them; the caller adds the report.
void js_emit_decode_events(JsnoopDecoder* d)
{
    JsnoopBatch* b = d->batch;
    if (!d->have_image || !b->event_words) return;
    const JsImage& im = b->imgs[d->img];
    hipSetDevice(b->device);
    // a flagged image of the parallel path whose report came from the chunked side pass: the messages are on the host already, in the
    // reference's order, each lane's gated by its own count -- the shared counter is applied here
    if (d->last_path == 1 && (size_t)d->img < b->side_mode.size() && b->side_mode[d->img] == 3) {
        const std::vector<uint32_t>& sv = b->side_events[d->img];
        unsigned cnt = d->head_counted;                             // (the first refill's messages went out in front of the heading: js_emit_head_events)
        for (size_t k = (size_t)d->head_events * JS_EV_WORDS; k + JS_EV_WORDS <= sv.size(); k += JS_EV_WORDS) {
            Ev e; e.kind = sv[k]; for (int q = 0; q < 5; q++) e.a[q] = sv[k + 1 + q]; e.order = k;
            const bool counted = e.kind == JS_EV_OVERREAD_BEFORE || e.kind == JS_EV_CANT_FIND || e.kind == JS_EV_MARKER || e.kind == JS_EV_BAD_MARKER ||
                                 e.kind == JS_EV_BAD_HUFF || e.kind == JS_EV_NUMCOEF || e.kind == JS_EV_BAD_SCAN_MCU;
            if (counted && cnt >= d->opt_err_max) continue;
            emit_event(d, im, e, cnt);
        }
        return;
    }
    std::vector<uint32_t> raw(1 + (size_t)JS_EV_WORDS * JS_EV_MAX);
    const bool cached = d->report_cache && d->h_events.size() == raw.size();      // (fetched with the side block, JsnoopDecoder::fetch_side)
    if (cached) raw = d->h_events;
    else if (b->d2h_staged(raw.data(), b->dev.events + im.ev_off, raw.size() * 4)) {   /* page-locked landing buffer: no pageable asynchronous copies */ d->log(2, "*** ERROR: reading the decoder's event log back from the device failed ***"); return; }
    std::vector<Ev> evs;
    const uint32_t n = std::min<uint32_t>(raw[0], JS_EV_MAX);
    const uint32_t nmcu = im.mcu_xmax * im.mcu_ymax;
    // clean image of the parallel path: only the end-of-scan markers come from the device (parallel side pass), the restart bookkeeping is
    // derived below; otherwise the exact-mirror reader wrote every message itself (at decode time, or in the side-only pass of an
    // image whose flags are bookkeeping only)
    // ... or the parallel side pass with its overflow records (js_side_only says which of the two produced the side outputs)
    const bool derived = d->last_path == 1 && ((size_t)d->img < b->side_mode.size() && b->side_mode[d->img] ? b->side_mode[d->img] == 1 : d->last_flags == 0);
    for (uint32_t i = derived ? 0u : std::min<uint32_t>(n, d->head_events); i < n; i++) {     // (the mirror's list starts with the first refill's messages: out already)
        Ev e; e.kind = raw[1 + i * JS_EV_WORDS]; for (int k = 0; k < 5; k++) e.a[k] = raw[2 + i * JS_EV_WORDS + k];
        e.order = ((uint64_t)(derived ? nmcu : 0) << 32) | (2u << 28) | i;
        evs.push_back(e);
    }
    if (derived) {
        // Restart bookkeeping of a well-formed scan, from the marker bytes and the per-MCU restart flags: an RSTn whose
        // number is not the expected one (:1416-1423, logged when the refill meets it, i.e. before the MCU behind it), and
        // an elapsed restart interval with no marker in the stream (:3180-3200, logged at the top of that MCU).
        std::vector<uint8_t> rstf(nmcu);
        if (d->report_cache && d->h_rstf.size() == nmcu) rstf = d->h_rstf;
        else if (b->d2h_staged(rstf.data(), b->dev.mcu_rst + im.mcu_off, nmcu)) { d->log(2, "*** ERROR: reading the restart flags back from the device failed ***"); return; }
        const uint8_t* f = b->pinned + im.file_off;
        uint32_t q = im.scan_start, expect = 0, left = im.rst_interval;
        const uint32_t end = im.scan_start + im.scan_len;
        // (an RSTn as the scan's first two bytes: met by the very first refill -- its message went out in front of the heading, js_emit_head_events -- and handled inside MCU 0)
        if (q + 1 < end && f[q] == 0xFF && f[q + 1] >= 0xD0 && f[q + 1] <= 0xD7) { expect = (f[q + 1] - 0xD0u + 1u) & 7u; q += 2; }
        for (uint32_t m = 0; m < nmcu; m++) {
            if (m && rstf[m]) {                                   // the RSTn in front of MCU m
                while (q + 1 < end && !(f[q] == 0xFF && f[q + 1] >= 0xD0 && f[q + 1] <= 0xD7)) q++;
                if (q + 1 < end) {
                    const uint32_t got = f[q + 1] - 0xD0u;
                    if (got != expect) { Ev e{}; e.kind = JS_EV_RST_INDEX; e.a[0] = expect; e.a[1] = got; e.a[2] = q; e.order = ((uint64_t)m << 32) | (0u << 28); evs.push_back(e); }
                    expect = (got + 1) & 7; q += 2;
                }
                left = im.rst_interval;                          // DecodeRestartScanBuf re-arms the counter (:4071)
            }
            if (im.rst_en && left == 0 && !(m && rstf[m])) {
                const uint32_t pk = d->h_side[JS_SIDE_MCUMAP + m];
                Ev e{}; e.kind = JS_EV_RST_NOT_DETECTED; e.a[0] = pk >> 4; e.a[1] = pk & 7; e.order = ((uint64_t)m << 32) | (1u << 28); evs.push_back(e);
            }
            if (im.rst_en) left--;
        }
        // An RSTn the reader's look-ahead meets BEHIND the last MCU (a file that goes on past its last MCU: a destroyed marker in front of it, say) is reported like
        // any other whose number is not the expected one (:1416-1423): the end-of-scan reader of the side pass records it with the expectation left open -- it is the
        // one the markers up to here leave (tools/fuzz_damaged_log.py seed 701 case 2164, round 6).
        for (size_t k = 0; k < evs.size();) {
            Ev& e = evs[k];
            if (e.kind == JS_EV_RST_INDEX && e.a[0] == 0xFFFFFFFFu) {
                const uint32_t got = e.a[1];
                if (got == expect) { expect = (got + 1) & 7; evs.erase(evs.begin() + (long)k); continue; }
                e.a[0] = expect; expect = (got + 1) & 7;
            }
            k++;
        }
        // Coefficient-index overflows (records of the side walk, block order): "nNumCoeffs>64" where the offending symbol starts (:1723-1735),
        // then CheckScanErrors' two lines at the end of the block (:2605-2650).  Positions are file offsets of bytes of the un-stuffed
        // stream: one pass over the scan bytes with the byte rules of BuffAddByte resolves them.
        if ((size_t)d->img < b->side_anoms.size() && !b->side_anoms[d->img].empty()) {
            const std::vector<uint32_t>& an = b->side_anoms[d->img];
            std::vector<std::pair<uint32_t, uint32_t*>> want;             // (un-stuffed byte index, where its file offset goes)
            std::vector<uint32_t> pos(an.size() / 2, 0);
            for (size_t k = 0; k < an.size() / 4; k++) { want.emplace_back(an[4 * k + 1] >> 3, &pos[2 * k]); want.emplace_back(an[4 * k + 3] >> 3, &pos[2 * k + 1]); }
            std::sort(want.begin(), want.end(), [](const std::pair<uint32_t, uint32_t*>& x, const std::pair<uint32_t, uint32_t*>& y) { return x.first < y.first; });
            const uint8_t* f = b->pinned + im.file_off; const uint32_t end = im.scan_start + im.scan_len;
            uint32_t o = im.scan_start, j = 0; size_t wi = 0;
            while (wi < want.size()) {
                if (o >= end) { *want[wi++].second = end; continue; }
                if (f[o] == 0xFF && o + 1 < end && f[o + 1] >= 0xD0 && f[o + 1] <= 0xD7) { o += 2; continue; }      // RSTn: not part of the stream
                while (wi < want.size() && want[wi].first == j) *want[wi++].second = o;
                o += (f[o] == 0xFF && o + 1 < end && f[o + 1] == 0x00) ? 2 : 1; j++;
            }
            const uint32_t nb = im.blk_per_mcu;
            for (size_t k = 0; k < an.size() / 4; k++) {
                const uint32_t blk = an[4 * k], m = blk / nb, c = blk % nb;
                Ev e1{}; e1.kind = JS_EV_NUMCOEF; e1.a[0] = pos[2 * k]; e1.a[1] = an[4 * k + 1] & 7u; e1.a[2] = an[4 * k + 2];
                e1.order = ((uint64_t)m << 32) | (2u << 28) | (c * 2u); evs.push_back(e1);
                Ev e2{}; e2.kind = JS_EV_BAD_SCAN_MCU; e2.a[0] = (m % im.mcu_xmax) | ((m / im.mcu_xmax) << 16);
                e2.a[1] = (uint32_t)im.blk_comp[c] | ((uint32_t)im.blk_ch[c] << 8) | ((uint32_t)im.blk_cv[c] << 16); e2.a[2] = pos[2 * k + 1]; e2.a[3] = an[4 * k + 3] & 7u;
                e2.order = ((uint64_t)m << 32) | (2u << 28) | (c * 2u + 1u); evs.push_back(e2);
            }
        }
        std::stable_sort(evs.begin(), evs.end(), [](const Ev& x, const Ev& y) { return x.order < y.order; });
    }
    unsigned count = derived ? 0u : d->head_counted;
    for (const Ev& e : evs) {
        // the messages that share the reference's warning counter stop once it has reached nErrMaxDecodeScan (the mirror applies that itself;
        // a merged list -- overflow records, then the markers at the end of the scan, each counted from zero -- gets it here)
        const bool counted = e.kind == JS_EV_OVERREAD_BEFORE || e.kind == JS_EV_CANT_FIND || e.kind == JS_EV_MARKER || e.kind == JS_EV_BAD_MARKER ||
                             e.kind == JS_EV_BAD_HUFF || e.kind == JS_EV_NUMCOEF || e.kind == JS_EV_BAD_SCAN_MCU;
        if (derived && counted && count >= d->opt_err_max) continue;
        emit_event(d, im, e, count);
    }
    if (raw[0] > JS_EV_MAX) d->log(1, "  (decoder log truncated: %u further messages)", raw[0] - JS_EV_MAX);
}

// The statistics report that follows the decode (only when !bQuiet), then the lines that are printed either way.
void js_emit_report(JsnoopDecoder* d, bool display, bool quiet)
{
    if (!d->have_image) return;
    const JsImage& im = d->batch->imgs[d->img];
    const JsTables& t = d->t;
    const uint32_t* sd = d->h_side.data();
    if (!quiet) {
        d->log(0, "  Compression stats:");
        const float ratio = (float)(im.dim_x * im.dim_y * im.ncomp * 8) / (float)((sd[4] - sd[7]) * 8);
        d->log(0, "    Compression Ratio: %5.2f:1", ratio);
        const float bpp = (float)((sd[4] - sd[7]) * 8) / (float)(im.dim_x * im.dim_y);
        d->log(0, "    Bits per pixel:    %5.2f:1", bpp);
        d->log(0, "");
        d->log(0, "  Huffman code histogram stats:");
        for (unsigned cls = 0; cls < 2; cls++)
            for (unsigned id = 0; id <= t.dht_setmax[cls] && id < 4; id++) {
                const uint32_t* h = sd + JS_SIDE_HISTO + (cls * 4 + id) * 17;
                unsigned total = 0; for (unsigned l = 1; l <= 16; l++) total += h[l];
                d->log(0, "    Huffman Table: (Dest ID: %u, Class: %s)", id, cls ? "AC" : "DC");
                for (unsigned l = 1; l <= 16; l++) d->log(0, "      # codes of length %02u bits: %8u (%3.0f%%)", l, h[l], (h[l] * 100.0) / total);
                d->log(0, "");
            }
        // ReportColorStats :3765-3837
        const uint32_t* st = d->stats; const int32_t* hi = reinterpret_cast<const int32_t*>(d->stats);
        const unsigned* cl = st + 37;
        d->log(0, "  YCC clipping in DC:");
        d->log(0, "    Y  component: [<0=%5u] [>255=%5u]", cl[0], cl[1]);
        d->log(0, "    Cb component: [<0=%5u] [>255=%5u]", cl[2], cl[3]);
        d->log(0, "    Cr component: [<0=%5u] [>255=%5u]", cl[4], cl[5]);
        d->log(0, "");
        if (d->hist_latched) {
            const float cnt = (float)st[36];
            auto trio = [&](const char* title, int base, const char* n0, const char* n1, const char* n2) {
                d->log(0, "%s", title);
                const char* names[3] = { n0, n1, n2 };
                for (int c = 0; c < 3; c++) d->log(0, "    %s component histo: [min=%5d max=%5d avg=%7.1f]", names[c], hi[base + 3 * c], hi[base + 3 * c + 1], (float)hi[base + 3 * c + 2] / cnt);
                d->log(0, "");
            };
            trio("  YCC histogram in DC (DCT sums : pre-ranged:", 0, "Y ", "Cb", "Cr");
            trio("  YCC histogram in DC:", 9, "Y ", "Cb", "Cr");
            trio("  RGB histogram in DC (before clip):", 27, "R ", "G ", "B ");
        }
        d->log(0, "  RGB clipping in DC:");
        d->log(0, "    R  component: [<0=%5u] [>255=%5u]", cl[6], cl[7]);
        d->log(0, "    G  component: [<0=%5u] [>255=%5u]", cl[8], cl[9]);
        d->log(0, "    B  component: [<0=%5u] [>255=%5u]", cl[10], cl[11]);
        d->log(0, "");
    }
    if (display && d->hist_latched && !quiet) {                   // DrawHistogram(bQuiet, ...) :3900-3916 (the bitmaps themselves are GDI, out of scope)
        const int32_t* hi = reinterpret_cast<const int32_t*>(d->stats); const float cnt = (float)d->stats[36];
        d->log(0, "  RGB histogram in DC (after clip):");
        const char* names[3] = { "R ", "G ", "B " };
        for (int c = 0; c < 3; c++) d->log(0, "    %s component histo: [min=%5d max=%5d avg=%7.1f]", names[c], hi[18 + 3 * c], hi[18 + 3 * c + 1], (float)hi[18 + 3 * c + 2] / cnt);
        d->log(0, "");
    }
    int ba[10]; jsnoop_bright_avg(d, ba);
    if (display && d->preview_is_jpeg) {                          // m_bAvgYValid / m_bBrightValid are set by CalcChannelPreviewFull
        d->log(0, "  Average Pixel Luminance (Y):");
        d->log(0, "    Y=[%3u] (range: 0..255)", (unsigned)ba[9]);
        d->log(0, "");
        d->log(0, "  Brightest Pixel Search:");
        d->log(0, "    YCC=[%5d,%5d,%5d] RGB=[%3u,%3u,%3u] @ MCU[%3u,%3u]", ba[1], ba[2], ba[3], (unsigned)ba[4], (unsigned)ba[5], (unsigned)ba[6], (unsigned)ba[7], (unsigned)ba[8]);
        d->log(0, "");
    }
    if (!quiet) {
        d->log(0, "  Finished Decoding SCAN Data");
        d->log(0, "    Number of RESTART markers decoded: %u", sd[2]);
        d->log(0, "    Next position in scan buffer: Offset 0x%08X.%u", sd[4], sd[5]);
        d->log(0, "");
    }
    if (display && d->hist_latched && d->opt_dump_histo_y) {      // ReportHistogramY :3740-3741, :3845-3868: m_anHistoYFull, eight bins a line
        const uint32_t* bins = d->stats + 434;
        d->log(0, "  Y Histogram in DC: (DCT sums) Full");
        for (unsigned row = 0; row < 2048 / 8; row++) {
            char line[160]; int n = snprintf(line, sizeof line, "    Y=%5d..%5d: ", -1024 + (int)(row * 8), -1024 + (int)(row * 8) + 7);
            for (unsigned col = 0; col < 8; col++) n += snprintf(line + n, sizeof line - (size_t)n, "0x%06x, ", bins[col + row * 8]);
            d->log(0, "%s", line);
        }
    }
}
