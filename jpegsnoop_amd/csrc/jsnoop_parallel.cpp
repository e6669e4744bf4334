// jsnoop_parallel.cpp -- host side of the parallel (self-synchronising) entropy path:
// LUT construction from the DHT code lists, stage launches, and the re-decode of flagged
// images on the sequential exact-mirror kernel (all on the device; no CPU decode).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "jsnoop_host.h"
#include "jsnoop_launch.h"

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    js_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); return -1; } } while (0)

// Builds the two-level decode tables of the parallel path from the code list the caller pushed
// through SetDhtEntry (canonical order, left-justified bits; reference source/JfifDecode.cpp:3535-3600).
// lut_ok stays 0 -- and the image goes to the exact-mirror kernel -- unless every table is a
// well-formed canonical prefix code whose symbols the fast path can interpret.
void js_build_parallel_luts(JsTableSet* ts, uint32_t ncomp)
{
    ts->lut_ok = 0; ts->n_rows = 0; ts->lut2_used = 0;
    memset(ts->lut1, 0, sizeof ts->lut1); memset(ts->lut2, 0, sizeof ts->lut2); memset(ts->slot_row, 0, sizeof ts->slot_row); memset(ts->lutp, 0, sizeof ts->lutp); memset(ts->lut2p, 0, sizeof ts->lut2p);
    uint32_t l2_used = 0;
    for (uint32_t slot = 0; slot < ncomp * 2; slot++) {
        const uint32_t n = ts->size[slot]; const bool is_dc = (slot & 1) == 0;
        if (n == 0 || n > 256) return;
        uint32_t prev_len = 0; uint64_t next_code = 0;           // canonical: codes count up, shifted left when the length grows
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t len = ts->bitlen[slot][i], bits = ts->bits[slot][i], sym = ts->code[slot][i];
            if (len < 1 || len > 16 || len < prev_len || sym > 255) return;
            if (ts->mask[slot][i] != (0xFFFFFFFFu << (32 - len))) return;
            next_code <<= (len - prev_len);
            if ((uint64_t)(bits >> (32 - len)) != next_code || (bits & ~ts->mask[slot][i])) return;
            if (next_code >= (1ull << len)) return;               // over-subscribed
            if (is_dc && (sym >> 4)) return;                       // a DC category with a run nibble: exact path mirrors the quirk
            next_code++; prev_len = len;
        }
        // identical code list as an earlier slot of the same class? share its row
        int shared = -1;
        for (uint32_t q = slot & 1; q < slot && shared < 0; q += 2)
            if (ts->size[q] == n && !memcmp(ts->bitlen[q], ts->bitlen[slot], n * 4) && !memcmp(ts->bits[q], ts->bits[slot], n * 4) &&
                !memcmp(ts->code[q], ts->code[slot], n * 4)) shared = (int)ts->slot_row[q];
        if (shared >= 0) { ts->slot_row[slot] = (uint32_t)shared; continue; }
        const uint32_t row = ts->n_rows++; ts->slot_row[slot] = row;
        uint16_t* l1 = ts->lut1[row];
        for (uint32_t i = 0; i < n; i++) {                        // first level: codes of <= JS_L1_BITS bits
            const uint32_t len = ts->bitlen[slot][i], top = ts->bits[slot][i] >> (32 - JS_L1_BITS), sym = ts->code[slot][i];
            if (len <= JS_L1_BITS) for (uint32_t k = 0; k < (1u << (JS_L1_BITS - len)); k++) l1[top + k] = (uint16_t)((len << 8) | sym);
        }
        for (uint32_t i = 0; i < n; ) {                           // second level: longer codes grouped by their first-level prefix
            const uint32_t len = ts->bitlen[slot][i];
            if (len <= JS_L1_BITS) { i++; continue; }
            const uint32_t prefix = ts->bits[slot][i] >> (32 - JS_L1_BITS);
            uint32_t j = i, maxlen = len;
            while (j < n && (ts->bits[slot][j] >> (32 - JS_L1_BITS)) == prefix) { maxlen = ts->bitlen[slot][j]; j++; }
            const uint32_t nb = maxlen - JS_L1_BITS;              // 1..5 extra index bits
            if (l2_used + (1u << nb) > JS_LUT2_MAX) return;
            l1[prefix] = (uint16_t)(0x8000u | (nb << 12) | l2_used);
            for (uint32_t k = i; k < j; k++) {
                const uint32_t l = ts->bitlen[slot][k], sym = ts->code[slot][k];
                const uint32_t sub = (ts->bits[slot][k] >> (32 - JS_L1_BITS - nb)) & ((1u << nb) - 1);
                for (uint32_t q = 0; q < (1u << (JS_L1_BITS + nb - l)); q++) ts->lut2[l2_used + sub + q] = (uint16_t)((l << 8) | sym);
            }
            l2_used += 1u << nb;
            i = j;
        }
    }
    ts->lut2_used = l2_used;
    // state-only pair entries, one row per distinct table (rows of DC tables describe single symbols)
    auto single = [](uint32_t e, bool is_dc) -> uint32_t {       // one symbol: bits | index advance << 8
        const uint32_t len = (e >> 8) & 31u, run = (e >> 4) & 15u, size = e & 15u;
        if (len == 0) return 0xC0000000u;                         // no code starts with these bits
        const uint32_t adv = is_dc ? 1u : ((e & 255u) == 0 ? 64u : run + 1u);
        return (len + size) | (adv << 8);
    };
    memset(ts->lut2p, 0, sizeof ts->lut2p);
    for (uint32_t slot = 0; slot < ncomp * 2; slot++) {
        const uint32_t row = ts->slot_row[slot]; const bool is_dc = (slot & 1) == 0;
        const uint16_t* l1 = ts->lut1[row]; uint32_t* lp = ts->lutp[row];
        for (uint32_t w = 0; w < (1u << JS_L1_BITS); w++) {
            const uint32_t e1 = l1[w];
            if (e1 & 0x8000u) {                                   // second level: [14:12] extra index bits, [11:0] base
                lp[w] = 0x80000000u | (e1 & 0x7FFFu);
                const uint32_t nb = (e1 >> 12) & 7u, base = e1 & 0xFFFu;
                for (uint32_t q = 0; q < (1u << nb); q++) ts->lut2p[base + q] = single(ts->lut2[base + q], is_dc);
                continue;
            }
            uint32_t v = single(e1, is_dc);
            const uint32_t bits1 = v & 255u, adv1 = (v >> 8) & 255u;
            if (!(v >> 31) && !is_dc && adv1 < 64u && bits1 < JS_L1_BITS) {
                const uint32_t known = JS_L1_BITS - bits1;                       // bits of the window behind symbol 1
                const uint32_t e2 = l1[(w << bits1) & ((1u << JS_L1_BITS) - 1u)], len2 = (e2 >> 8) & 31u;
                if (!(e2 & 0x8000u) && len2 != 0 && len2 <= known) {            // its whole code was visible
                    const uint32_t v2 = single(e2, false);
                    v |= ((bits1 + (v2 & 255u)) << 16) | ((adv1 + ((v2 >> 8) & 255u)) << 24);
                }
            }
            lp[w] = v;
        }
    }
    // value form of the tables for the write pass, rows numbered per class
    memset(ts->lutw, 0, sizeof ts->lutw); memset(ts->row_sub, 0, sizeof ts->row_sub); ts->n_dc_rows = ts->n_ac_rows = 0;
    bool seen[6] = { false, false, false, false, false, false };
    for (uint32_t slot = 0; slot < ncomp * 2; slot++) {
        const uint32_t row = ts->slot_row[slot]; const bool is_dc = (slot & 1) == 0;
        if (seen[row]) continue;
        seen[row] = true;
        ts->row_sub[row] = is_dc ? ts->n_dc_rows++ : ts->n_ac_rows++;
        const uint16_t* l1 = ts->lut1[row]; uint32_t* lw = ts->lutw[row];
        if (is_dc) {                                               // DC rows: single-symbol entries, 16 bits -- [3:0] code length, [7:4] size (the symbol), escape as in lut1
            for (uint32_t w = 0; w < (1u << JS_L1_BITS); w++) {
                const uint32_t e1 = l1[w], len1 = (e1 >> 8) & 31u;
                lw[w] = (e1 & 0x8000u) ? (0x8000u | (e1 & 0x7FFFu)) : (len1 ? (len1 | ((e1 & 15u) << 4)) : 0u);     // 16 significant bits: bit 15 = escape, 0 = no code
            }
            continue;
        }
        for (uint32_t w = 0; w < (1u << JS_L1_BITS); w++) {
            const uint32_t e1 = l1[w], len1 = (e1 >> 8) & 31u;
            if (e1 & 0x8000u) { lw[w] = 0x80000000u | (e1 & 0x7FFFu); continue; }
            if (len1 == 0) { lw[w] = 0xC0000000u; continue; }
            const uint32_t run1 = (e1 >> 4) & 15u, size1 = e1 & 15u, bits1 = len1 + size1;
            uint32_t v = len1 | (size1 << 4) | (run1 << 8);
            if ((e1 & 255u) != 0 && bits1 < JS_L1_BITS) {                          // not EOB, and bits of the window remain behind it
                const uint32_t e2 = l1[(w << bits1) & ((1u << JS_L1_BITS) - 1u)], len2 = (e2 >> 8) & 31u;
                if (!(e2 & 0x8000u) && len2 != 0 && len2 <= JS_L1_BITS - bits1)
                    v |= (len2 << 12) | ((e2 & 15u) << 16) | (((e2 >> 4) & 15u) << 20) | (1u << 24);
            }
            lw[w] = v;
        }
    }
    ts->lut_ok = 1;
}

// ---------------------------------------------------------------------------------------------------------------------
// Host-only self test of the table builders above (no device needed): random canonical Huffman tables, every first-level
// window of every table checked against a plain search through the code list -- the two-level decode tables, the state-only
// pair entries of the sync pass (+ their second level) and the value-pair entries of the write pass.  Returns the number
// of disagreements (0 = pass), -1 when no usable table set could be generated.
namespace {
struct SelfRng { uint64_t s; uint32_t next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); } uint32_t below(uint32_t n) { return next() % n; } };
void random_code(SelfRng& r, bool dc, JsTableSet* ts, uint32_t slot)
{
    std::vector<uint32_t> syms;
    if (dc) { for (uint32_t v = 0; v < 12; v++) if (v < 2 || r.below(4)) syms.push_back(v); }
    else {
        syms.push_back(0x00); syms.push_back(0xF0);
        for (uint32_t run = 0; run < 16; run++) for (uint32_t size = 1; size <= 10; size++) if (r.below(3)) syms.push_back((run << 4) | size);
    }
    for (size_t i = syms.size(); i > 1; i--) std::swap(syms[i - 1], syms[r.below((uint32_t)i)]);      // which symbol gets which length
    std::vector<uint32_t> depth = { 1, 1 };                                                          // grow a random prefix tree
    while (depth.size() < syms.size()) {
        const uint32_t k = r.below((uint32_t)depth.size());
        if (depth[k] >= 16) { bool room = false; for (uint32_t d : depth) room = room || d < 16; if (!room) break; continue; }
        depth[k]++; depth.push_back(depth[k]);
    }
    if (depth.size() > 2 && r.below(2)) depth.pop_back();                                            // sometimes an incomplete code (JPEG leaves all-ones unused)
    std::sort(depth.begin(), depth.end());
    const uint32_t n = (uint32_t)std::min(depth.size(), syms.size());
    uint32_t code = 0, prev = depth[0];
    ts->size[slot] = n;
    for (uint32_t i = 0; i < n; i++) {
        code <<= (depth[i] - prev); prev = depth[i];
        ts->bitlen[slot][i] = depth[i]; ts->bits[slot][i] = code << (32 - depth[i]); ts->mask[slot][i] = 0xFFFFFFFFu << (32 - depth[i]); ts->code[slot][i] = syms[i];
        code++;
    }
}
// plain search (what ReadScanVal's slow path does, :1110-1160): length and symbol of the code at the top of `win`, 0 = none
uint32_t search_code(const JsTableSet* ts, uint32_t slot, uint32_t win, uint32_t* sym)
{
    for (uint32_t i = 0; i < ts->size[slot]; i++) if ((win & ts->mask[slot][i]) == ts->bits[slot][i]) { *sym = ts->code[slot][i]; return ts->bitlen[slot][i]; }
    return 0;
}
}  // namespace
int js_selftest_tables(unsigned seed, unsigned rounds)
{
    SelfRng r = { 0x9E3779B97F4A7C15ull ^ seed };
    int bad = 0; unsigned usable = 0;
    std::vector<JsTableSet> store(1); JsTableSet* ts = &store[0];
    for (unsigned round = 0; round < rounds; round++) {
        memset(ts, 0, sizeof *ts);
        const bool share = r.below(2);                                                               // Cb and Cr usually share their tables
        for (uint32_t slot = 0; slot < 6; slot++) {
            if (share && slot >= 4) { ts->size[slot] = ts->size[slot - 2]; memcpy(ts->bitlen[slot], ts->bitlen[slot - 2], sizeof ts->bitlen[slot]); memcpy(ts->bits[slot], ts->bits[slot - 2], sizeof ts->bits[slot]);
                                      memcpy(ts->mask[slot], ts->mask[slot - 2], sizeof ts->mask[slot]); memcpy(ts->code[slot], ts->code[slot - 2], sizeof ts->code[slot]); }
            else random_code(r, (slot & 1) == 0, ts, slot);
        }
        js_build_parallel_luts(ts, 3);
        if (!ts->lut_ok) continue;                                                                   // second level too large for the LUT form: the exact kernel's case
        usable++;
        for (uint32_t slot = 0; slot < 6; slot++) {
            const bool is_dc = (slot & 1) == 0; const uint32_t row = ts->slot_row[slot];
            for (uint32_t w = 0; w < (1u << JS_L1_BITS); w++) for (int fill = 0; fill < 6; fill++) {
                const uint32_t win = (w << (32 - JS_L1_BITS)) | (r.next() >> JS_L1_BITS);
                uint32_t sym1 = 0; const uint32_t len1 = search_code(ts, slot, win, &sym1);
                const uint32_t size1 = sym1 & 15u, run1 = sym1 >> 4, adv1 = is_dc ? 1u : (sym1 == 0 ? 64u : run1 + 1u);
                // two-level tables
                uint32_t e = ts->lut1[row][w];
                if (e & 0x8000u) { const uint32_t nb = (e >> 12) & 7u; e = ts->lut2[(e & 0xFFFu) + ((win >> (32 - JS_L1_BITS - nb)) & ((1u << nb) - 1u))]; if (len1 && len1 <= JS_L1_BITS) bad++; }
                if (((e >> 8) & 31u) != len1 || (len1 && (e & 255u) != sym1)) bad++;
                // the AC symbol behind symbol 1, as far as the window shows it
                bool vis2 = false; uint32_t len2 = 0, sym2 = 0;
                const uint32_t bits1 = len1 + size1;
                if (!is_dc && len1 && len1 <= JS_L1_BITS && sym1 != 0 && bits1 < JS_L1_BITS) {
                    len2 = search_code(ts, slot, win << bits1, &sym2);
                    // visible: its whole code lies in what is left of the window (a prefix code: no filling of the unknown bits changes that)
                    vis2 = len2 != 0 && len2 <= JS_L1_BITS - bits1;
                }
                const uint32_t size2 = sym2 & 15u, run2 = sym2 >> 4, adv2 = sym2 == 0 ? 64u : run2 + 1u;
                // state-only pair entries of the sync pass
                uint32_t pe = ts->lutp[row][w];
                if ((int32_t)pe < 0) {
                    if (pe & 0x40000000u) { if (len1) bad++; }
                    else {
                        if (len1 && len1 <= JS_L1_BITS) bad++;
                        const uint32_t nb = (pe >> 12) & 7u; const uint32_t e2 = ts->lut2p[(pe & 0xFFFu) + ((win >> (32 - JS_L1_BITS - nb)) & ((1u << nb) - 1u))];
                        if (!len1) { if (e2 != 0xC0000000u) bad++; }
                        else if ((e2 & 255u) != bits1 || ((e2 >> 8) & 255u) != adv1 || (e2 >> 16)) bad++;
                    }
                } else {
                    if (!len1 || len1 > JS_L1_BITS || (pe & 255u) != bits1 || ((pe >> 8) & 255u) != adv1) bad++;
                    const uint32_t b12 = (pe >> 16) & 255u;
                    if (vis2 != (b12 != 0)) bad++;
                    if (vis2 && (b12 != bits1 + len2 + size2 || (pe >> 24) != adv1 + adv2)) bad++;
                }
                // write pass: DC rows hold single-symbol entries in 16 bits (code length | size << 4, bit 15 = escape, 0 = no code)
                if (is_dc) {
                    const uint32_t we = ts->lutw[row][w];
                    if (we >> 16) bad++;
                    if (we & 0x8000u) { if (len1 && len1 <= JS_L1_BITS) bad++; if ((we & 0x7FFFu) != (ts->lut1[row][w] & 0x7FFFu)) bad++; }
                    else if (!len1 || len1 > JS_L1_BITS) { if (we != 0u && !(ts->lut1[row][w] & 0x8000u)) bad++; }
                    else if ((we & 15u) != len1 || ((we >> 4) & 15u) != size1 || (we >> 8)) bad++;
                }
                // value-pair entries of the write pass (AC tables)
                if (!is_dc) {
                    const uint32_t we = ts->lutw[row][w];
                    if ((int32_t)we < 0) { if ((we & 0x40000000u) ? len1 != 0 : (len1 && len1 <= JS_L1_BITS)) bad++; if (!(we & 0x40000000u) && (we & 0x7FFFu) != (ts->lut1[row][w] & 0x7FFFu)) bad++; }
                    else {
                        if (!len1 || (we & 15u) != len1 || ((we >> 4) & 15u) != size1 || ((we >> 8) & 15u) != run1) bad++;
                        if (vis2 != (((we >> 24) & 1u) != 0)) bad++;
                        if (vis2 && (((we >> 12) & 15u) != len2 || ((we >> 16) & 15u) != size2 || ((we >> 20) & 15u) != run2)) bad++;
                    }
                }
            }
        }
    }
    return usable ? bad : -1;
}

// JSNOOP_DBG_CAND_LINKS: the links the candidate chain left open, per image, with the memos and maps around the first one (stops the stream).
void js_debug_cand_links(JsnoopBatch* b, hipStream_t st, uint32_t i0, uint32_t n)
{
    uint32_t* sub = (uint32_t*)b->dev.sub;
    if (hipStreamSynchronize(st) == hipSuccess) {
        const uint64_t ns = b->total_subseq; std::vector<uint32_t> h(6 * ns);
        if (hipMemcpy(h.data(), sub, 6 * ns * 4, hipMemcpyDeviceToHost) == hipSuccess)
            for (uint32_t k = i0; k < i0 + n; k++) {
                const JsImage& im = b->imgs[k]; uint32_t open = 0, first = 0xFFFFFFFFu;
                for (uint32_t i = 0; i < im.n_subseq; i++) {
                    const uint64_t g = im.subseq_off + i; const uint32_t lp = i ? h[g - 1] : 0u, ls = i ? h[ns + g - 1] : 0u;
                    if (lp != h[2 * ns + g] || ls != h[3 * ns + g]) { open++; if (first == 0xFFFFFFFFu) first = i; }
                }
                fprintf(stderr, "[cand] image %u: %u sub-sequences, %u open links, first at %u", k, im.n_subseq, open, first);
                if (first != 0xFFFFFFFFu) { const uint64_t g = im.subseq_off + first; fprintf(stderr, " (left exit %08x/%08x, entry %08x/%08x, exit %08x/%08x)", first ? h[g - 1] : 0u, first ? h[ns + g - 1] : 0u, h[2 * ns + g], h[3 * ns + g], h[g], h[ns + g]); }
                fprintf(stderr, "\n");
                if (first != 0xFFFFFFFFu && first > 0) {      // the memos and maps around it (layout of cand_arrays: X 2 x 6 n, memo 8 x 7 n, maps 2 n, middle states 3 n words, selections)
                    std::vector<uint32_t> c(js_cand_bytes(ns) / 4);
                    std::vector<uint32_t> dg(JS_CAND_REQ_WORDS);
                    if (hipMemcpy(c.data(), b->dev.cand, c.size() * 4, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(dg.data(), b->dev.cand_req + (size_t)k * JS_CAND_REQ_WORDS, JS_CAND_REQ_WORDS * 4, hipMemcpyDeviceToHost) == hipSuccess) {
                        const uint32_t* m = c.data() + 12 * ns; const uint32_t* r = m + 56 * ns; const uint8_t* sel = reinterpret_cast<const uint8_t*>(r + 5 * ns);
                        fprintf(stderr, "       chain diag: left %u; open %u %u %u %u; queued %u %u %u %u\n", dg[0], dg[4], dg[5], dg[6], dg[7], dg[8], dg[9], dg[10], dg[11]);
                        for (uint64_t g = im.subseq_off + first - 1; g <= im.subseq_off + first; g++) {
                            fprintf(stderr, "       sub-sequence %u: selection %u, map %08x %08x\n", (unsigned)(g - im.subseq_off), sel[g], r[2 * g], r[2 * g + 1]);
                            for (uint32_t e = 0; e < 7; e++) fprintf(stderr, "         slot %u: entry %08x/%08x exit %08x/%08x blocks %u%s\n", e, m[e * ns + g], m[7 * ns + e * ns + g], m[14 * ns + e * ns + g], m[21 * ns + e * ns + g], m[28 * ns + e * ns + g],
                                                                  e < 6 ? "" : " (filled)");
                            fprintf(stderr, "         speculative exits:"); for (uint32_t e = 0; e < 6; e++) fprintf(stderr, " %08x/%08x", c[e * ns + g], c[6 * ns + e * ns + g]); fprintf(stderr, "\n");
                        }
                    }
                }
            }
    }
}

// Launches stages 1..5 (unstuff, sync, block scan, write, DC scan) for images [i0, i0 + n) of the batch on stream st.  The arenas and
// the prefix tables are the batch's: a part passes pointers to its first image / first prefix entry (the kernels add the first
// entry to their block index) and the number of workgroups its images own.
int js_parallel_entropy_part(JsnoopBatch* b, hipStream_t st, uint32_t i0, uint32_t n, hipEvent_t* evs, hipEvent_t after_stage, int which_stage)
{
    const uint32_t N = (uint32_t)b->imgs.size();
    uint32_t* sub = (uint32_t*)b->dev.sub;
    const JsImage* imgs = b->dev.imgs + i0;
    const uint32_t* us_base = b->dev.us_base + i0; const uint32_t* sy_base = b->dev.sy_base + i0; const uint32_t* sn_base = b->dev.sy_base + (N + 1) + i0;
    uint32_t* flags = b->dev.flags + 2 * (size_t)i0;             // two words per image
    const uint32_t us_chunks = b->h_us_base[i0 + n] - b->h_us_base[i0], sy_wgs = b->h_sy_base[i0 + n] - b->h_sy_base[i0], sn_wgs = b->h_sn_base[i0 + n] - b->h_sn_base[i0];
    roctxRangePushA("jsnoop:unstuff");
    js_launch_unstuff(st, b->sub_wl, imgs, us_base, n, us_chunks, b->dev.raw, b->dev.chunk_keep, b->dev.chunk_rst,
                      b->dev.ustr_lin, b->dev.ustr, b->dev.seg, b->dev.side, flags, sy_base, sy_wgs,
                      (b->tune.cross_checks & JSNOOP_XC_UNSTUFF_3PASS) ? nullptr : b->dev.us_state, b->us_epoch,
                      b->dev.us_base + (N + 1) + i0, b->h_us4_base[i0 + n] - b->h_us4_base[i0],
                      reinterpret_cast<uint32_t*>(b->dev.us_state + b->us_chunks) + (i0 ? 1 : 0), &b->us_ticket_base[i0 ? 1 : 0]);
    roctxRangePop();
    if (evs) HIP_TRY(hipEventRecord(evs[2], st));
    if (after_stage && which_stage == 1) HIP_TRY(hipEventRecord(after_stage, st));
    roctxRangePushA("jsnoop:sub-sequence sync");
    if (b->cand_rounds >= 0) {
        // small job: candidates and a chain of look-ups instead of rounds; what the chain left open is walked by k_sync in its verification mode,
        // one more launch carries a change across workgroup boundaries
        js_launch_cand_sync(st, b->sub_wl, b->tab_rows, b->tab_lut2, imgs, sy_base, n, sy_wgs, b->cand_blk, b->dev.tables, b->dev.ustr, b->dev.seg, b->dev.side, sub, b->total_subseq,
                            b->dev.cand, b->dev.cand_req + (size_t)i0 * JS_CAND_REQ_WORDS, b->cand_rounds, b->cand_half ? 1 : 0);
        if (b->tune.debug & JSNOOP_DBG_CAND_LINKS) js_debug_cand_links(b, st, i0, n);
        // A chain that ran through needs nothing more; one that did not (the walk rounds were used up: rare) leaves links marked open, the write
        // pass's verification trips over them and js_parallel_resume repairs them with k_sync's verification mode.  (JSNOOP_CAND_VERIFY=1: run that
        // mode here, on every decode -- two launches that return at once in the normal case, 12 us of a 390 us decode.)
        if (b->tune.cross_checks & JSNOOP_XC_CAND_VERIFY) {
            js_launch_sync(st, b->sub_wl, b->tab_rows, b->tab_lut2, imgs, sn_base, n, sn_wgs, b->dev.tables, b->dev.ustr, b->dev.seg, b->dev.side, sub, b->total_subseq, 2);
            js_launch_sync(st, b->sub_wl, b->tab_rows, b->tab_lut2, imgs, sn_base, n, sn_wgs, b->dev.tables, b->dev.ustr, b->dev.seg, b->dev.side, sub, b->total_subseq, 0);
        }
    } else if (b->sync_rounds > 0) {
        uint32_t* lists = sub + 6 * (size_t)b->total_subseq + 16;            // (behind the six state arrays and their slack)
        js_launch_sync_rounds(st, b->sub_wl, b->tab_rows, b->tab_lut2, imgs, sn_base, sn_wgs, sy_base, sy_wgs, n, b->dev.tables, b->dev.ustr, b->dev.seg, b->dev.side, sub, b->total_subseq,
                              lists, lists + 2 * (size_t)b->total_subseq + (size_t)i0 * JS_SYR_SLOTS, b->sync_rounds);
    } else
    for (int l = 0; l < b->sync_launches; l++)
        js_launch_sync(st, b->sub_wl, b->tab_rows, b->tab_lut2, imgs, sn_base, n, sn_wgs, b->dev.tables, b->dev.ustr, b->dev.seg, b->dev.side, sub, b->total_subseq, l == 0);
    roctxRangePop();
    if (evs) HIP_TRY(hipEventRecord(evs[3], st));
    if (after_stage && which_stage == 2) HIP_TRY(hipEventRecord(after_stage, st));
    roctxRangePushA("jsnoop:block scan + coefficient write + DC scan");
    js_launch_block_scan(st, b->sub_wl, imgs, n, b->dev.tables, sub, b->total_subseq, b->dev.side, flags);
    if (evs) HIP_TRY(hipEventRecord(evs[4], st));
    js_launch_write(st, b->sub_wl, b->tab_rows_w, b->tab_lut2, imgs, sy_base, n, sy_wgs, b->dev.tables, b->dev.ustr, b->dev.seg, b->dev.side, sub, b->total_subseq,
                    b->dev.coef, b->dev.dccum, b->dev.mcu_rst, flags, b->cand_half ? b->dev.cand : nullptr, (b->tune.cross_checks & JSNOOP_XC_WRITE_V1) != 0, b->rec_pos);
    if (evs) HIP_TRY(hipEventRecord(evs[5], st));
    js_launch_dc_scan(st, imgs, n, b->dev.tables, b->dev.dccum, b->dev.mcu_rst, n == N ? b->dev.dc_parts : nullptr);   // (one scratch area: whole-batch launches only)
    roctxRangePop();
    if (evs) HIP_TRY(hipEventRecord(evs[6], st));
    return 1;
}
int js_parallel_entropy(JsnoopBatch* b, bool timed)
{
    bool any = false;
    for (const JsTableSet& t : b->tables) any = any || t.lut_ok;
    if (!any) return 0;
    return js_parallel_entropy_part(b, b->stream, 0, (uint32_t)b->imgs.size(), timed ? b->ev : nullptr);
}

// Sequential exact-mirror decode (entropy only) of the listed images, their intermediate ranges cleared first: coefficients, cumulative DC and the
// whole side block are the mirror's afterwards.  The PIXELS are not made here: the caller runs redo_back_end over everything that changed.
int JsnoopBatch::run_exact(const std::vector<uint32_t>& which)
{
    if (which.empty()) return 0;
    HIP_TRY(hipSetDevice(device));
    for (uint32_t i : which) {
        const JsImage& im = imgs[i];
        HIP_TRY(hipMemsetAsync(dev.coef + im.coef_off * 64, 0, (size_t)im.total_blocks * 128, stream));
        HIP_TRY(hipMemsetAsync(dev.dccum + im.coef_off, 0, (size_t)im.total_blocks * 2, stream));
        HIP_TRY(hipMemsetAsync(dev.side + im.side_off, 0, (size_t)js_side_words(im.mcu_xmax * im.mcu_ymax, im.blk_xmax * im.blk_ymax) * 4, stream));
        if (event_words) HIP_TRY(hipMemsetAsync(dev.events + im.ev_off, 0, 4, stream));     // (a side pass launched behind the decode may have logged the end-of-scan markers, js_side_prelaunch)
    }
    HIP_TRY(hipMemcpyAsync(dev.sel, which.data(), which.size() * 4, hipMemcpyHostToDevice, stream));
    js_launch_entropy_exact(stream, dev.imgs, dev.sel, (uint32_t)which.size(), dev.tables, dev.raw, dev.coef, dev.dccum, dev.side, 0, event_words ? dev.events : nullptr);
    HIP_TRY(hipGetLastError());
    return 0;
}
// The back end ran on the flagged images' first (partial) coefficients; their reductions in the side block (brightest pixel, sum of Y) are cleared
// and recomputed with their pixels.  A handful of images: one launch each over the image's own workgroups -- the batch's other images are not
// touched (one hostile file must not cost a batch its whole back end again); many: the whole batch in one launch.
int JsnoopBatch::redo_back_end(const std::vector<uint32_t>& which)
{
    if (which.empty()) return 0;
    HIP_TRY(hipSetDevice(device));
    const uint32_t n = (uint32_t)imgs.size();
    if (which.size() * 8 >= n && which.size() > 1) {
        // (words 12-13: brightest-pixel key, 15: sum of Y -- the back end's reductions; word 14, the block count of k_block_scan, stays)
        for (uint32_t i = 0; i < n; i++) { HIP_TRY(hipMemsetAsync(dev.side + imgs[i].side_off + 12, 0, 8, stream)); HIP_TRY(hipMemsetAsync(dev.side + imgs[i].side_off + 15, 0, 4, stream)); }
        if (launch_back_end(n)) return -1;
    } else for (uint32_t i : which) {
        HIP_TRY(hipMemsetAsync(dev.side + imgs[i].side_off + 12, 0, 8, stream)); HIP_TRY(hipMemsetAsync(dev.side + imgs[i].side_off + 15, 0, 4, stream));
        if (launch_back_end_part(stream, i, 1)) return -1;
    }
    HIP_TRY(hipStreamSynchronize(stream));
    HIP_TRY(hipGetLastError());
    return 0;
}

// Continues the sub-sequence synchronisation from its current state (extra k_sync launches) and redoes
// everything downstream.  Used when k_write found the chain short of its fixed point (JSNOOP_FLAG_NOSYNC).
static int js_parallel_resume(JsnoopBatch* b, int extra_launches)
{
    const uint32_t n = (uint32_t)b->imgs.size();
    uint32_t* sub = (uint32_t*)b->dev.sub;
    HIP_TRY(hipMemsetAsync(b->dev.coef, 0, b->total_blocks * 128, b->stream));
    HIP_TRY(hipMemsetAsync(b->dev.dccum, 0, b->total_blocks * 2, b->stream));
    HIP_TRY(hipMemsetAsync(b->dev.mcu_rst, 0, b->mcu_bytes, b->stream));
    if (js_clear_flags(b)) return -1;
    for (uint32_t i = 0; i < n; i++) HIP_TRY(hipMemsetAsync(b->dev.side + b->imgs[i].side_off + 12, 0, 16, b->stream));
    for (int l = 0; l < extra_launches; l++)      // the first launch checks every link (a candidate chain may have left open ones anywhere), the others carry changes across workgroup boundaries
        js_launch_sync(b->stream, b->sub_wl, b->tab_rows, b->tab_lut2, b->dev.imgs, b->dev.sy_base + (n + 1), n, b->sn_wgs, b->dev.tables, b->dev.ustr, b->dev.seg, b->dev.side, sub, b->total_subseq, l == 0 ? 2 : 0);
    js_launch_block_scan(b->stream, b->sub_wl, b->dev.imgs, n, b->dev.tables, sub, b->total_subseq, b->dev.side, b->dev.flags);
    js_launch_write(b->stream, b->sub_wl, b->tab_rows_w, b->tab_lut2, b->dev.imgs, b->dev.sy_base, n, b->sy_wgs, b->dev.tables, b->dev.ustr, b->dev.seg, b->dev.side, sub, b->total_subseq,
                    b->dev.coef, b->dev.dccum, b->dev.mcu_rst, b->dev.flags, nullptr, (b->tune.cross_checks & JSNOOP_XC_WRITE_V1) != 0);      // (the middle states are the candidate chain's: one lane per sub-sequence here)
    js_launch_dc_scan(b->stream, b->dev.imgs, n, b->dev.tables, b->dev.dccum, b->dev.mcu_rst, b->dev.dc_parts);
    if (b->launch_back_end(n)) return -1;
    return js_read_flags(b);
}

static int js_side_scratch(JsnoopBatch* b, uint32_t i, uint32_t** mcu_pos, uint32_t** us_out);
int js_clear_flags(JsnoopBatch* b)
{
    HIP_TRY(hipMemsetAsync(b->dev.flags, 0, b->imgs.size() * 8, b->stream));
    return 0;
}
int js_read_flags(JsnoopBatch* b)
{
    const size_t n = b->imgs.size();
    std::vector<uint32_t> both(2 * n);
    if (b->d2h_staged(both.data(), b->dev.flags, 2 * n * 4)) return -1;          // (through the page-locked landing buffer)
    b->host_flags.resize(n); b->host_anom.resize(n); b->host_anom_kind.resize(n);
    for (size_t i = 0; i < n; i++) {                              // (the arena keeps the complement of block << 4 | kind: 0 = none)
        const uint32_t key = ~both[2 * i + 1];
        b->host_flags[i] = both[2 * i]; b->host_anom[i] = key == 0xFFFFFFFFu ? 0xFFFFFFFFu : key >> 4; b->host_anom_kind[i] = key == 0xFFFFFFFFu ? 0 : (uint8_t)(key & 15u);
    }
    if (b->cand_rounds >= 0 && (b->tune.debug & JSNOOP_DBG_CAND)) {      // candidate chain of image 0: walks queued by the last chain launch, open sub-sequences after each launch
        uint32_t h[12]; if (b->d2h_staged(h, b->dev.cand_req, sizeof h)) return -1;
        fprintf(stderr, "[cand] rounds %d: queued by the last chain %u; open after chain 0..: %u %u %u %u; queued: %u %u %u %u\n", b->cand_rounds, h[0], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11]);
    }
    return 0;
}

// The private one-image batch a flagged image is decoded in a second time (through the markers of its scan): created on first use, on the batch's device.
// (A helper that cannot be set up costs the image its short cut, not the batch its result: nullptr.)
static JsnoopBatch* js_helper_batch(JsnoopBatch* b)
{
    if (!b->helper) {
        b->helper = new JsnoopBatch(nullptr); b->helper->device = b->device; b->helper->is_helper = true; b->helper->tune = b->tune;
        if (b->helper->init()) { delete b->helper; b->helper = nullptr; }
    }
    return b->helper;
}
int js_parallel_fixup(JsnoopBatch* b)
{
    const uint32_t n = (uint32_t)b->imgs.size();
    b->host_flags.assign(n, 0); b->host_path.assign(n, b->opt_force_exact ? 2u : 1u);
    if (b->opt_force_exact) { for (uint32_t i = 0; i < n; i++) b->host_flags[i] = JSNOOP_FLAG_FORCED; return 0; }
    if (!b->last_used_parallel) {                                   // every image went through the exact-mirror kernel already (decode tables
        for (uint32_t i = 0; i < n; i++) { b->host_flags[i] = JSNOOP_FLAG_TABLES; b->host_path[i] = 2; }   // outside the parallel path's LUT form)
        return 0;
    }
    if (js_read_flags(b)) return -1;
    // (a side pass launched behind the decode saw what the decode left: any flag -- a chain that needed more rounds included -- makes it stale)
    for (uint32_t i = 0; i < n && i < b->side_pre.size(); i++) if (b->host_flags[i]) b->side_pre[i] = 0;
    // An unconverged chain is not a malformed stream: give it more synchronisation rounds first.
    for (int attempt = 0, extra = 4; attempt < 4; attempt++, extra *= 4) {
        bool nosync = false;
        for (uint32_t i = 0; i < n; i++) nosync = nosync || (b->host_flags[i] & JSNOOP_FLAG_NOSYNC);
        if (!nosync) break;
        if (js_parallel_resume(b, extra)) return -1;
    }
    // What the flags mean for the PIXELS: a run past the 64th coefficient (JSNOOP_FLAG_COEF_OVERFLOW) ends the block without a store in the
    // reference (:1723-1735: "ncoef > 64 -> done", the value bits are consumed) exactly as in the parallel walks -- coefficients, planes
    // and DIB of such an image are already the reference's; what the flag stands for is bookkeeping (scan_bad, the warning counter, two
    // log lines per event), which the side pass produces on request (js_side_only: exact-mirror reader in side-only mode).  It is by far
    // the most common trace a damaged byte leaves (tools/damage_survey.py: 85 % of the flagged files) and no reason for a 1.2 s decode.
    // Anything else: the parallel path vouches for the blocks before the first anomaly it saw (host_anom); the exact-mirror reader takes over
    // at the top of the MCU that holds it and decodes from there to the end (k_entropy_exact in tail mode) -- a damaged byte near the end of a
    // file, a scan that ends a few blocks early or late: milliseconds instead of a sequential decode of the whole file.  An anomaly in the
    // first MCU (or none recorded), tables outside the LUT form: the whole image through the mirror, as before.
    const bool no_tail = (b->tune.cross_checks & JSNOOP_XC_NO_TAIL) != 0;     // (cross-check: every flagged image through the whole mirror)
    const bool dbg_tail = (b->tune.debug & JSNOOP_DBG_TAIL) != 0;
    // Second attempt first, for an image whose entropy data "ended" before its MCUs did although the file goes on: the scan was cut at a
    // marker that is no RSTn (or at an FF FF pair) -- which the reference does not stop at: it keeps the FF as data, reports it and reads on
    // (BuffAddByte :1486-1561).  A private one-image batch decodes the same file with the scan running to the end of the file and those
    // bytes left in the stream (us_classify); its coefficients and cumulative DC replace the image's.  What such bytes change besides
    // the data is bookkeeping (scan_bad, messages): the side-only pass of the mirror, on request, like for every flagged image.
    std::vector<uint32_t> redo;                                     // images whose coefficients change below: their pixels are made again
    // First of all the images whose FIRST anomaly is the end of the reference's own decode (value bits of a symbol past the end of a restart interval:
    // its register over-reads, :1229-1282, and no block decodes any more): everything up to that block is the parallel path's, everything behind it is
    // determined -- empty blocks, predictors standing, only the first MCU of every later row reached (:3623-3625) -- and is filled in on the device.
    // Whatever the walks met behind that block (they went on decoding bits the reference never reads) is of no consequence.
    std::vector<uint8_t> dead(n, 0);
    if (!no_tail) for (uint32_t i = 0; i < n; i++) {
        const JsImage& im = b->imgs[i];
        if (!b->host_anom_kind[i] || (b->host_flags[i] & (JSNOOP_FLAG_TABLES | JSNOOP_FLAG_NOSYNC | JSNOOP_FLAG_FORCED)) || b->host_anom[i] >= im.total_blocks) continue;
        js_launch_dead_fill(b->stream, b->dev.imgs, i, b->host_anom[i], b->host_anom_kind[i], b->dev.tables, b->dev.coef, b->dev.dccum, b->dev.mcu_rst);
        if (dbg_tail) fprintf(stderr, "[tail] image %u flags 0x%04x: the reference's decode ends in block %u (MCU %u of %u, kind %u): filled in\n", i, b->host_flags[i], b->host_anom[i],
                              b->host_anom[i] / im.blk_per_mcu, im.mcu_xmax * im.mcu_ymax, b->host_anom_kind[i]);
        dead[i] = 1; redo.push_back(i);
    }
    bool patched = false;
    if (!b->is_helper && !no_tail) for (uint32_t i = 0; i < n; i++) {
        const JsImage& im = b->imgs[i];
        if (dead[i]) continue;
        if (!(b->host_flags[i] & JSNOOP_FLAG_SHORT) || (b->host_flags[i] & (JSNOOP_FLAG_TABLES | JSNOOP_FLAG_NOSYNC | JSNOOP_FLAG_FORCED))) continue;
        if ((uint64_t)im.scan_start + im.scan_len + 2 > im.file_len || !b->tables[im.tableset].lut_ok) continue;      // the data really ends with the file
        // (a helper that cannot be set up, or whose decode fails, costs the image its short cut, not the batch its result: the image keeps
        //  its flags and goes through the mirror below)
        JsnoopBatch* h = js_helper_batch(b);
        if (!h) break;
        h->clear(); h->opt_decode_ac = (int)im.decode_ac; h->opt_want_planes = 0; h->opt_force_exact = 0;
        if (h->add_clone(b, i, true) < 0 || h->upload() || h->decode(false) || h->sync()) continue;
        const JsImage& hm = h->imgs[0];
        if (hm.total_blocks != im.total_blocks) continue;
        HIP_TRY(hipMemcpyAsync(b->dev.coef + im.coef_off * 64, h->dev.coef + hm.coef_off * 64, (size_t)im.total_blocks * 128, hipMemcpyDeviceToDevice, b->stream));
        HIP_TRY(hipMemcpyAsync(b->dev.dccum + im.coef_off, h->dev.dccum + hm.coef_off, (size_t)im.total_blocks * 2, hipMemcpyDeviceToDevice, b->stream));
        HIP_TRY(hipStreamSynchronize(b->stream));                  // the helper's arenas are reused by the next flagged image (its own stream): the copies must have left them
        if (dbg_tail) fprintf(stderr, "[tail] image %u flags 0x%04x: decoded through the markers of its scan (second attempt: path %u, flags 0x%04x)\n", i, b->host_flags[i], h->host_path[0], h->host_flags[0]);
        b->host_flags[i] = (b->host_flags[i] & JS_FLAGS_PIXEL_EXACT) | JSNOOP_FLAG_MARKER | (h->host_flags[0] & ~(uint32_t)JSNOOP_FLAG_FORCED);
        b->host_anom[i] = 0xFFFFFFFFu;                              // nothing left for the tail pass below: the second attempt had its own
        patched = true; redo.push_back(i);
    }
    std::vector<uint32_t> bad, tails;
    for (uint32_t i = 0; i < n; i++) {
        if (dead[i] || !(b->host_flags[i] & ~JS_FLAGS_PIXEL_EXACT)) continue;
        const JsImage& im = b->imgs[i];
        if ((b->host_flags[i] & JSNOOP_FLAG_MARKER) && b->host_anom[i] == 0xFFFFFFFFu) continue;    // resolved by the second attempt
        // (round 5: also behind restarts the walks followed off an MCU boundary -- the take-over reads the mark of the MCU before its own for the predictors)
        const bool tail_ok = !no_tail && !(b->host_flags[i] & (JSNOOP_FLAG_TABLES | JSNOOP_FLAG_NOSYNC | JSNOOP_FLAG_FORCED)) && b->tables[im.tableset].lut_ok &&
                             b->host_anom[i] != 0xFFFFFFFFu && b->host_anom[i] / im.blk_per_mcu >= 1u && b->host_anom[i] < im.total_blocks;
        if (dbg_tail) fprintf(stderr, "[tail] image %u flags 0x%04x first anomalous block %u (MCU %u of %u) -> %s\n", i, b->host_flags[i], b->host_anom[i],
                                                 b->host_anom[i] / im.blk_per_mcu, im.mcu_xmax * im.mcu_ymax, tail_ok ? "tail take-over" : "whole mirror");
        if (tail_ok) tails.push_back(i); else { bad.push_back(i); b->host_path[i] = 2; }
    }
    if (!tails.empty()) {
        HIP_TRY(hipSetDevice(b->device));
        for (uint32_t i : tails) {
            const JsImage& im = b->imgs[i];
            uint32_t *mcu_pos = nullptr, *us_out = nullptr;
            if (js_side_scratch(b, i, &mcu_pos, &us_out)) return -1;
            const uint32_t us0 = b->h_us_base[i], usn = b->h_us_base[i + 1] - us0, sy0 = b->h_sy_base[i], syn = b->h_sy_base[i + 1] - sy0;
            HIP_TRY(hipMemsetAsync(mcu_pos, 0, ((size_t)im.mcu_xmax * im.mcu_ymax + 2) * 4, b->stream));
            HIP_TRY(hipMemcpyAsync(b->dev.sel, &i, 4, hipMemcpyHostToDevice, b->stream));
            js_launch_tail_pass(b->stream, b->sub_wl, b->tab_rows_w, b->tab_lut2, b->dev.imgs, b->dev.us_base, b->dev.sy_base, n, us0, usn, sy0, syn, b->dev.tables, b->dev.raw,
                                b->dev.chunk_keep, b->dev.chunk_rst, b->dev.ustr, b->dev.seg, b->dev.side, (uint32_t*)b->dev.sub, b->total_subseq,
                                b->dev.coef, b->dev.dccum, b->dev.mcu_rst, mcu_pos, us_out, b->dev.flags, b->dev.sel);
            HIP_TRY(hipStreamSynchronize(b->stream));              // (dev.sel and the scratch area are reused by the next image)
            redo.push_back(i);
            if (dbg_tail) {
                const uint32_t ma = b->host_anom[i] / im.blk_per_mcu; uint32_t pos[3] = { 0, 0, 0 };
                hipMemcpy(pos, mcu_pos + (ma ? ma - 1 : 0), 12, hipMemcpyDeviceToHost);
                fprintf(stderr, "[tail] image %u: bit positions of MCU tops %u..%u: %u %u %u\n", i, ma ? ma - 1 : 0, (ma ? ma - 1 : 0) + 2, pos[0], pos[1], pos[2]);
            }
        }
        patched = true;
    }
    (void)patched;
    // Flagged images whose every MCU top the walks vouch for -- the flags are bookkeeping (JS_FLAGS_PIXEL_EXACT), or the reference's own decode ends at a
    // block the walks recorded (dead) -- get their report from the chunked side pass (js_side_only); a tail take-over, a second attempt or a whole mirror
    // decode replaced blocks the side walk knows nothing of: the mirror's side-only pass stays theirs.
    b->side_chunk_ok.assign(n, 0);
    for (uint32_t i = 0; i < n; i++) {
        if (!b->host_flags[i] || b->host_path[i] != 1 || (b->host_flags[i] & (JSNOOP_FLAG_TABLES | JSNOOP_FLAG_NOSYNC | JSNOOP_FLAG_FORCED))) continue;
        if (std::find(bad.begin(), bad.end(), i) != bad.end()) continue;
        if (std::find(tails.begin(), tails.end(), i) != tails.end()) { b->side_chunk_ok[i] = 3; continue; }       // 3: the walks vouch for the MCUs in front of block host_anom[i]; one reader goes on from there
        if (b->host_flags[i] & JSNOOP_FLAG_MARKER) { if (b->host_anom[i] == 0xFFFFFFFFu && !b->is_helper) b->side_chunk_ok[i] = 4; continue; }   // 4: decoded a second time through its markers -- the helper batch's walks are the ones to ask
        b->side_chunk_ok[i] = dead[i] ? 2 : (!(b->host_flags[i] & ~JS_FLAGS_PIXEL_EXACT) ? 1 : 0);      // (2: host_anom[i] is the block the reference's decode ends in)
    }
    // whole images through the mirror (entropy), then the pixels of everything that changed since the batch's back end ran
    if (b->run_exact(bad)) return -1;
    for (uint32_t i : bad) redo.push_back(i);
    std::sort(redo.begin(), redo.end()); redo.erase(std::unique(redo.begin(), redo.end()), redo.end());
    return b->redo_back_end(redo);
}

// Side outputs (MCU file map, block-DC maps, Huffman code-length histogram, status words) of image i, produced on
// request without touching the coefficient / pixel data.  Images the parallel path decoded get them from the
// parallel side pass (k_write<.., true> + k_side_maps + k_side_tail); images that went through the exact-mirror
// kernel already have them, and that kernel remains the producer for anything flagged.
// scratch of the side walk of image i: bit position of every MCU top, then the inverse byte map of the un-stuffing pass
static size_t js_side_scratch_bytes(const JsnoopBatch* b, uint32_t i)
{
    const JsImage& im = b->imgs[i];
    const size_t nmcu = (size_t)im.mcu_xmax * im.mcu_ymax, usn = b->h_us_base[i + 1] - b->h_us_base[i];
    return (nmcu + 2 + usn * 256 + 64 + 16 + 4 + 4 * (size_t)JS_ANOM_MAX) * 4;
}
static int js_side_scratch(JsnoopBatch* b, uint32_t i, uint32_t** mcu_pos, uint32_t** us_out)
{
    const JsImage& im = b->imgs[i];
    const size_t nmcu = (size_t)im.mcu_xmax * im.mcu_ymax, usn = b->h_us_base[i + 1] - b->h_us_base[i];
    const size_t need = (nmcu + 2 + usn * 256 + 64 + 16 + 4 + 4 * (size_t)JS_ANOM_MAX) * 4;     // ... and the overflow records of the side walk behind it
    if (need > b->side_tmp_cap) {
        if (b->d_side_tmp) hipFree(b->d_side_tmp);
        b->d_side_tmp = nullptr; b->side_tmp_cap = 0;
        HIP_TRY(hipMalloc((void**)&b->d_side_tmp, need + need / 8));
        b->side_tmp_cap = need + need / 8;
    }
    *mcu_pos = b->d_side_tmp; *us_out = *mcu_pos + ((nmcu + 2 + 15) & ~(size_t)15);        // (mcu_pos[nmcu + 1]: the top of the image's last block)
    return 0;
}
// The chunked side pass of a flagged image (k_side_chunks): 0 = done (side block complete, b->side_events[i] holds the messages), 1 = not
// representable (a chunk logged more events than its record holds): the caller falls back to the mirror's side-only pass, -1 = error.
static int js_side_chunked(JsnoopBatch* b, uint32_t i)
{
    const JsImage& im = b->imgs[i];
    const uint32_t nmcu = im.mcu_xmax * im.mcu_ymax;
    const uint32_t us0 = b->h_us_base[i], usn = b->h_us_base[i + 1] - us0, sy0 = b->h_sy_base[i], syn = b->h_sy_base[i + 1] - sy0;
    if (!usn || !syn || !nmcu) return 1;
    uint32_t *mcu_pos = nullptr, *us_out = nullptr;
    if (js_side_scratch(b, i, &mcu_pos, &us_out)) return -1;
    HIP_TRY(hipMemsetAsync(mcu_pos, 0, ((size_t)nmcu + 2) * 4, b->stream));
    if (b->event_words) HIP_TRY(hipMemsetAsync(b->dev.events + im.ev_off, 0, 4, b->stream));
    // chunks of at least 8 MCUs, at most 4096 of them; a lane keeps the first err_max counted events and some more of the uncounted kinds
    // (MCUs per exact reader: 8 / 4 / 2 give a 1080p file's report in 2.5-2.9 / 1.6-2.3 / 2.0-2.7 ms -- the readers' latency against the records the host merges)
    const uint32_t ch = std::max<uint32_t>(4u, (nmcu + 8191u) / 8192u), nchunks = (nmcu + ch - 1) / ch;
    // run_on: the MCU from which the walks do not vouch for the stream (a tail take-over): the lane of its chunk goes on alone -- and keeps the block-DC maps from its
    // chunk's first MCU on, which k_side_maps therefore leaves out (cut)
    const uint32_t run_on = b->side_chunk_ok[i] == 3 ? std::min(b->host_anom[i] / im.blk_per_mcu, nmcu - 1) : 0xFFFFFFFFu;
    const uint32_t cut = run_on == 0xFFFFFFFFu ? 0xFFFFFFFFu : run_on / ch * ch;
    js_launch_side_pass(b->stream, b->sub_wl, b->tab_rows_w, b->tab_lut2, b->dev.imgs, b->dev.us_base, b->dev.sy_base, (uint32_t)b->imgs.size(), i, us0, usn, sy0, syn,
                        b->dev.tables, b->dev.raw, b->dev.chunk_keep, b->dev.chunk_rst, b->dev.ustr, b->dev.seg, b->dev.side, (uint32_t*)b->dev.sub, b->total_subseq,
                        b->dev.dccum, b->dev.mcu_rst, mcu_pos, us_out, nullptr, nullptr, b->side_chunk_ok[i] == 2 ? b->host_anom[i] : 0xFFFFFFFFu, cut);
    const uint32_t ev_cap = std::min<uint32_t>(im.err_max, 64u) + 32u, stride = JS_SC_HDR + ev_cap * JS_EV_WORDS;      // (a chunk with more messages than that: the mirror, below)
    const size_t words = (((size_t)nchunks * stride + nmcu + 1) & ~(size_t)1) + 2 * (size_t)nmcu + nchunks + 64 + 64;
    if (words * 4 > b->chunk_tmp_cap) {
        if (b->d_chunk_tmp) hipFree(b->d_chunk_tmp);
        b->d_chunk_tmp = nullptr; b->chunk_tmp_cap = 0;
        HIP_TRY(hipMalloc((void**)&b->d_chunk_tmp, words * 4 + words / 2));
        b->chunk_tmp_cap = words * 4 + words / 2;
    }
    uint32_t* recs = b->d_chunk_tmp; uint32_t* map_own = recs + (size_t)nchunks * stride;
    const size_t bey_at = ((size_t)nchunks * stride + nmcu + 1) & ~(size_t)1;                      // (64-bit entries: chunk << 32 | packed offset, the smallest chunk wins)
    unsigned long long* map_beyond = reinterpret_cast<unsigned long long*>(recs + bey_at); uint32_t* left0 = recs + bey_at + 2 * (size_t)nmcu;
    // the restart countdown at every chunk's first MCU top (m_nRestartMcusLeft: re-armed by every restart HANDLED, :4071, one down per MCU :3618):
    // the marks of the walks say in which MCU a restart was handled -- one on the chunk's own first MCU is handled inside it, behind its top
    std::vector<uint8_t> rf(nmcu);
    if (b->d2h_staged(rf.data(), b->dev.mcu_rst + im.mcu_off, nmcu)) return -1;
    std::vector<uint32_t> left(nchunks);
    { uint32_t cur = im.rst_interval; for (uint32_t m = 0; m < nmcu; m++) { if (m % ch == 0) left[m / ch] = cur; if (rf[m]) cur = im.rst_interval; if (im.rst_en) cur--; } }
    HIP_TRY(hipMemcpyAsync(left0, left.data(), (size_t)nchunks * 4, hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipMemsetAsync(map_own, 0, (size_t)nmcu * 4, b->stream));
    HIP_TRY(hipMemsetAsync(map_beyond, 0xFF, (size_t)nmcu * 8, b->stream));
    uint32_t* fill_desc = left0 + nchunks;                          // (the closed form of a run of zero bytes, handed from the run-on lane to k_side_fill)
    HIP_TRY(hipMemsetAsync(fill_desc, 0, 64 * 4, b->stream));
    js_launch_side_chunks(b->stream, b->dev.imgs, i, b->dev.tables, b->dev.raw, b->dev.seg, b->dev.mcu_rst, mcu_pos, us_out, usn * 256u, b->dev.side, b->dev.dccum, ch, nchunks, ev_cap, left0, recs, map_own, map_beyond, run_on, fill_desc);
    // (the records are read where they land -- the page-locked landing buffer: a zeroed 3 MB vector and a copy into it were 0.3 ms of a 1080p file's report)
    const size_t h_words = bey_at + 2 * (size_t)nmcu;
    std::vector<uint32_t> h_own;
    const uint32_t* h = nullptr;
    if (h_words * 4 <= (32u << 20)) {
        if (!b->d2h_land) HIP_TRY(hipHostMalloc((void**)&b->d2h_land, 32u << 20, hipHostMallocDefault));
        HIP_TRY(hipMemcpyAsync(b->d2h_land, recs, h_words * 4, hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(hipStreamSynchronize(b->stream));
        h = reinterpret_cast<const uint32_t*>(b->d2h_land);
    } else {
        h_own.resize(h_words);
        if (b->d2h_staged(h_own.data(), recs, h_words * 4)) return -1;
        h = h_own.data();
    }
    HIP_TRY(hipGetLastError());
    uint32_t last = nchunks - 1; bool died = false;                 // died: one lane went on alone behind its chunk (the end of the decode, or the run-on lane)
    if (run_on != 0xFFFFFFFFu) { last = run_on / ch; died = true; }
    for (uint32_t c = 0; c <= last; c++) if (h[(size_t)c * stride] != 0xFFFFFFFFu) { last = c; died = true; break; }
    std::vector<uint32_t>& ev = b->side_events[i]; ev.clear();
    uint32_t sw[16] = { 0 }, histo[2 * 4 * 17] = { 0 }, scan_bad = 0, rst = 0, pix = 0; uint64_t warn = 0;
    for (uint32_t c = 0; c <= last; c++) {
        const uint32_t* r = &h[(size_t)c * stride];
        if (r[JS_SC_HDR - 1] > ev_cap) return 1;
        ev.insert(ev.end(), r + JS_SC_HDR, r + JS_SC_HDR + (size_t)r[JS_SC_HDR - 1] * JS_EV_WORDS);
        scan_bad = (r[1] & 1u) ? ((r[1] >> 1) & 1u) : (scan_bad | ((r[1] >> 1) & 1u));       // DecodeRestartScanBuf clears it (:4038-4075), the errors set it
        rst += r[4]; pix += r[5]; warn += r[6];
        for (uint32_t q = 0; q < 2 * 4 * 17; q++) histo[q] += r[8 + q];
    }
    const uint32_t* rl = &h[(size_t)last * stride];
    sw[0] = scan_bad; sw[1] = (rl[1] >> 2) & 1u; sw[2] = rst; sw[3] = pix; sw[4] = rl[2]; sw[5] = rl[3]; sw[6] = (uint32_t)std::min<uint64_t>(warn, im.err_max); sw[7] = im.scan_start;
    std::vector<uint32_t> map(nmcu);
    const uint32_t* own = &h[(size_t)nchunks * stride]; const uint32_t* bey = &h[bey_at];            // (little-endian pairs: [2m] = packed offset, [2m + 1] = chunk)
    for (uint32_t m = 0; m < nmcu; m++) map[m] = (m / ch <= last || !died) ? own[m] : (bey[2 * m + 1] == last ? bey[2 * m] : 0u);
    uint32_t* sd = b->dev.side + im.side_off;
    HIP_TRY(hipMemcpyAsync(sd, sw, 8 * 4, hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipMemcpyAsync(sd + JS_SIDE_HISTO, histo, sizeof histo, hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipMemcpyAsync(sd + JS_SIDE_MCUMAP, map.data(), (size_t)nmcu * 4, hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));                     // (the host arrays above are the copies' sources)
    return 0;
}
// An image whose scan was decoded a second time through its markers (js_parallel_fixup: helper batch): the walks that followed the reference are the
// HELPER's.  The file is decoded there once more (the helper serves one image at a time), its chunked side pass runs on the helper's arenas, and the
// side block and the messages are carried over.  0 = done, 1 = not representable there (the caller's mirror pass), -1 = error.
static int js_side_via_helper(JsnoopBatch* b, uint32_t i)
{
    const JsImage& im = b->imgs[i];
    JsnoopBatch* h = js_helper_batch(b);
    if (!h) return 1;
    h->clear(); h->opt_decode_ac = (int)im.decode_ac; h->opt_want_planes = 0; h->opt_force_exact = 0;
    if (h->add_clone(b, i, true) < 0 || h->upload() || h->decode(false) || h->sync()) return 1;
    const JsImage& hm = h->imgs[0];
    if (h->host_path[0] != 1 || (h->host_flags[0] & (JSNOOP_FLAG_TABLES | JSNOOP_FLAG_NOSYNC | JSNOOP_FLAG_FORCED | JSNOOP_FLAG_MARKER)) || hm.total_blocks != im.total_blocks) return 1;
    const size_t words = js_side_words(hm.mcu_xmax * hm.mcu_ymax, hm.blk_xmax * hm.blk_ymax);
    HIP_TRY(hipMemsetAsync(h->dev.side + hm.side_off, 0, 8 * 4, h->stream));
    HIP_TRY(hipMemsetAsync(h->dev.side + hm.side_off + JS_SIDE_HISTO, 0, (words - JS_SIDE_HISTO) * 4, h->stream));
    if (h->side_events.size() != 1) h->side_events.assign(1, std::vector<uint32_t>());
    const int rc = js_side_chunked(h, 0);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpyAsync(b->dev.side + im.side_off, h->dev.side + hm.side_off, 8 * 4, hipMemcpyDeviceToDevice, b->stream));
    HIP_TRY(hipMemcpyAsync(b->dev.side + im.side_off + JS_SIDE_HISTO, h->dev.side + hm.side_off + JS_SIDE_HISTO, (words - JS_SIDE_HISTO) * 4, hipMemcpyDeviceToDevice, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    b->side_events[i] = h->side_events[0];
    return 0;
}
// The parallel side pass of image i, enqueued on the batch stream (nothing waited for): the side block's outputs cleared, then the inverse map, the side
// walk and the maps.  with_anoms: the walk records its coefficient-index overflows (an image whose only flag is that one).
static int js_side_parallel_enqueue(JsnoopBatch* b, uint32_t i, bool with_anoms, bool clear = true, bool walked = false, hipStream_t st = nullptr)
{
    if (!st) st = b->stream;
    const JsImage& im = b->imgs[i];
    const uint32_t nmcu = im.mcu_xmax * im.mcu_ymax;
    const size_t words = js_side_words(nmcu, im.blk_xmax * im.blk_ymax);
    const uint32_t us0 = b->h_us_base[i], usn = b->h_us_base[i + 1] - us0, sy0 = b->h_sy_base[i], syn = b->h_sy_base[i + 1] - sy0;
    uint32_t *mcu_pos = nullptr, *us_out = nullptr;
    if (js_side_scratch(b, i, &mcu_pos, &us_out)) return -1;
    uint32_t* anoms = us_out + (((size_t)usn * 256 + 64 + 15) & ~(size_t)15);
    // one launch for the five areas to zero (status words 0..7, histogram + maps, MCU positions, overflow records, event counter): a memset each is ~8 us of
    // enqueue time in a call that takes a few hundred
    if (clear)
    js_launch_clear5(st, b->dev.side + im.side_off, 8, b->dev.side + im.side_off + JS_SIDE_HISTO, words - JS_SIDE_HISTO, mcu_pos, (size_t)nmcu + 2, anoms, 4,
                     b->event_words ? b->dev.events + im.ev_off : nullptr, b->event_words ? 1 : 0);      // (the pass logs the end-of-scan markers: a repeated pass must not log them twice)
    js_launch_side_pass(st, b->sub_wl, b->tab_rows_w, b->tab_lut2, b->dev.imgs, b->dev.us_base, b->dev.sy_base, (uint32_t)b->imgs.size(), i, us0, usn, sy0, syn,
                        b->dev.tables, b->dev.raw, b->dev.chunk_keep, b->dev.chunk_rst, b->dev.ustr, b->dev.seg, b->dev.side, (uint32_t*)b->dev.sub, b->total_subseq,
                        b->dev.dccum, b->dev.mcu_rst, mcu_pos, us_out, b->event_words ? b->dev.events : nullptr, with_anoms ? anoms : nullptr, 0xFFFFFFFFu, 0xFFFFFFFFu, walked);
    return 0;
}
// ... and, one step earlier: the outputs of the side pass are cleared BEFORE the decode and the decode's own write pass records what the side walk would (MCU-top
// positions, code-length histogram: k_write2<4, ., true>) -- a one-image job with 64-byte pieces.  js_side_prelaunch then launches the inverse map and the maps only.
int js_side_prepare(JsnoopBatch* b, uint32_t i)
{
    b->rec_pos = nullptr;
    if (b->imgs.size() != 1 || i != 0 || !b->uploaded || b->opt_force_exact || b->sub_wl != 4 || js_prog_count(b)) return 0;
    if (b->tune.cross_checks & (JSNOOP_XC_SIDE_EXACT | JSNOOP_XC_WRITE_V1)) return 0;
    if (!b->tables[b->imgs[0].tableset].lut_ok) return 0;
    HIP_TRY(hipSetDevice(b->device));
    const JsImage& im = b->imgs[0];
    const uint32_t nmcu = im.mcu_xmax * im.mcu_ymax, usn = b->h_us_base[1] - b->h_us_base[0];
    const size_t words = js_side_words(nmcu, im.blk_xmax * im.blk_ymax);
    uint32_t *mcu_pos = nullptr, *us_out = nullptr;
    if (js_side_scratch(b, 0, &mcu_pos, &us_out)) return -1;
    uint32_t* anoms = us_out + (((size_t)usn * 256 + 64 + 15) & ~(size_t)15);
    js_launch_clear5(b->stream, b->dev.side + im.side_off, 8, b->dev.side + im.side_off + JS_SIDE_HISTO, words - JS_SIDE_HISTO, mcu_pos, (size_t)nmcu + 2, anoms, 4,
                     b->event_words ? b->dev.events + im.ev_off : nullptr, b->event_words ? 1 : 0);
    b->rec_pos = mcu_pos;
    return 0;
}
// A caller that will ask for the side outputs of image i in any case (DecodeScanImg with a log callback) has the clean-image side pass enqueued right behind the
// decode, before the wait: js_side_only finds it done when the image turns out clean (the usual case) and runs its own otherwise.
int js_side_prelaunch(JsnoopBatch* b, uint32_t i)
{
    if (b->side_pre.size() != b->imgs.size()) b->side_pre.assign(b->imgs.size(), 0);
    b->side_pre[i] = 0;
    if (i >= b->imgs.size() || !b->uploaded || b->opt_force_exact || !b->last_used_parallel || (b->tune.cross_checks & JSNOOP_XC_SIDE_EXACT) || js_prog_count(b)) return 0;
    if (!b->tables[b->imgs[i].tableset].lut_ok) return 0;         // (the exact kernel decodes this image and fills the side block itself)
    HIP_TRY(hipSetDevice(b->device));
    const bool recorded = b->rec_pos != nullptr;                   // (js_side_prepare: cleared before the decode, walked by its write pass)
    b->rec_pos = nullptr;
    if (js_side_parallel_enqueue(b, i, false, !recorded, recorded)) return -1;
    b->side_pre[i] = 1;
    return 0;
}
// The side passes of ALL clean images of a decoded batch at once (a caller that asks for the per-image results of a batch image after image -- what
// DoBatchFileProcess does per file, source/JPEGsnoopCore.cpp:805-808): one image's pass is a chain of small launches on a nearly empty chip (0.8 ms with its
// wait; the side walk alone 0.35 ms of one wave's latency), so the second request of a decode runs them for every clean image in four launches over the whole
// batch (js_launch_side_pass_all: 256 x 1080p 214 -> ~25 ms for the side outputs of every image).  Flagged images keep their own paths (js_side_only).
static int js_side_all(JsnoopBatch* b)
{
    const uint32_t n = (uint32_t)b->imgs.size();
    if (b->side_done.size() != n) b->side_done.assign(n, 0);
    if (b->side_mode.size() != n) { b->side_mode.assign(n, 0); b->side_anoms.assign(n, std::vector<uint32_t>()); }
    if (b->side_events.size() != n) b->side_events.assign(n, std::vector<uint32_t>());
    std::vector<uint8_t> mask(n, 0); uint32_t cnt = 0;
    for (uint32_t i = 0; i < n; i++)
        if (!b->side_done[i] && i < b->host_path.size() && b->host_path[i] == 1 && b->host_flags[i] == 0 && b->tables[b->imgs[i].tableset].lut_ok) { mask[i] = 1; cnt++; }
    if (cnt < 2) return 0;
    HIP_TRY(hipSetDevice(b->device));
    const size_t o_pos = ((size_t)n + 255) & ~(size_t)255, o_us = o_pos + (((size_t)b->rec_words * 4 + 255) & ~(size_t)255), total = o_us + ((size_t)b->us_chunks * 256 + 64) * 4;
    if (total > b->side_all_cap) {
        if (b->d_side_all) hipFree(b->d_side_all);
        b->d_side_all = nullptr; b->side_all_cap = 0;
        HIP_TRY(hipMalloc((void**)&b->d_side_all, total + total / 8));
        b->side_all_cap = total + total / 8;
    }
    HIP_TRY(hipMemcpyAsync(b->d_side_all, mask.data(), n, hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipMemsetAsync(b->d_side_all + o_pos, 0, (size_t)b->rec_words * 4, b->stream));
    js_launch_side_pass_all(b->stream, b->sub_wl, b->tab_rows_w, b->tab_lut2, b->dev.imgs, b->dev.us_base, b->dev.sy_base, n, b->us_chunks, b->sy_wgs, b->dev.tables, b->dev.raw,
                            b->dev.chunk_keep, b->dev.chunk_rst, b->dev.ustr, b->dev.seg, b->dev.side, (uint32_t*)b->dev.sub, b->total_subseq, b->dev.dccum, b->dev.mcu_rst,
                            reinterpret_cast<uint32_t*>(b->d_side_all + o_pos), reinterpret_cast<uint32_t*>(b->d_side_all + o_us), b->event_words ? b->dev.events : nullptr, b->d_side_all);
    HIP_TRY(hipStreamSynchronize(b->stream));                     // (the mask vector goes out of scope)
    HIP_TRY(hipGetLastError());
    for (uint32_t i = 0; i < n; i++) if (mask[i]) { b->side_done[i] = 1; b->side_mode[i] = 1; b->side_anoms[i].clear(); b->side_events[i].clear(); if (i < b->side_pre.size()) b->side_pre[i] = 0; }
    return 0;
}
int js_side_only(JsnoopBatch* b, uint32_t i)
{
    HIP_TRY(hipSetDevice(b->device));
    // the second request for a per-image result of a batch: the clean images' passes all at once (js_side_all)
    if (b->imgs.size() > 1 && ++b->side_requests == 2 && !(b->tune.cross_checks & JSNOOP_XC_SIDE_EXACT) && !js_prog_count(b)) {
        if (js_side_all(b)) return -1;
        if (i < b->side_done.size() && b->side_done[i]) return 0;
    }
    const JsImage& im = b->imgs[i];
    const uint32_t nmcu = im.mcu_xmax * im.mcu_ymax;
    const size_t words = js_side_words(nmcu, im.blk_xmax * im.blk_ymax);
    const bool prelaunched = i < b->side_pre.size() && b->side_pre[i] == 1 && i < b->host_path.size() && b->host_path[i] == 1 && b->host_flags[i] == 0 && !(b->tune.cross_checks & JSNOOP_XC_SIDE_EXACT);
    if (!prelaunched) {
        HIP_TRY(hipMemsetAsync(b->dev.side + im.side_off, 0, 8 * 4, b->stream));
        HIP_TRY(hipMemsetAsync(b->dev.side + im.side_off + JS_SIDE_HISTO, 0, (words - JS_SIDE_HISTO) * 4, b->stream));
    }
    // An image whose only flag is the coefficient-index overflow walks exactly as the reference does (js_parallel_fixup): its maps,
    // histogram and final position come from the parallel side pass like a clean image's; what the overflows add -- scan_bad, the
    // warning counter, two messages per block -- is bookkeeping worked out below from the records the side walk leaves.
    bool parallel = i < b->host_path.size() && b->host_path[i] == 1 && (b->host_flags[i] & ~JS_FLAGS_SIDE_PARALLEL) == 0 && !(b->tune.cross_checks & JSNOOP_XC_SIDE_EXACT);
    if (b->side_mode.size() != b->imgs.size()) { b->side_mode.assign(b->imgs.size(), 0); b->side_anoms.assign(b->imgs.size(), std::vector<uint32_t>()); }
    b->side_anoms[i].clear();
    if (parallel) {
        const uint32_t usn = b->h_us_base[i + 1] - b->h_us_base[i];
        uint32_t *mcu_pos = nullptr, *us_out = nullptr;
        if (js_side_scratch(b, i, &mcu_pos, &us_out)) return -1;
        uint32_t* anoms = us_out + (((size_t)usn * 256 + 64 + 15) & ~(size_t)15);
        // (a clean image's pass may have been launched behind the decode already, js_side_prelaunch)
        const bool pre = i < b->side_pre.size() && b->side_pre[i] == 1 && b->host_flags[i] == 0;
        if (!pre && js_side_parallel_enqueue(b, i, b->host_flags[i] != 0)) return -1;
        if (i < b->side_pre.size()) b->side_pre[i] = 0;
        if (b->host_flags[i]) {
            // overflow records -> bookkeeping.  Not representable here (more records than the list holds; a block that ends on the last bit of
            // a restart interval, where the reader's position array shows a stale slot): the mirror's side-only pass below.
            std::vector<uint32_t> rec(4 + 4 * (size_t)JS_ANOM_MAX);
            if (b->d2h_staged(rec.data(), anoms, rec.size() * 4)) return -1;
            const uint32_t cnt = rec[0];
            bool ok = cnt <= JS_ANOM_MAX;
            std::vector<std::pair<uint32_t, uint32_t>> order;                  // (block, record)
            for (uint32_t k = 0; ok && k < cnt; k++) order.emplace_back(rec[4 + 4 * k], k);
            std::sort(order.begin(), order.end());
            uint32_t sdw[16];
            if (ok && b->d2h_staged(sdw, b->dev.side + im.side_off, sizeof sdw)) return -1;
            // An overflow within the reader's look-ahead (up to four bytes, :1292-1323) of the end of its restart interval / of the scan: the marker behind it has been MET
            // before that symbol is decoded -- "Expected RST marker index ..." / "Scan Data encountered marker" come first, and the latter may use up the warning budget
            // (tools/fuzz_damaged_log.py seed 303 case 2893, seeds 401 / 405).  Which is logged first is the exact reader's to say: the chunked pass below.
            for (uint32_t k = 0; ok && k < cnt; k++) if ((uint64_t)rec[4 + 4 * k + 1] + 40u >= (uint64_t)sdw[10] * 8u) ok = false;
            if (ok && sdw[11] > 1) {                                           // restart intervals: the same at every interval's end; and block ends that sit on an interval boundary
                const uint32_t nseg = std::min<uint32_t>(sdw[11], im.seg_cap - 1);
                std::vector<uint32_t> st(nseg + 1);
                if (b->d2h_staged(st.data(), b->dev.seg + im.seg_off, st.size() * 4)) return -1;
                for (auto& o : order) {
                    const uint32_t ps = rec[4 + 4 * o.second + 1], pe = rec[4 + 4 * o.second + 3];
                    if (!(pe & 7u) && std::binary_search(st.begin(), st.end(), pe >> 3)) ok = false;
                    const size_t sg = (size_t)(std::upper_bound(st.begin(), st.end(), ps >> 3) - st.begin());      // first interval start behind the symbol's byte
                    if (sg < st.size() && (uint64_t)ps + 40u >= (uint64_t)st[sg] * 8u) ok = false;
                }
            }
            if (ok) {
                std::vector<uint32_t>& out = b->side_anoms[i];
                for (auto& o : order) for (int q = 0; q < 4; q++) out.push_back(rec[4 + 4 * o.second + q]);
                // status words: scan_bad, and the warning counter the messages share (each overflow: up to two counted messages, then the
                // markers the look-ahead meets at the end of the scan, which the side pass counted from zero)
                const uint32_t emax = im.err_max, w = std::min<uint32_t>(emax, std::min<uint32_t>(2 * cnt, emax) + rec[1]);   // rec[1]: the end-of-scan markers alone (k_side_maps)
                // (scan_bad is cleared by every restart, DecodeRestartScanBuf :4038-4075: it stays set only for an overflow behind the last marker)
                uint32_t one = sdw[0];
                if (cnt) {
                    uint32_t m_last = 0;
                    if (sdw[11] > 1) {
                        std::vector<uint8_t> rf(nmcu);
                        if (b->d2h_staged(rf.data(), b->dev.mcu_rst + im.mcu_off, nmcu)) return -1;
                        for (uint32_t m = 0; m < nmcu; m++) if (rf[m]) m_last = m;
                    }
                    if (order.back().first / im.blk_per_mcu >= m_last) one = 1u;
                }
                HIP_TRY(hipMemcpyAsync(b->dev.side + im.side_off + 0, &one, 4, hipMemcpyHostToDevice, b->stream));
                HIP_TRY(hipMemcpyAsync(b->dev.side + im.side_off + 6, &w, 4, hipMemcpyHostToDevice, b->stream));
                HIP_TRY(hipStreamSynchronize(b->stream));
            } else parallel = false;
        }
    }
    if (b->side_events.size() != b->imgs.size()) b->side_events.assign(b->imgs.size(), std::vector<uint32_t>());
    b->side_events[i].clear();
    bool chunked = false;
    if (!parallel && i < b->side_chunk_ok.size() && b->side_chunk_ok[i] && !(b->tune.cross_checks & JSNOOP_XC_SIDE_EXACT)) {
        const int rc = b->side_chunk_ok[i] == 4 ? js_side_via_helper(b, i) : js_side_chunked(b, i);
        if (rc < 0) return -1;
        chunked = rc == 0;
    }
    if (parallel) b->side_mode[i] = 1;
    else if (chunked) b->side_mode[i] = 3;
    else {
        b->side_mode[i] = 2;
        HIP_TRY(hipMemsetAsync(b->dev.side + im.side_off, 0, 8 * 4, b->stream));
        HIP_TRY(hipMemsetAsync(b->dev.side + im.side_off + JS_SIDE_HISTO, 0, (words - JS_SIDE_HISTO) * 4, b->stream));
        // (an image of the parallel path whose flags are bookkeeping only: the mirror reader's messages are part of that bookkeeping)
        const bool with_events = b->event_words && i < b->host_path.size() && b->host_path[i] == 1;
        if (with_events) HIP_TRY(hipMemsetAsync(b->dev.events + im.ev_off, 0, 4, b->stream));
        HIP_TRY(hipMemcpyAsync(b->dev.sel, &i, 4, hipMemcpyHostToDevice, b->stream));
        js_launch_entropy_exact(b->stream, b->dev.imgs, b->dev.sel, 1, b->dev.tables, b->dev.raw, b->dev.coef, b->dev.dccum, b->dev.side, 1, with_events ? b->dev.events : nullptr);
    }
    HIP_TRY(hipStreamSynchronize(b->stream));
    HIP_TRY(hipGetLastError());
    if (b->side_done.size() != b->imgs.size()) b->side_done.assign(b->imgs.size(), 0);
    b->side_done[i] = 1;
    return 0;
}
