// jsnoop_kernels.hip -- hand-written CDNA4 (gfx950) kernels of the JPEGsnoop scan-decode path.
//
// Stage map (reference function -> kernel), see DESIGN.md for rooflines and byte counts:
//   BuffAddByte byte rules (ImgDecode.cpp:1386-1573)          -> k_unstuff_*       (parallel path)
//   ReadScanVal / DecodeScanComp (:1072-1286, :1604-1835)      -> k_sync_*, k_write  (parallel path)
//                                                              -> k_entropy_exact    (sequential mirror)
//   DC predictors (:3280, :3355, :3386; reset :2693)            -> k_dc_scan
//   DecodeIdctCalcFloat + SetFullRes + CalcChannelPreviewFull
//   + ConvertYCCtoRGBFastFloat (:2372, :2468, :4619, :4086)     -> k_idct_color
//
// Numerics contract (SURVEY.md findings 1-2): every fp32 operation that the reference
// performs is issued here as a separately rounded IEEE operation (__fmul_rn/__fadd_rn/
// __fsub_rn/__fdiv_rn), in the reference's order; the file is compiled with
// -ffp-contract=off so nothing is fused.  No MFMA: the 8x8 IDCT must keep the scalar
// accumulation order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "jsnoop_types.h"
#include "jsnoop_launch.h"

#define WAVE 64
// wave vote on a bool: the HIP wrapper __ballot(int) makes the compiler materialise the predicate as 0 / 1 and compare it again (two
// half-rate VALU instructions per vote in the entropy loops); the builtin takes the lane mask as it stands
#define WBALLOT(x) __builtin_amdgcn_ballot_w64(static_cast<bool>(x))

__device__ __constant__ uint8_t c_zigzag[64] = {       // ITU-T T.81 Figure A.6 (General.cpp:257-267)
     0, 1, 8,16, 9, 2, 3,10, 17,24,32,25,18,11, 4, 5,
    12,19,26,33,40,48,41,34, 27,20,13, 6, 7,14,21,28,
    35,42,49,56,57,50,43,36, 29,22,15,23,30,37,44,51,
    58,59,52,45,38,31,39,46, 53,60,61,54,47,55,62,63 };

// =====================================================================================
//  Sequential exact-mirror entropy kernel: one lane == one image.
//  Mirrors the reference's 32-bit MSB-first bit register byte for byte, including its
//  behaviour on malformed streams (FF FF, stray markers, bad codes, out-of-place RSTn),
//  so that streams the parallel path refuses still come out reference-exact on device.
// =====================================================================================
enum { RSV_OK, RSV_EOB, RSV_UNDERFLOW, RSV_RST_TERM };
enum { SB_OK, SB_BADMARK, SB_RST };

struct ExactReader {
    const uint8_t* file; uint32_t flen;
    const JsTableSet* ts;
    uint32_t* histo;
    uint32_t buff, vacant, ptr, ptr_first;
    uint32_t pos0, pos1, pos2, pos3, err0, err1, err2, err3, latch, num, align;
    uint32_t scan_end, scan_bad, cur_err, restart_read;
    uint32_t rst_count, rst_last, rst_expect, mcus_left, rst_interval, warn_bad, err_max;
    uint32_t used1, used2, precision, rst_handled;
    uint32_t warn_marker;                                      // of warn_bad: the "Scan Data encountered marker" messages alone
    uint32_t* ev; uint32_t ev_cap, ev_only;                    // event log (JS_EV_*): nullptr = off; ev_only != 0: record just that kind
    uint32_t ev_end;                                           // the reader covers only the END of the scan (k_side_maps): an RSTn its look-ahead meets is recorded with the
                                                               // expected index left open (~0) -- the host, which has followed the markers up to there, fills it in or drops the record
    uint64_t win; uint32_t win_at;                             // 8 file bytes in registers (file images are 16-byte aligned and zero padded)
    const uint32_t* fast;                                      // m_anDhtLookupfast of the six slots, [6][1 << JS_FAST_BITS] (LDS copy in k_entropy_exact)
    const uint32_t* meta;                                      // [0..5] m_anDhtLookupSize per slot, [6..11] DHT destination id per slot
    const uint16_t* q;                                         // quantiser [3][64], zig-zag order
    const uint8_t* zz;                                         // zig-zag index -> natural index
};

// (Everything below is force-inlined into the kernel: a real call takes the reader by reference, which puts ALL of its state into private memory -- a
//  scratch round trip per field access, several per symbol: the un-inlined block decoder cost the mirror 2.6 us per symbol, profiles/r05_experiments.txt.)
// One line (group) of what the reference writes to its log while decoding; formatted on the host (jsnoop_report.cpp).
__device__ __forceinline__ void ex_event(ExactReader& r, uint32_t kind, uint32_t a0 = 0, uint32_t a1 = 0, uint32_t a2 = 0, uint32_t a3 = 0, uint32_t a4 = 0)
{
    if (!r.ev || (r.ev_only && r.ev_only != kind && !(r.ev_end && kind == JS_EV_RST_INDEX))) return;
    const uint32_t n = r.ev[0];
    if (n < r.ev_cap) { uint32_t* e = r.ev + 1 + (size_t)n * JS_EV_WORDS; e[0] = kind; e[1] = a0; e[2] = a1; e[3] = a2; e[4] = a3; e[5] = a4; }
    r.ev[0] = n + 1;
}

__device__ __forceinline__ uint32_t ex_byte(ExactReader& r, uint32_t off)          // CwindowBuf::Buf: 0 past the end of the file
{
    if (off >= r.flen) return 0u;
    const uint32_t a = off & ~7u;
    if (a != r.win_at) { r.win = *reinterpret_cast<const uint64_t*>(r.file + a); r.win_at = a; }   // one aligned load per 8 bytes instead of one per byte
    return (uint32_t)(r.win >> ((off & 7u) * 8)) & 255u;
}

__device__ __forceinline__ void ex_restart_scan_buf(ExactReader& r, uint32_t file_pos, bool restart)   // DecodeRestartScanBuf :4038-4075
{
    r.scan_end = 0; r.scan_bad = 0; r.buff = 0; r.ptr = file_pos;
    if (!restart) r.ptr_first = file_pos;
    r.align = 0; r.pos0 = r.pos1 = r.pos2 = r.pos3 = 0; r.err0 = r.err1 = r.err2 = r.err3 = SB_OK;
    r.latch = SB_OK; r.num = 0; r.vacant = 32; r.cur_err = 0; r.restart_read = 0; r.mcus_left = r.rst_interval;
}
__device__ __forceinline__ void ex_consume(ExactReader& r, uint32_t nbits)                              // ScanBuffConsume :921-955
{
    r.buff = nbits >= 32 ? 0u : r.buff << nbits; r.vacant += nbits;
    uint32_t nbytes = (r.align + nbits) >> 3;
    for (uint32_t i = 0; i < nbytes; i++) {
        r.pos0 = r.pos1; r.pos1 = r.pos2; r.pos2 = r.pos3;
        r.err0 = r.err1; r.err1 = r.err2; r.err2 = r.err3; r.err3 = SB_OK;
        if (r.err0 != SB_OK) r.latch = r.err0;
        r.num--;
    }
    r.align = (r.align + nbits) & 7;
}
__device__ __forceinline__ void ex_add(ExactReader& r, uint32_t byte, uint32_t ptr, uint32_t e)         // ScanBuffAdd(Err) :974-1004
{
    r.buff += byte << (r.vacant - 8); r.vacant -= 8;
    if (r.num < 4) {
        switch (r.num) { case 0: r.err0 = SB_OK; r.pos0 = ptr; break; case 1: r.err1 = SB_OK; r.pos1 = ptr; break;
                         case 2: r.err2 = SB_OK; r.pos2 = ptr; break; default: r.err3 = SB_OK; r.pos3 = ptr; break; }
        r.num++;
    }
    if (e != SB_OK) switch ((r.num - 1) & 3) { case 0: r.err0 = e; break; case 1: r.err1 = e; break; case 2: r.err2 = e; break; default: r.err3 = e; break; }
}
__device__ __forceinline__ void ex_add_byte(ExactReader& r)                                             // BuffAddByte :1386-1573
{
    if (r.restart_read) return;
    uint32_t b0 = ex_byte(r, r.ptr), b1 = ex_byte(r, r.ptr + 1);
    if (b0 == 0xFF && b1 >= 0xD0 && b1 <= 0xD7) {
        r.rst_count++; r.rst_last = b1 - 0xD0;
        if (r.ev_end) ex_event(r, JS_EV_RST_INDEX, 0xFFFFFFFFu, r.rst_last, r.ptr);
        else if (r.rst_last != r.rst_expect) ex_event(r, JS_EV_RST_INDEX, r.rst_expect, r.rst_last, r.ptr);       // :1416-1423
        r.rst_expect = (r.rst_last + 1) & 7; r.restart_read = 1; return;
    }
    if (b0 == 0xFF && b1 == 0x00)      { ex_add(r, b0, r.ptr, SB_OK); r.ptr += 2; }
    else if (b0 == 0xFF && b1 == 0xFF) { ex_add(r, b0, r.ptr, SB_OK); r.ptr += 1; }
    else if (b0 == 0xFF)               { if (r.warn_bad < r.err_max) { ex_event(r, JS_EV_MARKER, b1, r.ptr); r.warn_bad++; r.warn_marker++; } ex_add(r, b0, r.ptr, SB_BADMARK); r.ptr += 1; }
    else                               { ex_add(r, b0, r.ptr, SB_OK); r.ptr += 1; }
}
__device__ __forceinline__ void ex_topup(ExactReader& r)                                                // BuffTopup :1292-1323
{
    bool done = r.vacant < 8 || r.scan_end;
    while (!done) {
        ex_add_byte(r);
        if (r.restart_read) done = true;
        if (r.vacant < 8) done = true;
    }
}
__device__ __forceinline__ int ex_read_scan_val(ExactReader& r, uint32_t t, uint32_t& zrl, int32_t& val)  // ReadScanVal :1072-1286
{
    uint32_t code = JS_CODE_UNUSED, ind = 0; bool done = false, found = false;
    r.used1 = r.used2 = 0; zrl = 0; val = 0;
    if (r.vacant == 32 && r.restart_read) return RSV_RST_TERM;
    if (r.vacant >= 32) { if (r.warn_bad < r.err_max) { ex_event(r, JS_EV_OVERREAD_BEFORE, r.pos0, r.align); r.warn_bad++; } r.scan_end = 1; r.scan_bad = 1; return RSV_UNDERFLOW; }
    ex_topup(r);
    if ((32 - r.vacant) >= JS_FAST_BITS) {
        uint32_t f = r.fast[t * (1u << JS_FAST_BITS) + (r.buff >> (32 - JS_FAST_BITS))];
        if (f != JS_CODE_UNUSED) { r.used1 += f >> 8; code = f & 0xFF; done = true; found = true; }
    }
    const uint32_t size = r.meta[t];
    if (!done && r.ts->lut_ok) {
        // The linear search of :1145-1164 (first entry of the code list whose bits match and whose length fits what the register holds) through the
        // two-level table of the parallel path: for a canonical prefix code -- lut_ok says the list is one -- at most ONE entry matches the register
        // (zero padding included), so "the first that matches and fits" is "the one that matches, if it fits".  The list search walks up to 162 entries
        // of global memory for every code longer than nine bits: 3 % of the symbols, half of the mirror's time.
        uint32_t e = r.ts->lut1[r.ts->slot_row[t]][r.buff >> (32 - JS_L1_BITS)];
        if (e & 0x8000u) { const uint32_t nbx = (e >> 12) & 7u; e = r.ts->lut2[(e & 0xFFFu) + ((r.buff >> (32 - JS_L1_BITS - nbx)) & ((1u << nbx) - 1u))]; }
        const uint32_t bl = (e >> 8) & 31u;
        if (bl != 0u && bl <= 32 - r.vacant) { code = e & 255u; r.used1 += bl; found = true; }
        done = true;
    }
    while (!done) {
        if ((r.buff & r.ts->mask[t][ind]) == r.ts->bits[t][ind]) {
            uint32_t bl = r.ts->bitlen[t][ind];
            if (bl <= 32 - r.vacant) { code = r.ts->code[t][ind]; r.used1 += bl; done = true; found = true; }
        }
        ind++;
        if (ind >= size) done = true;
    }
    if (!found) {
        if (r.restart_read) return RSV_RST_TERM;
        r.used1 = 1; code = JS_CODE_UNUSED;
    }
    if (r.used1 < 17) r.histo[((t & 1) * 4 + r.meta[6 + t]) * 17 + r.used1]++;
    ex_consume(r, r.used1);
    if (r.vacant > 32) { ex_event(r, JS_EV_OVERREAD_CODE, r.pos0, r.align); r.scan_end = 1; r.scan_bad = 1; return RSV_UNDERFLOW; }
    ex_topup(r);
    if (code != JS_CODE_UNUSED) {
        zrl = (code & 0xF0) >> 4; r.used2 = code & 0x0F;
        if (zrl == 0 && r.used2 == 0) return RSV_EOB;
        if (r.used2 == 0) { val = 0; return RSV_OK; }
        uint32_t v = r.buff >> (32 - r.used2);
        val = v >= (1u << (r.used2 - 1)) ? (int32_t)v : (int32_t)(v - ((1u << r.used2) - 1));   // HuffmanDc2Signed :859
        if (r.precision >= 8) val /= (int32_t)(1u << ((r.precision - 8) & 31));
        ex_consume(r, r.used2);
        if (r.vacant > 32) { ex_event(r, JS_EV_OVERREAD_BITS, r.pos0, r.align); r.scan_end = 1; r.scan_bad = 1; return RSV_UNDERFLOW; }
        return RSV_OK;
    }
    if (r.warn_bad < r.err_max) { ex_event(r, JS_EV_CANT_FIND, r.pos0, r.align, r.meta[6 + t], r.buff); r.warn_bad++; }   // :1266-1277
    r.scan_bad = 1;
    return RSV_UNDERFLOW;
}

// DecodeScanComp :1604-1835 for one 8x8 block; coefficients go straight to HBM (natural order).
// Returns the dequantised value stored at natural index 0 (what the caller adds to the DC predictor).
__device__ __forceinline__ int16_t ex_decode_block(ExactReader& r, uint32_t comp, uint32_t decode_ac, int16_t* __restrict__ out,
                                   int16_t& dc_y, int16_t& dc_cb, int16_t& dc_cr)
{
    const uint32_t tdc = (comp - 1) * 2, tac = tdc + 1;
    const uint16_t* q = r.q + (comp - 1) * 64;
    uint32_t zrl, ncoef = 0; int32_t val; bool done = false, is_dc = true, failed = false; int16_t dct0 = 0;
    while (!done) {
        ex_topup(r);
        const uint32_t saved_err = r.latch, saved_pos = r.pos0, saved_align = r.align;
        int rv = ex_read_scan_val(r, is_dc ? tdc : tac, zrl, val);
        if (rv == RSV_RST_TERM) {                               // marker-driven restart :1644-1680
            dc_y = dc_cb = dc_cr = 0; r.rst_handled++;      // DecodeRestartDcState :2693
            r.ptr += 2; ex_restart_scan_buf(r, r.ptr, true); r.restart_read = 0;
            ex_topup(r);
            rv = ex_read_scan_val(r, is_dc ? tdc : tac, zrl, val);
        }
        if (saved_err == SB_BADMARK) { r.cur_err = 1; r.scan_bad = 1; if (r.warn_bad < r.err_max) { ex_event(r, JS_EV_BAD_MARKER, saved_pos, saved_align); r.warn_bad++; } r.latch = SB_OK; }
        int16_t v16 = (int16_t)(val & 0xFFFF);
        bool store = false;
        if (rv == RSV_OK)       { if (is_dc) { store = true; is_dc = false; } else store = decode_ac != 0; }
        else if (rv == RSV_EOB) { if (is_dc) { store = true; is_dc = false; } else done = true; }
        else if (rv == RSV_UNDERFLOW) { if (r.warn_bad < r.err_max) { ex_event(r, JS_EV_BAD_HUFF, saved_pos, saved_align); r.warn_bad++; } r.cur_err = 1; failed = true; break; }
        if (store) {                                            // DecodeIdctSet :2270-2303
            uint32_t ind = ncoef + zrl;
            if (ind < 64) {
                int16_t dq = (int16_t)((int32_t)v16 * (int32_t)q[ind]);
                uint32_t nat = r.zz[ind];
                if (nat == 0) dct0 = dq; else out[nat] = dq;
            }
        }
        ncoef += 1 + zrl;
        if (ncoef == 64) done = true;
        else if (ncoef > 64) { if (r.warn_bad < r.err_max) { ex_event(r, JS_EV_NUMCOEF, saved_pos, saved_align, ncoef); r.warn_bad++; } r.cur_err = 1; r.scan_bad = 1; done = true; }
    }
    if (failed) for (int i = 1; i < 64; i++) out[i] = 0;         // IDCT skipped (:1737-1757): AC contributes 0.0f
    out[0] = dct0;
    return dct0;
}

// Tail mode (tail.flags != nullptr; one image, sel[0]): the parallel path has decoded the image and vouches for every block before
// flags[2 * img + 1] (the first block at which it saw something other than a coefficient-index overflow).  The mirror reader takes over
// at the top of the MCU that holds that block -- seeded from the MCU's bit position (side walk: mcu_pos) and the cumulative DC values the
// parallel path left for the MCU before it -- and decodes from there to the last MCU, writing coefficients and cumulative DC only.  At
// an MCU top the part of the reader's state that decides WHAT is decoded (bit register after its refill, next byte to load, "restart
// marker seen") is a function of the bit position alone; what is not (slots of its position array, a pending bad-marker latch) only
// shows in the MCU file map and the messages, which such an image gets from a side-only pass of the whole mirror on request.
struct ExactTail { const uint32_t* flags; const uint32_t* seg_tab; const uint8_t* mcu_rst; const uint32_t* mcu_pos; const uint32_t* us_out; uint32_t us_threads; };
__device__ uint32_t raw_of_compacted(const JsImage& im, const uint8_t* __restrict__ raw, const uint32_t* __restrict__ us_out, uint32_t nthreads, uint32_t u);

__global__ void __launch_bounds__(64) k_entropy_exact(const JsImage* __restrict__ imgs, const uint32_t* __restrict__ sel, uint32_t nsel,
                                                      const JsTableSet* __restrict__ tables, const uint8_t* __restrict__ raw,
                                                      int16_t* __restrict__ coef, int16_t* __restrict__ dccum, uint32_t* __restrict__ side, int side_only,
                                                      uint32_t* __restrict__ events, ExactTail tail)
{
    // side_only: recompute only the decoder's side outputs (MCU file map, block-DC maps, code-length
    // histogram, status words) for an image whose pixels came from the parallel path.
    // One workgroup (one wave) per image, lane 0 decodes: the mirror is a sequential program whose every step waits for the
    // previous one, so what counts is latency -- the 9-bit look-up tables and the histogram live in LDS, and lanes of
    // different images do not share a wave (their branches would serialise).
    __shared__ uint32_t s_fast[6 * (1 << JS_FAST_BITS)]; __shared__ uint32_t s_histo[2 * 4 * 17];
    __shared__ uint32_t s_meta[12]; __shared__ uint16_t s_q[3 * 64]; __shared__ uint8_t s_zz[64];
    // (indexed at run time: in LDS, not in private memory -- every access of a private array is a trip to scratch, and the mirror is a chain of
    //  dependent steps on one lane)
    __shared__ int16_t scratch[64]; __shared__ int16_t s_css[3][16];
    const uint32_t j = blockIdx.x;
    if (j >= nsel) return;
    const JsImage& im = imgs[sel ? sel[j] : j];
    {
        const uint32_t* src = &tables[im.tableset].fast[0][0];
        for (uint32_t i = threadIdx.x; i < 6 * (1u << JS_FAST_BITS); i += blockDim.x) s_fast[i] = src[i];
        for (uint32_t i = threadIdx.x; i < 2 * 4 * 17; i += blockDim.x) s_histo[i] = 0;
        const JsTableSet& tset = tables[im.tableset];
        if (threadIdx.x < 6) { s_meta[threadIdx.x] = tset.size[threadIdx.x]; s_meta[6 + threadIdx.x] = tset.dest_id[threadIdx.x]; }
        for (uint32_t i = threadIdx.x; i < 3 * 64; i += blockDim.x) s_q[i] = (&tset.qzz[0][0])[i];
        if (threadIdx.x < 64) s_zz[threadIdx.x] = c_zigzag[threadIdx.x];
    }
    __syncthreads();
    uint32_t* sd = side + im.side_off;
    const uint32_t nmcu = im.mcu_xmax * im.mcu_ymax, nblk = im.blk_xmax * im.blk_ymax;
    __shared__ uint32_t s_m0;
    if (tail.flags) {
        // Tail mode: the MCU the mirror takes over at, and everything from there on emptied first -- the mirror stores what it decodes, and
        // MCUs it never reaches (the scan-stop logic after an overread, :3623-3625) must read as the cleared arrays of the reference.
        if (threadIdx.x == 0) {
            uint32_t m0 = min((~tail.flags[2u * sel[j] + 1u] >> 4) / im.blk_per_mcu, nmcu);   // (the word holds the complement of block << 4 | kind)
            while (m0 && tail.mcu_pos[m0] == 0u) m0--;             // (an MCU top the side walk did not reach: fall back to an earlier one)
            s_m0 = m0;
        }
        __syncthreads();
        const size_t b0 = (size_t)s_m0 * im.blk_per_mcu, b1 = im.total_blocks;
        uint4* zc = reinterpret_cast<uint4*>(coef + (im.coef_off + b0) * 64); const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
        for (size_t q = threadIdx.x; q < (b1 - b0) * 8; q += blockDim.x) zc[q] = zero;
        int16_t* zd = dccum + im.coef_off;
        for (size_t q = b0 + threadIdx.x; q < b1; q += blockDim.x) zd[q] = 0;
        __syncthreads();
    }
    // Periodic tail (tail mode): once the reader has run out of file -- every byte it will ever load is the zero CwindowBuf::Buf returns
    // past the end (WindowBuf.cpp:639), its register holds zero bits only, nothing is pending -- every further MCU decodes to the same
    // blocks and moves the DC predictors by the same amounts.  The mirror decodes ONE such MCU; the lanes of the wave replicate it over
    // the rest of the image (a truncated file: the common case of a carved one) instead of 0.6 ms of sequential decode per MCU.
    __shared__ uint32_t s_fill_from;                               // MCU index of the template, 0xFFFFFFFF: none
    __shared__ int s_dc_after[3], s_dc_step[3]; __shared__ int s_dc_in[JS_MAX_BLK_PER_MCU];
    __shared__ __attribute__((aligned(16))) int16_t s_tpl[JS_MAX_BLK_PER_MCU * 64];
    if (threadIdx.x == 0) s_fill_from = 0xFFFFFFFFu;
    auto serial = [&]() {
    uint32_t* mcu_map = sd + JS_SIDE_MCUMAP;
    int16_t* bdc[3]; bdc[0] = (int16_t*)(mcu_map + nmcu); bdc[1] = bdc[0] + 2 * ((nblk + 1) / 2); bdc[2] = bdc[1] + 2 * ((nblk + 1) / 2);

    ExactReader r;
    r.file = raw + im.file_off; r.flen = im.file_len; r.ts = tables + im.tableset; r.histo = s_histo; r.fast = s_fast; r.meta = s_meta; r.q = s_q; r.zz = s_zz; r.win_at = 0xFFFFFFFFu; r.win = 0;
    r.rst_interval = im.rst_interval; r.precision = im.precision; r.err_max = im.err_max; r.warn_bad = 0; r.warn_marker = 0;
    r.rst_count = 0; r.rst_last = 0; r.rst_expect = 0; r.rst_handled = 0;
    r.ev = (im.ev_cap && events) ? events + im.ev_off : nullptr; r.ev_cap = im.ev_cap; r.ev_only = 0; r.ev_end = 0;       // (side-only passes log when the caller hands the event area over)
    ex_restart_scan_buf(r, im.scan_start, false);
    int16_t dc_y = 0, dc_cb = 0, dc_cr = 0;
    int16_t (*css)[16] = s_css;
    for (int c = 0; c < 3; c++) for (int i = 0; i < 16; i++) css[c][i] = 0;
    ex_topup(r);
    uint32_t num_pixels = 0;
    int16_t* cbase = coef + im.coef_off * 64;
    int16_t* dbase = dccum + im.coef_off;
    uint32_t m_first = 0;
    const bool tail_mode = tail.flags != nullptr;
    if (tail_mode) {
        const uint32_t nb = im.blk_per_mcu;
        const uint32_t m0 = s_m0;
        m_first = m0;
        if (m0) {
            const uint32_t p = tail.mcu_pos[m0];
            r.ev = nullptr;
            // The reader is seeded at the byte that holds the LAST bit consumed before the MCU top and walks the remaining 1..8 bits of it:
            // when the top sits on a restart-interval boundary that byte is the last one in front of the marker(s), the refill meets the
            // marker, and the reader arrives with an empty register and "restart marker seen" -- it handles the restart (or two markers
            // back to back, whose second one the reference meets inside the retry of :1644-1680) exactly as the reference does.
            const uint32_t ub = (p - 1u) >> 3;
            ex_restart_scan_buf(r, raw_of_compacted(im, raw, tail.us_out, tail.us_threads, ub), true); ex_topup(r); ex_consume(r, p - 8u * ub);
            const uint32_t n1 = im.samp_h[1] * im.samp_v[1], n2 = im.ncomp == 3 ? n1 + im.samp_h[2] * im.samp_v[2] : nb;
            const int16_t* dprev = dbase + (size_t)(m0 - 1) * nb;   // cumulative DC of the last block of each component in the MCU before
            dc_y = dprev[n1 - 1]; if (im.ncomp == 3) { dc_cb = dprev[n2 - 1]; dc_cr = dprev[nb - 1]; }
            // ... unless a restart was followed INSIDE that MCU behind the component's last block (its mark: predictors cleared in front of block j): the
            // component's predictor is then the zero the restart left, not the sum its last block had reached before it
            const uint32_t rj = tail.mcu_rst[im.mcu_off + m0 - 1u] & 63u;
            if (rj) { if (n1 < rj) dc_y = 0; if (im.ncomp == 3 && n2 < rj) dc_cb = 0; }
            r.ptr_first = im.scan_start;
        }
    }

    for (uint32_t my = m_first / im.mcu_xmax; my < im.mcu_ymax; my++) {
        bool stop = false;
        for (uint32_t mx = (my == m_first / im.mcu_xmax ? m_first % im.mcu_xmax : 0u); mx < im.mcu_xmax && !stop; mx++) {
            const uint32_t mi = my * im.mcu_xmax + mx;
            if (im.rst_en && r.mcus_left == 0 && !r.restart_read) ex_event(r, JS_EV_RST_NOT_DETECTED, r.pos0, r.align);   // :3180-3200
            if (!tail_mode) mcu_map[mi] = (r.pos0 << 4) + r.align;        // PackFileOffset :5104
            const bool zero_state = tail_mode && mi + 1 < nmcu && r.buff == 0u && r.ptr >= r.flen && !r.restart_read && !r.scan_end && r.latch == SB_OK &&
                                    r.err0 == SB_OK && r.err1 == SB_OK && r.err2 == SB_OK && r.err3 == SB_OK;
            const int dc_in0 = dc_y, dc_in1 = dc_cb, dc_in2 = dc_cr;
            for (uint32_t c = 0; c < im.blk_per_mcu; c++) {
                const uint32_t comp = im.blk_comp[c];
                const size_t b = (size_t)mi * im.blk_per_mcu + c;
                const uint32_t rst_before = r.rst_handled;
                int16_t d0 = ex_decode_block(r, comp, im.decode_ac, side_only ? scratch : cbase + b * 64, dc_y, dc_cb, dc_cr);
                if (r.rst_handled != rst_before) for (int cc = 0; cc < 3; cc++) for (int i = 0; i < 16; i++) css[cc][i] = 0;
                if (r.cur_err) {                                          // CheckScanErrors :2605
                    if (r.warn_bad < r.err_max) { ex_event(r, JS_EV_BAD_SCAN_MCU, mx | (my << 16), comp | (im.blk_ch[c] << 8) | (im.blk_cv[c] << 16), r.pos0, r.align); r.warn_bad++; }
                    r.cur_err = 0; }
                int16_t* acc = comp == 1 ? &dc_y : comp == 2 ? &dc_cb : &dc_cr;
                *acc = (int16_t)(*acc + d0);
                css[comp - 1][im.blk_cv[c] * 4 + im.blk_ch[c]] = *acc;
                if (!side_only) dbase[b] = *acc;
                if (comp == 1) num_pixels += 64;
            }
            if (!tail_mode) {   // per-block cumulative DC maps :3524-3608 (sequential overwrite order preserved)
                uint32_t lin = (my * im.expand_v[1]) * im.blk_xmax + mx * im.expand_h[1];
                for (uint32_t cv = 0; cv < im.samp_v[1]; cv++) for (uint32_t ch = 0; ch < im.samp_h[1]; ch++) {
                    uint32_t bi = lin + cv * im.blk_xmax + ch; if (bi < nblk) bdc[0][bi] = css[0][cv * 4 + ch]; }
                if (im.ncomp == 3) for (uint32_t comp = 2; comp <= 3; comp++)
                    for (uint32_t cv = 0; cv < im.samp_v[comp]; cv++) for (uint32_t ch = 0; ch < im.samp_h[comp]; ch++) {
                        uint32_t bi = (my * im.expand_v[comp] + cv) * im.blk_xmax + (mx * im.expand_h[comp] + ch);
                        if (bi < nblk) bdc[comp - 1][bi] = css[comp - 1][cv * 4 + ch]; }
            }
            if (im.rst_en) r.mcus_left--;
            if (r.scan_end && r.scan_bad) stop = true;                    // :3623-3625
            if (zero_state && !stop && r.buff == 0u) {                    // this MCU is the template of all that follow
                const uint32_t nb = im.blk_per_mcu;
                const int din[3] = { dc_in0, dc_in1, dc_in2 };
                s_dc_after[0] = dc_y; s_dc_after[1] = dc_cb; s_dc_after[2] = dc_cr;
                s_dc_step[0] = dc_y - dc_in0; s_dc_step[1] = dc_cb - dc_in1; s_dc_step[2] = dc_cr - dc_in2;
                for (uint32_t c = 0; c < nb; c++) {
                    s_dc_in[c] = (int)dbase[(size_t)mi * nb + c] - din[im.blk_comp[c] - 1];   // the predictor's gain inside the MCU up to and including block c
                    const uint4* src = reinterpret_cast<const uint4*>(cbase + ((size_t)mi * nb + c) * 64); uint4* dst = reinterpret_cast<uint4*>(s_tpl + c * 64);
                    for (int q = 0; q < 8; q++) dst[q] = src[q];
                }
                s_fill_from = mi;
                return;
            }
        }
    }
    if (tail_mode) return;                                        // (status words, histogram and maps: the side-only pass, on request)
    sd[0] = r.scan_bad; sd[1] = r.scan_end; sd[2] = r.rst_count; sd[3] = num_pixels;
    sd[4] = r.pos0; sd[5] = r.align; sd[6] = r.warn_bad; sd[7] = r.ptr_first; if (!side_only) sd[9] = 2;
    for (uint32_t i = 0; i < 2 * 4 * 17; i++) sd[JS_SIDE_HISTO + i] = s_histo[i];
    };
    if (threadIdx.x == 0) serial();
    __syncthreads();
    if (s_fill_from != 0xFFFFFFFFu) {
        const uint32_t nb = im.blk_per_mcu, mt = s_fill_from;
        const size_t rows = (size_t)(nmcu - mt - 1) * nb;            // block rows behind the template MCU
        uint4* dst = reinterpret_cast<uint4*>(coef + (im.coef_off + (size_t)(mt + 1) * nb) * 64);
        const uint4* tpl = reinterpret_cast<const uint4*>(s_tpl);
        for (size_t q = threadIdx.x; q < rows * 8; q += blockDim.x) dst[q] = tpl[((q >> 3) % nb) * 8 + (q & 7)];
        int16_t* dd = dccum + im.coef_off + (size_t)(mt + 1) * nb;
        for (size_t q = threadIdx.x; q < rows; q += blockDim.x) {
            const uint32_t c = (uint32_t)(q % nb), comp0 = im.blk_comp[c] - 1u; const int n = (int)(q / nb);   // n whole MCUs lie between the template and this one
            dd[q] = (int16_t)(s_dc_after[comp0] + n * s_dc_step[comp0] + s_dc_in[c]);
        }
    }
}

// =====================================================================================
//  Back end: sparse fp32 IDCT -> int16 samples -> chroma replication -> fp32 YCbCr->RGB
//  -> bottom-up BGRA DIB.   One workgroup = one strip of G adjacent MCUs of one MCU row.
//  wave = 8x8 block, lane = output sample; the transposed cosine table lives in LDS
//  (lane-contiguous rows => conflict-free ds_read_b32); samples are staged in LDS as
//  full-resolution int16 planes so the DIB rows leave as 16-byte coalesced stores.
// =====================================================================================
#define BK_THREADS 512
#define BK_WAVES (BK_THREADS / 64)
#define BK_CHUNK 6                      // blocks per wave held in registers at a time (one 4:2:0 MCU)

// ConvertYCCtoRGBFastFloat :4086-4139 on clamped Y, Cb, Cr (already int -> float).  The reference divides by 0.587f (IEEE).
// The quotient is formed here as q0 = x * RN(1/0.587f) followed by ONE fused correction step; for every numerator this
// function can produce (y, cb, cr in [-128, 127] after the clamp: 2^24 cases) that is bit-identical to the IEEE quotient --
// checked exhaustively on the device against the oracle's true division (tests/test_gpu_parity.py::test_color_sweep).
// Plain fp32 multiplies and adds issue at full rate on gfx950; their packed forms do not (profiles/r02_instr_rates.txt), so
// the pixels are converted one at a time.  crm = RN(Cr * (2 - 2*0.299f)) and cbm = RN(Cb * (2 - 2*0.114f)) depend on the
// chroma sample only: subsampled images form them once per chroma sample, not per pixel.
#define RB_BIAS (127.5f + 0.000244140625f)                        // see pack_bgr
struct Rgbf { float r, g, b; };                                  // the three channels after "+= 128", before the range cap
__device__ __forceinline__ float chroma_r(float fcr) { const float kr = 0.299f; return __fmul_rn(fcr, 2 - 2 * kr); }   // folded in fp32 exactly as the reference's expression
__device__ __forceinline__ float chroma_b(float fcb) { const float kb = 0.114f; return __fmul_rn(fcb, 2 - 2 * kb); }
// PACK: the result goes to pack_bgr (R and B carry pack_bgr's bias instead of the reference's + 128; G is the reference's)
template <bool PACK>
__device__ __forceinline__ Rgbf ycc_core(float fy, float crm, float cbm)
{
    const float kr = 0.299f, kg = 0.587f, kb = 0.114f, rkg = 1.0f / kg;
    Rgbf o;
    const float r = __fadd_rn(crm, fy), b = __fadd_rn(cbm, fy);
    const float x = __fsub_rn(__fsub_rn(fy, __fmul_rn(kb, b)), __fmul_rn(kr, r));
    const float q0 = __fmul_rn(x, rkg);
    const float g = __builtin_fmaf(__builtin_fmaf(-kg, q0, x), rkg, q0);          // == x / kg, see above
    o.r = __fadd_rn(r, PACK ? RB_BIAS : 128.0f); o.g = __fadd_rn(g, 128.0f); o.b = __fadd_rn(b, PACK ? RB_BIAS : 128.0f);
    return o;
}
// <0 -> 0, >255 -> 255, else truncate (:4128-4136); bytes B,G,R,0 (:4786-4789).  floor() equals the truncation wherever the value
// is not capped to 0 anyway, and v_cvt_pk_u8_f32 (round to nearest) saturates to [0, 255] into the byte it is told to fill.  R and B
// are RN(chroma term + Y): for every (Y, Cb) / (Y, Cr) pair (2 x 65536 cases) the conversion of RN(v + (127.5 + 2^-12)) equals
// floor(RN(v + 128)) under either tie rule -- no floor instruction for them (rb_bias below; tests/test_gpu_parity.py::test_color_sweep is
// the gate, all 2^24 triples against the oracle).  G has arbitrary fractions: it keeps the floor.
__device__ __forceinline__ uint32_t pack_bgr(const Rgbf& c)
{
    uint32_t o = __builtin_amdgcn_cvt_pk_u8_f32(c.b, 0, 0u);
    o = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_floorf(c.g), 1, o);
    return __builtin_amdgcn_cvt_pk_u8_f32(c.r, 2, o);
}
// ... then ChannelExtract :4832-4872 on the capped values (the preview modes other than RGB)
__device__ __forceinline__ uint32_t channel_extract(const Rgbf& c, int y, int cb, int cr, uint32_t mode)
{
    uint32_t R = (uint32_t)(int)__builtin_amdgcn_fmed3f(c.r, 0.0f, 255.0f), G = (uint32_t)(int)__builtin_amdgcn_fmed3f(c.g, 0.0f, 255.0f),
             B = (uint32_t)(int)__builtin_amdgcn_fmed3f(c.b, 0.0f, 255.0f);
    const uint32_t FY = (uint32_t)(y + 128), FCB = (uint32_t)(cb + 128), FCR = (uint32_t)(cr + 128);
    switch (mode) {
    case 2: R = FCR; G = FY; B = FCB; break;      // PREVIEW_YCC
    case 3: G = B = R; break;                     // PREVIEW_R
    case 4: R = B = G; break;                     // PREVIEW_G
    case 5: R = G = B; break;                     // PREVIEW_B
    case 6: R = G = B = FY; break;                // PREVIEW_Y
    case 7: R = G = B = FCB; break;               // PREVIEW_CB
    case 8: R = G = B = FCR; break;               // PREVIEW_CR
    default: break;
    }
    return B | (G << 8) | (R << 16);
}
__device__ __forceinline__ int clamp_s8(int v) { return min(max(v, -128), 127); }
// one pixel from raw int16-range samples (:4096-4104: >> 3, clamp to [-128, 127])
template <bool RGB_ONLY>
__device__ __forceinline__ uint32_t ycc_to_bgra(int py, int pcb, int pcr, uint32_t mode)
{
    const int y = clamp_s8(py >> 3), cb = clamp_s8(pcb >> 3), cr = clamp_s8(pcr >> 3);
    const Rgbf c = ycc_core<RGB_ONLY>((float)y, chroma_r((float)cr), chroma_b((float)cb));
    return RGB_ONLY ? pack_bgr(c) : channel_extract(c, y, cb, cr, mode);
}
__device__ __forceinline__ int s16_lo(uint32_t w) { return (int)(int16_t)w; }
__device__ __forceinline__ int s16_hi(uint32_t w) { return (int)w >> 16; }

// LDS through its own 32-bit addresses: an address kept in a register goes into the ds instruction as it is (no base added per access)
typedef __attribute__((address_space(3))) uint8_t lds_u8_t;
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(lds_u8_t*)p; }
__device__ __forceinline__ uint32_t lds_r32(uint32_t a) { return *(const __attribute__((address_space(3))) uint32_t*)(uintptr_t)a; }
__device__ __forceinline__ uint2 lds_r64(uint32_t a) { const u32x2_t v = *(const __attribute__((address_space(3))) u32x2_t*)(uintptr_t)a; return make_uint2(v.x, v.y); }
__device__ __forceinline__ void lds_w64(uint32_t a, uint32_t x, uint32_t y) { u32x2_t v; v.x = x; v.y = y; *(__attribute__((address_space(3))) u32x2_t*)(uintptr_t)a = v; }

// ---- IDCT term loop ------------------------------------------------------------------------------------------------------
// What an instruction costs on gfx950 (tools/probes/pk_f32_rate.hip, profiles/r02_instr_rates.txt): plain fp32 / integer VOP2
// with VGPR operands issue in 2.2 cycles per wave, anything with a DPP or SGPR operand, every conversion, min/max/med3, left
// shifts and the three-operand integer ops in 4.1, one LDS read in 8.5 cycles of the CU's LDS pipe per SIMD.  A term of the
// sum therefore costs: the table row (one LDS read), the multiply with the coefficient as a DPP row-broadcast operand (4.1)
// and the add (2.2) -- and nothing else: the row's LDS address comes from M0 (ds_read_addtid_b32: M0 + lane * 4), written by
// the scalar unit from row offsets that sit two to a dword (one v_readlane per two terms, one scalar instruction per term).  The cosine table therefore has
// to start at LDS offset 0 (the kernels use dynamic shared memory only and check it).
//
// List of a block (this wave's LDS): the coefficients as fp32 and the LDS offsets of their table rows (16 bits each), both written by
// EVERY lane -- the non-zero AC coefficients at slot = rank (ascending natural order, the reference's summation order), the zero lanes
// behind them (slot 63 - number of zero lanes below: a full permutation of the 64 slots, no predicate, no bank conflict), each with its
// own row: their products are exact +-0 and leave the sum unchanged, so the list is padded to any multiple of four for free and never
// holds a stale entry.  A wave has TWO such lists: the list of block j + 1 is written, and its first reads are issued, before the terms
// of block j run (idct_prep / idct_run) -- the LDS round trips of the list build are off the critical path of the wave.
// Held per lane: the slot offset of a zero lane (63 - lane), the row word it contributes, and the addresses it reads back from: the pair
// of row words (lane q < 32: terms 2q, 2q+1) and the coefficient of term (lane & 15) of a round; the wave's base is wave-uniform.
struct WaveList { uint32_t base /*SGPR*/, z_off, row_w, a_rows, a_ey; };
#define LIST_BUF_BYTES (64 * 4 + 64 * 2)
#define LIST_BYTES 1024                                       // per wave: two one-block lists (general path) or the two halves' lists of the pair form
__device__ __forceinline__ WaveList wave_list(const void* mem, uint32_t lane)
{
    WaveList L; L.base = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr(mem));
    L.z_off = 63u - lane; L.row_w = lane << 8; L.a_rows = L.base + 256u + (lane & 31u) * 4u; L.a_ey = L.base + (lane & 15u) * 4u;
    return L;
}
__device__ __forceinline__ void lds_w32(uint32_t a, uint32_t x) { *(__attribute__((address_space(3))) uint32_t*)(uintptr_t)a = x; }
__device__ __forceinline__ void lds_w16(uint32_t a, uint32_t x) { *(__attribute__((address_space(3))) uint16_t*)(uintptr_t)a = (uint16_t)x; }

#define DPP_BC(I) " row_newbcast:" #I " row_mask:0xf bank_mask:0xf\n\t"
#define T_RD(X) "ds_read_addtid_b32 %[" #X "]\n\t"
#define T_MUL(X, I) "v_mul_f32_dpp %[" #X "], %[ey], %[" #X "]" DPP_BC(I)
#define T_ADD(X) "v_add_f32 %[acc], %[acc], %[" #X "]\n\t"
#define T_RL(S, Q) "v_readlane_b32 %[" #S "], %[rows], " #Q "\n\t"
#define T_M0(OP) OP
// first group of a round: four table reads in flight (M0 needs one wait state before the read that uses it)
#define T_ISSUE(L0, L1, L2, L3, QA, QB)                                                                                          \
    T_RL(sp0, QA) T_M0("s_and_b32 m0, %[sp0], 0xffff\n\t") T_RL(sp1, QB)                                                            \
    T_RD(L0) T_M0("s_lshr_b32 m0, %[sp0], 16\n\t") "s_nop 0\n\t" T_RD(L1)                                                          \
    T_M0("s_and_b32 m0, %[sp1], 0xffff\n\t") "s_nop 0\n\t" T_RD(L2)                                                               \
    T_M0("s_lshr_b32 m0, %[sp1], 16\n\t") "s_nop 0\n\t" T_RD(L3)
// one term of the group in flight retires (multiply with the coefficient as DPP operand, add) while the read of a term of the next
// group is issued: per term one scalar instruction (M0), one LDS read, two vector instructions
#define T_STEP1(X, Y, I, M0OP)                                                                                                   \
    T_M0(M0OP) "s_waitcnt lgkmcnt(3)\n\t" T_MUL(X, I) T_RD(Y) T_ADD(X)
#define T_STEP(X0, X1, X2, X3, Y0, Y1, Y2, Y3, I0, I1, I2, I3, QA, QB)                                                           \
    T_RL(sp0, QA) T_RL(sp1, QB)                                                                                                  \
    T_STEP1(X0, Y0, I0, "s_and_b32 m0, %[sp0], 0xffff\n\t") T_STEP1(X1, Y1, I1, "s_lshr_b32 m0, %[sp0], 16\n\t")                     \
    T_STEP1(X2, Y2, I2, "s_and_b32 m0, %[sp1], 0xffff\n\t") T_STEP1(X3, Y3, I3, "s_lshr_b32 m0, %[sp1], 16\n\t")
#define T_DRAIN1(X, I, CNT) "s_waitcnt lgkmcnt(" #CNT ")\n\t" T_MUL(X, I) T_ADD(X)
#define T_DRAIN(X0, X1, X2, X3, I0, I1, I2, I3) T_DRAIN1(X0, I0, 3) T_DRAIN1(X1, I1, 2) T_DRAIN1(X2, I2, 1) T_DRAIN1(X3, I3, 0)
#define T_EXIT_IF_LE(K, LABEL) "s_cmp_le_u32 %[nl], " #K "\n\t" "s_cbranch_scc1 " LABEL "%=\n\t"
// sixteen terms (one round): rows of the round in dwords Q0 .. Q0+7 of `rows`, coefficients in lanes 0 .. 15 of every row of `ey`,
// nl = terms left including this round's (> 0; the list is padded to a multiple of four)
#define IDCT_ROUND(Q0, Q1, Q2, Q3, Q4, Q5, Q6, Q7)                                                                               \
    asm volatile(                                                                                                                \
        T_ISSUE(a0, a1, a2, a3, Q0, Q1)                                                                                          \
        T_EXIT_IF_LE(4, ".Lda")                                                                                                  \
        T_STEP(a0, a1, a2, a3, b0, b1, b2, b3, 0, 1, 2, 3, Q2, Q3)                                                               \
        T_EXIT_IF_LE(8, ".Ldb")                                                                                                  \
        T_STEP(b0, b1, b2, b3, a0, a1, a2, a3, 4, 5, 6, 7, Q4, Q5)                                                               \
        T_EXIT_IF_LE(12, ".Ldc")                                                                                                 \
        T_STEP(a0, a1, a2, a3, b0, b1, b2, b3, 8, 9, 10, 11, Q6, Q7)                                                             \
        T_DRAIN(b0, b1, b2, b3, 12, 13, 14, 15) "s_branch .Lend%=\n\t"                                                           \
        ".Lda%=:\n\t" T_DRAIN(a0, a1, a2, a3, 0, 1, 2, 3) "s_branch .Lend%=\n\t"                                                 \
        ".Ldb%=:\n\t" T_DRAIN(b0, b1, b2, b3, 4, 5, 6, 7) "s_branch .Lend%=\n\t"                                                 \
        ".Ldc%=:\n\t" T_DRAIN(a0, a1, a2, a3, 8, 9, 10, 11)                                                                      \
        ".Lend%=:"                                                                                                               \
        : [acc] "+v"(acc), [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [b0] "=&v"(b0), [b1] "=&v"(b1), [b2] "=&v"(b2), [b3] "=&v"(b3), \
          [sp0] "=&s"(sp0), [sp1] "=&s"(sp1)                                                                                     \
        : [rows] "v"(rows2), [ey] "v"(ey), [nl] "s"(nl) : "scc")

// DecodeIdctCalcFloat(64) :2372-2392 on one block held one coefficient per lane (lane = natural index; the caller has zeroed
// lane 0: DC is excluded from the sum, :2381).  Only non-zero coefficients are visited, in ascending natural order, separate
// multiply and add.  The table in LDS holds 2 x the reference's entries (an exact scaling of every product and every partial sum):
// idct_run returns 2 x the reference's sum before its final * 0.25, which is the f * 8 SetFullRes forms (to_sample).
// cvu: the coefficient as the 16 bits the arena holds, zero-extended (the sign extension rides on the conversion: a loop-carried
// int16 that is sign-extended separately costs an instruction per block and row).
// idct_prep builds the list in buffer `buf` (0 / 1) and issues the reads of its row words and of its first sixteen coefficients;
// idct_run, any time later, runs the terms (nothing else may have written that buffer in between).
struct IdctPrep { uint32_t n, rows2; float ey; };
__device__ __forceinline__ IdctPrep idct_prep(uint32_t cvu, const WaveList L, uint32_t buf)
{
    IdctPrep P; P.rows2 = 0; P.ey = 0.0f;
    const bool nz = cvu != 0;
    const uint64_t mask = WBALLOT(nz);
    asm("s_bcnt1_i32_b64 %0, %1" : "=s"(P.n) : "s"(mask) : "scc");      // wave-uniform, 32 bits, in an SGPR: the exit tests are scalar compares
    // (no branch around this for a block without AC coefficients: straight-line code lets the waits ahead of idct_run count exactly the
    //  LDS operations issued behind the ones they wait for -- a join of two paths makes them wait for everything)
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    uint32_t cf;                                                 // (float)(int16_t)cvu
    asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(cf) : "v"(cvu));
    const uint32_t slot = nz ? rank : rank + L.z_off;            // zero lanes: slot 63 - (zero lanes below)
    const uint32_t bo = buf * LIST_BUF_BYTES;
    lds_w32(L.base + (slot << 2) + bo, cf);
    lds_w16(L.base + (slot << 1) + (bo + 256u), L.row_w);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    P.rows2 = lds_r32(L.a_rows + bo);
    P.ey = __uint_as_float(lds_r32(L.a_ey + bo));
    return P;
}
__device__ __forceinline__ float idct_run(const IdctPrep& P, const WaveList L, uint32_t buf, uint32_t lane)
{
    float acc = 0.0f;
    const uint32_t n = P.n, rows2 = P.rows2;
    if (n) {
        const uint32_t a_ey = L.a_ey + buf * LIST_BUF_BYTES;
        float a0, a1, a2, a3, b0, b1, b2, b3; uint32_t sp0, sp1;
        { const float ey = P.ey; const uint32_t nl = n; IDCT_ROUND(0, 1, 2, 3, 4, 5, 6, 7); }
        if (n > 16) { const float ey = __uint_as_float(lds_r32(a_ey + 64u)); const uint32_t nl = n - 16; IDCT_ROUND(8, 9, 10, 11, 12, 13, 14, 15); }
        if (n > 32) { const float ey = __uint_as_float(lds_r32(a_ey + 128u)); const uint32_t nl = n - 32; IDCT_ROUND(16, 17, 18, 19, 20, 21, 22, 23); }
        if (n > 48) { const float ey = __uint_as_float(lds_r32(a_ey + 192u)); const uint32_t nl = n - 48; IDCT_ROUND(24, 25, 26, 27, 28, 29, 30, 31); }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return acc;
}

// ---- two blocks per wave -------------------------------------------------------------------------------------------------
// The fast layouts run the same sum on TWO blocks at a time: lanes 0..31 hold block A, lanes 32..63 block B, a lane owns the output
// samples 2l and 2l+1 of its block (l = lane & 31) and the coefficients 2l and 2l+1 (one dword of the block's row).  A term of both blocks
// is then ONE table read -- ds_read_b64 at row offset + l * 8: the two table entries of the lane's samples sit side by side in the row,
// and an 8-byte read costs the LDS pipe what a 4-byte one does (profiles/r02_instr_rates.txt) -- where the one-block form needs two: the
// LDS pipe, which the term loop loads as heavily as the vector unit, does half the work (with every second table read simply left out
// the kernel runs 6.91 -> 6.33 ms, profiles/r04_backend_experiments.txt).  The price: the two lists advance in lock step (the shorter one
// runs on its zero padding), and the row address is a vector add (v_add_u32_dpp with the row word of term k broadcast in its half) in
// place of the scalar M0 write -- which also retires the v_readlane per two terms.
// Lists, per half: 64 coefficients as fp32, 64 row words (natural index * 256) as dwords; slots as in the one-block form.
struct PairList { uint32_t a_half, z0, a_rw, l8, row0; };
#define PAIR_LIST_BYTES (2 * 512)
__device__ __forceinline__ PairList pair_list(const void* mem, uint32_t lane)
{
    PairList L; const uint32_t l = lane & 31u;
    L.a_half = lds_addr(mem) + (lane >> 5) * 512u; L.z0 = 63u - 2u * l; L.a_rw = L.a_half + 256u + (lane & 15u) * 4u;
    L.l8 = l * 8u; L.row0 = (2u * l) << 8;
    return L;
}
// The rounds are generated text (tools/gen/gen_pair_round.py -> jsnoop_pair_round.h): sixteen steps, every step leaves when the longer list
// has no such term (the scalar unit is nearly idle in this form: two scalar instructions per step are free); what is fetched past the end
// of the lists is dropped behind the last step.  Round 6: the coefficient of a term no longer rides on the two multiplies as a DPP operand
// (v_mul_f32_dpp issues in 4.2 cycles, v_mul_f32 in 2.2: profiles/r04_instr_rates.txt) -- the coefficients of two consecutive terms come as
// ONE ds_read_b64 that every lane of a half aims at the same eight bytes of its list (a broadcast: one LDS cycle per half), three such pairs
// in flight; a step is then 13 vector cycles (two multiplies, two adds, the DPP address add) and 1.5 LDS instructions where it was 17 and 1
// (tools/probes/idct_bcast.hip, profiles/r06_term_loop.txt).  Registers: five table pairs v[54:63] (four reads in flight), three
// coefficient pairs v[48:53] -- fixed, inline asm cannot name the halves of a 64-bit operand.
#include "jsnoop_pair_round.h"
#define PAIR_ROUND(R)                                                                                                            \
    asm volatile(PAIR_ROUND_ASM_##R                                                                                              \
        : [acc0] "+v"(acc0), [acc1] "+v"(acc1), [ad] "=&v"(ad), [rw] "=&v"(rw)                                                   \
        : [ah] "v"(L.a_half), [arw] "v"(L.a_rw), [l8] "v"(L.l8), [nl] "s"(nl) : "scc", PAIR_ROUND_CLOBBERS)

// d: the lane's two coefficients, c[2l] | c[2l+1] << 16 (DC and, in DC-only mode, everything already masked out).
// Returns in acc0 / acc1 the sums (x 2, see idct_run) of the lane's samples 2l and 2l+1 of its block.
__device__ __forceinline__ void idct_pair(uint32_t d, const PairList L, uint32_t lane, float& acc0, float& acc1)
{
    acc0 = 0.0f; acc1 = 0.0f;
    const bool nz0 = (d & 0xFFFFu) != 0u, nz1 = (d >> 16) != 0u;
    const uint64_t m0 = WBALLOT(nz0), m1 = WBALLOT(nz1);
    const uint32_t na = (uint32_t)__builtin_popcount((uint32_t)m0) + (uint32_t)__builtin_popcount((uint32_t)m1);
    const uint32_t nb = (uint32_t)__builtin_popcount((uint32_t)(m0 >> 32)) + (uint32_t)__builtin_popcount((uint32_t)(m1 >> 32));
    uint32_t n = na > nb ? na : nb;
    asm("" : "+s"(n));                                           // wave-uniform, in an SGPR: the exit tests are scalar compares
    if (n) {
        // coefficients of the half below this one in natural order: c[2l'] and c[2l'+1] of the lanes l' < l of the same half
        uint32_t r0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u));
        r0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, r0));
        r0 -= lane >= 32u ? na : 0u;
        const uint32_t r1 = r0 + (nz0 ? 1u : 0u);
        const uint32_t s0 = nz0 ? r0 : r0 + L.z0, s1 = nz1 ? r1 : r1 + (L.z0 - 1u);      // zero entries: slot 63 - (zero entries below)
        uint32_t cf0, cf1;
        asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(cf0) : "v"(d));
        asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(cf1) : "v"(d));
        const uint32_t w0 = L.a_half + (s0 << 2), w1 = L.a_half + (s1 << 2);
        lds_w32(w0, cf0); lds_w32(w0 + 256u, L.row0); lds_w32(w1, cf1); lds_w32(w1 + 256u, L.row0 + 256u);      // (in this order the compiler pairs them: two ds_write2st64_b32)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        uint32_t ad, rw;
        { const uint32_t nl = n; PAIR_ROUND(0); }
        if (n > 16) { const uint32_t nl = n - 16; PAIR_ROUND(1); }
        if (n > 32) { const uint32_t nl = n - 32; PAIR_ROUND(2); }
        if (n > 48) { const uint32_t nl = n - 48; PAIR_ROUND(3); }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// SetFullRes :2468-2561 into the wave's LDS MCU tile (replicated eH x eV times).
// meta = comp-1 | eh<<4 | ev<<8 | (blk_ch*8)<<12 | (blk_cv*8)<<20 of the block's slot in the MCU.
__device__ __forceinline__ uint32_t tile_offset(uint32_t meta, uint32_t plane_elems, uint32_t rs, uint32_t lane)
{
    const uint32_t comp0 = meta & 15u, eh = (meta >> 4) & 15u, ev = (meta >> 8) & 15u;
    const uint32_t x0 = ((meta >> 12) & 255u) + (lane & 7) * eh, y0 = ((meta >> 20) & 255u) + (lane >> 3) * ev;
    return comp0 * plane_elems + y0 * rs + x0;
}
// fp32 sum of the terms -> sample: the reference forms f = sum * 0.25 (:2389) and (short)((short)(f * 8) + dc) (:2517-2519); both
// scalings are exact powers of two (no term sum is small enough to go denormal), so f * 8 is sum * 2 bit for bit -- and that factor
// sits in the LDS copy of the table (idct_terms): sum2 is 2 x the reference's sum already.
__device__ __forceinline__ int16_t to_sample(float sum2, int16_t dc) { return (int16_t)((int16_t)(int)sum2 + dc); }
__device__ __forceinline__ void sample_to_lds(uint32_t meta /*wave-uniform*/, int16_t smp, int16_t* pl, uint32_t rs)
{
    const uint32_t eh = (meta >> 4) & 15u, ev = (meta >> 8) & 15u;
    if (eh == 1 && ev == 1) pl[0] = smp;
    else if (eh == 2 && ev <= 2) {                               // 4:2:2 / 4:2:0 chroma: one or two 32-bit stores
        const uint32_t two = (uint32_t)(uint16_t)smp * 0x10001u;
        *reinterpret_cast<uint32_t*>(pl) = two;
        if (ev == 2) *reinterpret_cast<uint32_t*>(pl + rs) = two;
    } else
        for (uint32_t jy = 0; jy < ev; jy++) for (uint32_t ix = 0; ix < eh; ix++) pl[jy * rs + ix] = smp;
}

// brightest-pixel search (:4722-4730) over the four Y samples of a lane's trip: larger Y wins, earlier raster position breaks
// ties.  The 64-bit key is only formed when one of the four can beat (or tie with) what this lane has seen so far.
__device__ __forceinline__ void bright4(const uint2 qy, uint32_t raster0, uint64_t& bright, int& best_y)
{
    const int vy[4] = { s16_lo(qy.x), s16_hi(qy.x), s16_lo(qy.y), s16_hi(qy.y) };
    if (max(max(vy[0], vy[1]), max(vy[2], vy[3])) >= best_y) {
        #pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint64_t key = ((uint64_t)(uint32_t)(vy[k] + 32768) << 32) | (0xFFFFFFFFu - (raster0 + k));
            bright = key > bright ? key : bright;
        }
        best_y = (int)(uint32_t)(bright >> 32) - 32768;
    }
}

// Colour conversion + DIB rows of one MCU, general form: any sampling, any preview mode, optional YCC shift; the tile holds
// three full-resolution (replicated) planes.  4 pixels (16 bytes) per lane and trip.
template <bool RGB_ONLY>
__device__ __forceinline__ void mcu_to_dib(const JsImage& im, const int16_t* tile, uint32_t plane_elems, uint32_t rs, uint32_t quads, uint32_t total,
                                           uint32_t lane, uint32_t ly0, uint32_t lq0, uint32_t my, uint32_t mx, uint32_t mw, uint32_t mh, bool shifted,
                                           uint8_t* __restrict__ dibp, int16_t* __restrict__ planes, uint32_t pw, bool want_planes,
                                           uint64_t& bright, int& best_y, uint32_t& sum_y)
{
    const uint32_t img_x = im.img_x, img_y = im.img_y, mode = im.preview_mode, ncomp = im.ncomp;
    const int sh_y = shifted ? im.shift_y : 0, sh_cb = shifted ? im.shift_cb : 0, sh_cr = shifted ? im.shift_cr : 0;   // nMcuInd >= nMcuShiftInd (:4735-4739): added in int before the >> 3
    const uint32_t dq = 64u % quads, dy = 64u / quads;
    uint32_t y = ly0, q = lq0;
    // the DIB is bottom-up: the MCU's last row has the lowest address (wave-uniform), rows above it follow at +img_x pixels
    uint8_t* mcu_low = dibp + ((size_t)(img_y - (my + 1u) * mh) * img_x + (size_t)mx * mw) * 4;
    for (uint32_t p = lane; p < total; p += 64) {
        const uint32_t x = q * 4, py = my * mh + y, px = mx * mw + x;
        const uint2 qy = *reinterpret_cast<const uint2*>(tile + y * rs + x);
        const uint2 qcb = *reinterpret_cast<const uint2*>(tile + plane_elems + y * rs + x);
        const uint2 qcr = *reinterpret_cast<const uint2*>(tile + 2 * plane_elems + y * rs + x);
        bright4(qy, py * img_x + px, bright, best_y);
        const int vy[4] = { s16_lo(qy.x), s16_hi(qy.x), s16_lo(qy.y), s16_hi(qy.y) };
        const int vcb[4] = { s16_lo(qcb.x), s16_hi(qcb.x), s16_lo(qcb.y), s16_hi(qcb.y) };
        const int vcr[4] = { s16_lo(qcr.x), s16_hi(qcr.x), s16_lo(qcr.y), s16_hi(qcr.y) };
        uint32_t o[4];
        #pragma unroll
        for (int k = 0; k < 4; k++) {
            o[k] = ycc_to_bgra<RGB_ONLY>(vy[k] + sh_y, vcb[k] + sh_cb, vcr[k] + sh_cr, mode);
            sum_y += (uint32_t)(clamp_s8((vy[k] + sh_y) >> 3) + 128);      // nSumY += nFinalY (:4751), wraps mod 2^32 like the reference
        }
        uint4 v; v.x = o[0]; v.y = o[1]; v.z = o[2]; v.w = o[3];
        *reinterpret_cast<uint4*>(mcu_low + ((mh - 1u - y) * img_x + x) * 4u) = v;          // scalar base + a lane offset below 2^32
        if (want_planes) {
            int16_t* pb = planes + im.plane_off;
            const size_t pi = (size_t)py * pw + px, psz = (size_t)pw * im.blk_ymax * 8;
            *reinterpret_cast<uint2*>(pb + pi) = qy;
            if (ncomp == 3) { *reinterpret_cast<uint2*>(pb + psz + pi) = qcb; *reinterpret_cast<uint2*>(pb + 2 * psz + pi) = qcr; }
        }
        q += dq; y += dy; if (q >= quads) { q -= quads; y++; }
    }
}

// The same for the common layouts -- three components, Y un-expanded, Cb and Cr one block each replicated EH x EV times with
// EH, EV in {1, 2} (4:4:4, 4:2:2, 4:4:0, 4:2:0), default preview mode, no YCC shift.  The tile holds the Y plane and the two chroma
// blocks UN-replicated (one LDS store per block instead of up to four); a lane's four pixels share EH-fold chroma samples, whose
// clamp, int -> float conversion and multiplication by (2 - 2*0.299f) / (2 - 2*0.114f) are done once per sample.
template <uint32_t EH, uint32_t EV>
__device__ __forceinline__ void mcu_to_dib_fast(const JsImage& im, uint32_t img_x, uint32_t img_y, const int16_t* tile, uint32_t plane_elems, uint32_t rs, uint32_t quads, uint32_t total,
                                                uint32_t lane, uint32_t ly0, uint32_t lq0, uint32_t my, uint32_t mx, uint32_t mw, uint32_t mh,
                                                uint8_t* __restrict__ dibp, int16_t* __restrict__ planes, uint32_t pw, bool want_planes,
                                                uint64_t& bright, int& best_y, uint32_t& sum_y)
{   // (img_x, img_y by value: read once per wave -- through `im` they are re-read from memory after every MCU's stores)
    const uint32_t dq = 64u % quads, dy = 64u / quads;
    uint32_t y = ly0, q = lq0;
    uint8_t* mcu_low = dibp + ((size_t)(img_y - (my + 1u) * mh) * img_x + (size_t)mx * mw) * 4;
    const int16_t* cbp = tile + plane_elems; const int16_t* crp = cbp + 64;
    for (uint32_t p = lane; p < total; p += 64) {
        const uint32_t x = q * 4, py = my * mh + y, px = mx * mw + x;
        const uint2 qy = *reinterpret_cast<const uint2*>(tile + y * rs + x);
        const uint32_t ci = (y / EV) * 8 + x / EH;                                         // first chroma sample of the lane's four pixels
        uint2 qcb, qcr;                                                                    // the four pixels' chroma samples (replicated form)
        float crm[4], cbm[4];
        if (EH == 2) {
            const uint32_t wb = *reinterpret_cast<const uint32_t*>(cbp + ci), wr = *reinterpret_cast<const uint32_t*>(crp + ci);
            const float r0 = chroma_r((float)clamp_s8(s16_lo(wr) >> 3)), r1 = chroma_r((float)clamp_s8(s16_hi(wr) >> 3));
            const float b0 = chroma_b((float)clamp_s8(s16_lo(wb) >> 3)), b1 = chroma_b((float)clamp_s8(s16_hi(wb) >> 3));
            crm[0] = crm[1] = r0; crm[2] = crm[3] = r1; cbm[0] = cbm[1] = b0; cbm[2] = cbm[3] = b1;
            qcb.x = (wb & 0xFFFFu) * 0x10001u; qcb.y = (wb >> 16) * 0x10001u; qcr.x = (wr & 0xFFFFu) * 0x10001u; qcr.y = (wr >> 16) * 0x10001u;
        } else {
            qcb = *reinterpret_cast<const uint2*>(cbp + ci); qcr = *reinterpret_cast<const uint2*>(crp + ci);
            const int vcb[4] = { s16_lo(qcb.x), s16_hi(qcb.x), s16_lo(qcb.y), s16_hi(qcb.y) }, vcr[4] = { s16_lo(qcr.x), s16_hi(qcr.x), s16_lo(qcr.y), s16_hi(qcr.y) };
            #pragma unroll
            for (int k = 0; k < 4; k++) { crm[k] = chroma_r((float)clamp_s8(vcr[k] >> 3)); cbm[k] = chroma_b((float)clamp_s8(vcb[k] >> 3)); }
        }
        bright4(qy, py * img_x + px, bright, best_y);
        const int cy[4] = { clamp_s8(s16_lo(qy.x) >> 3), clamp_s8(s16_hi(qy.x) >> 3), clamp_s8(s16_lo(qy.y) >> 3), clamp_s8(s16_hi(qy.y) >> 3) };
        uint4 v;
        v.x = pack_bgr(ycc_core<true>((float)cy[0], crm[0], cbm[0])); v.y = pack_bgr(ycc_core<true>((float)cy[1], crm[1], cbm[1]));
        v.z = pack_bgr(ycc_core<true>((float)cy[2], crm[2], cbm[2])); v.w = pack_bgr(ycc_core<true>((float)cy[3], crm[3], cbm[3]));
        sum_y += (uint32_t)(cy[0] + cy[1] + cy[2] + cy[3] + 512);                          // nSumY += nFinalY (:4751)
        *reinterpret_cast<uint4*>(mcu_low + ((mh - 1u - y) * img_x + x) * 4u) = v;
        if (want_planes) {
            int16_t* pb = planes + im.plane_off;
            const size_t pi = (size_t)py * pw + px, psz = (size_t)pw * im.blk_ymax * 8;
            *reinterpret_cast<uint2*>(pb + pi) = qy; *reinterpret_cast<uint2*>(pb + psz + pi) = qcb; *reinterpret_cast<uint2*>(pb + 2 * psz + pi) = qcr;
        }
        q += dq; y += dy; if (q >= quads) { q -= quads; y++; }
    }
}

// One WAVE owns one MCU at a time: IDCT of its blocks in decode order (so self-overlapping replication,
// SetFullRes :2498-2557, resolves exactly as in the reference: later blocks overwrite earlier ones),
// samples staged in a wave-private LDS tile, then colour conversion and the MCU's DIB rows.  No
// workgroup barrier in the loop; the next MCU's coefficient rows are prefetched into registers while
// the current MCU is converted.
struct BackEndCtx {
    const JsImage* im; const int16_t* cbase; const int16_t* dccum; uint8_t* dibp; int16_t* planes;
    WaveList L; const void* listmem; int16_t* tile; const uint32_t* s_meta; uint32_t lane, wg_in_img, wgs_in_img, wave;
};
// FAST: the layouts of mcu_to_dib_fast (EH, EV = chroma expansion); otherwise the general path.
template <bool FAST, uint32_t EH, uint32_t EV>
__device__ __forceinline__ void back_end_mcus(const BackEndCtx& C, uint64_t& bright, uint32_t& sum_y)
{
    const JsImage& im = *C.im;
    const uint32_t lane = C.lane;
    const WaveList L = C.L; int16_t* tile = C.tile;
    const uint32_t nb = FAST ? EH * EV + 2u : im.blk_per_mcu, nmcu = im.mcu_xmax * im.mcu_ymax, pw = im.blk_xmax * 8;   // FAST: known at compile time
    const uint32_t mw = FAST ? 8u * EH : im.mcu_w, mh = FAST ? 8u * EV : im.mcu_h, rs = mw + 8, plane_elems = mh * rs, ncomp = im.ncomp;
    const uint32_t img_x = im.img_x, img_y = im.img_y;
    const uint32_t mcus_across = img_x / mw, shift_ind = im.shift_mcu_y * mcus_across + im.shift_mcu_x;
    const bool want_planes = im.want_planes != 0, rgb_only = im.preview_mode == 1, any_shift = (im.shift_y | im.shift_cb | im.shift_cr) != 0;
    const uint32_t quads = mw / 4, total = quads * mh, ly0 = lane / quads, lq0 = lane % quads;
    int best_y = -0x7FFFFFFF;
    if (!FAST && ncomp == 1) for (uint32_t i = lane; i < 2 * plane_elems; i += 64) tile[plane_elems + i] = 0;   // Cb = Cr = 0 for grayscale (:4709-4715)
    // A component with 1 < H < Hmax (or V) does not cover its share of the MCU: SetFullRes places its blocks 8 samples
    // apart but replicates each Hmax/H times (:2498-2557), so part of the MCU keeps the zeros of ClrFullRes (:2443).
    // The same happens when H does not divide Hmax (expansion = Hmax / H truncates: H = 2 under Hmax = 3 fills 16 of 24 columns).
    bool partial = false;
    if (!FAST) for (uint32_t cc = 1; cc <= ncomp; cc++) {
        const bool full_h = (im.samp_h[cc] * 8 == mw && im.expand_h[cc] == 1) || (im.samp_h[cc] == 1 && im.expand_h[cc] * 8 == mw);
        const bool full_v = (im.samp_v[cc] * 8 == mh && im.expand_v[cc] == 1) || (im.samp_v[cc] == 1 && im.expand_v[cc] * 8 == mh);
        partial = partial || !full_h || !full_v;
    }
    uint32_t cv[BK_CHUNK]; int16_t dcv[BK_CHUNK];
    // DC-only mode: the reference does not run the IDCT at all (:1827), whatever sits at the AC positions -- a DC symbol with a run
    // nibble stores its value there (DecodeIdctSet with ind = zrl, :1713)
    const bool with_ac = im.decode_ac != 0;
    uint32_t ac_mask = (lane != 0 && with_ac) ? 0xFFFFu : 0u;    // lane 0 holds the DC difference: not part of the sum (:2381)
    asm volatile("" : "+v"(ac_mask));                         // (kept a mask in a VGPR: v_and_b32 issues at full rate, the v_cndmask the compiler prefers does not)
    uint32_t meta[BK_CHUNK], toff[BK_CHUNK];                     // placement word (wave-uniform) and this lane's tile offset per block slot
    // m is wave-uniform (kept in SGPRs): the row addresses are a scalar base plus the lane, the DC words a scalar address.
    // A chunk is FETCHED (loads issued, nothing waits) and later TAKEN (first use of what came back): the next MCU's fetch is issued before
    // the colour phase of the current one and taken at the top of the next iteration, so the loads fly during the colour phase -- a mask or a
    // shift applied at fetch time would make the wave wait for its loads on the spot.
    const size_t coef_off = im.coef_off;                         // (a local: the image record is not re-read after the DIB stores)
    uint32_t raw[BK_CHUNK]; uint32_t dw0 = 0, dw1 = 0, dw2 = 0, dw3 = 0, dodd = 0;
    // (a lane fetches the DWORD that holds its coefficient -- lanes 2q and 2q+1 the same one -- and shifts its half down when the row is
    // taken: a 16-bit value carried around the loop is widened by the compiler at the loop's end, which is where the wave would wait)
    const uint32_t half_sh = (lane & 1u) * 16u;
    auto fetch_rows = [&](uint32_t m, uint32_t base) {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(C.cbase + ((size_t)m * nb + base) * 64) + (lane >> 1);
        #pragma unroll
        for (int j = 0; j < BK_CHUNK; j++) raw[j] = base + j < nb ? p[j * 32] : 0u;
    };
    // the DC words of the chunk: wave-uniform, so they come as aligned dwords through the scalar cache into SGPRs.  (Scalar loads share
    // their counter with the LDS: issued before the colour phase, its first LDS wait would wait for them too -- they are issued behind it.)
    auto fetch_dc = [&](uint32_t m, uint32_t base) {
        const size_t d0 = coef_off + (size_t)m * nb + base;
        const uint32_t* q32 = reinterpret_cast<const uint32_t*>(C.dccum) + (d0 >> 1);
        dodd = (uint32_t)d0 & 1u;
        dw0 = q32[0]; dw1 = q32[1]; dw2 = q32[2]; dw3 = q32[3];                 // 8 int16 from an even index cover BK_CHUNK = 6 from d0 (the arena has slack)
    };
    auto take_chunk = [&](uint32_t base) {
        // (the scalar loads are waited for HERE: while one is pending, every LDS wait of the block loop would have to wait for everything)
        asm volatile("" : : "s"(dw0), "s"(dw1), "s"(dw2), "s"(dw3));
        #pragma unroll
        for (int j = 0; j < BK_CHUNK; j++) {
            const uint32_t c = base + j;
            cv[j] = (raw[j] >> half_sh) & ac_mask;
            const uint32_t h = (uint32_t)j + dodd, w = (h >> 1) == 0 ? dw0 : ((h >> 1) == 1 ? dw1 : ((h >> 1) == 2 ? dw2 : dw3));
            dcv[j] = c < nb ? (int16_t)(w >> ((h & 1u) * 16u)) : (int16_t)0;
        }
    };
    auto place_chunk = [&](uint32_t base) {
        #pragma unroll
        for (int j = 0; j < BK_CHUNK; j++) {
            meta[j] = base + j < nb ? (uint32_t)__builtin_amdgcn_readfirstlane((int)C.s_meta[base + j]) : 0u;
            if (FAST) {                                          // Y blocks into the Y plane, the two chroma blocks as they are (64 samples each)
                const uint32_t comp0 = meta[j] & 15u;
                toff[j] = comp0 == 0 ? (((meta[j] >> 20) & 255u) + (lane >> 3)) * rs + ((meta[j] >> 12) & 255u) + (lane & 7) : plane_elems + (comp0 - 1u) * 64u + lane;
            } else toff[j] = tile_offset(meta[j], plane_elems, rs, lane);
        }
    };
    place_chunk(0);
    const uint32_t wstride = C.wgs_in_img * BK_WAVES;
    uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)(C.wg_in_img * BK_WAVES + C.wave));
    const uint32_t xmax = im.mcu_xmax, step_x = wstride % xmax, step_y = wstride / xmax;
    uint32_t mx = m % xmax, my = m / xmax;                       // MCU coordinates, stepped along with m (no division in the loop)
    if (m < nmcu) { fetch_rows(m, 0); fetch_dc(m, 0); }
    #pragma nounroll
    for (; m < nmcu; m += wstride, mx += step_x, my += step_y) {
        if (mx >= xmax) { mx -= xmax; my++; }
        take_chunk(0);
        if (partial) { for (uint32_t i = lane; i < ncomp * plane_elems; i += 64) tile[i] = 0; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); }
        // ---- IDCT of the MCU's blocks, decode order ---------------------------------------------------
        for (uint32_t base = 0; base < nb; base += BK_CHUNK) {
            if (base) { fetch_rows(m, base); fetch_dc(m, base); take_chunk(base); }
            if (nb > BK_CHUNK) place_chunk(base);
            IdctPrep P = idct_prep(cv[0], L, 0);                 // (every chunk holds at least one block)
            #pragma unroll
            for (int j = 0; j < BK_CHUNK; j++)
                if (base + j < nb) {
                    IdctPrep Pn = P;
                    if (j + 1 < BK_CHUNK && base + j + 1 < nb) Pn = idct_prep(cv[j + 1], L, (uint32_t)(j + 1) & 1u);   // the next block's list, while this block's terms run
                    const int16_t smp = to_sample(idct_run(P, L, (uint32_t)j & 1u, lane), dcv[j]);
                    if (FAST) tile[toff[j]] = smp; else sample_to_lds(meta[j], smp, tile + toff[j], rs);
                    P = Pn;
                }
        }
        const uint32_t m_next = m + wstride < nmcu ? m + wstride : m;   // (the last round fetches its own MCU again: no branch around the fetches)
        fetch_rows(m_next, 0);                                        // next MCU's rows fly during the colour phase
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (FAST) mcu_to_dib_fast<EH, EV>(im, img_x, img_y, tile, plane_elems, rs, quads, total, lane, ly0, lq0, my, mx, mw, mh, C.dibp, C.planes, pw, want_planes, bright, best_y, sum_y);
        else {
            const bool shifted = any_shift && my * mcus_across + mx >= shift_ind;
            if (rgb_only) mcu_to_dib<true>(im, tile, plane_elems, rs, quads, total, lane, ly0, lq0, my, mx, mw, mh, shifted, C.dibp, C.planes, pw, want_planes, bright, best_y, sum_y);
            else          mcu_to_dib<false>(im, tile, plane_elems, rs, quads, total, lane, ly0, lq0, my, mx, mw, mh, shifted, C.dibp, C.planes, pw, want_planes, bright, best_y, sum_y);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        fetch_dc(m_next, 0);
    }
}

// The fast layouts (three components, Y un-expanded, Cb and Cr one block each expanded EH x EV): blocks two at a time (idct_pair), decode order
// kept -- (Y0, Y1), (Y2, Y3), (Cb, Cr) for 4:2:0; a layout with an odd block count leaves half B of its last pair idle.
// HAND: the loop's loads are issued and waited for by inline asm (below) -- the one-layout kernels, whose ISA tools/check_inflight_regs.py can vouch for;
// the any-layout kernel (four instances of this loop and the general path in one function) leaves its loads and waits to the compiler.
template <uint32_t EH, uint32_t EV, bool HAND>
__device__ __forceinline__ void back_end_pairs(const BackEndCtx& C, uint64_t& bright, uint32_t& sum_y)
{
    const JsImage& im = *C.im;
    const uint32_t lane = C.lane, l = lane & 31u, hi = lane >> 5;
    const PairList L = pair_list(C.listmem, lane); int16_t* tile = C.tile;
    constexpr uint32_t nb = EH * EV + 2u, np = (nb + 1u) / 2u, mw = 8u * EH, mh = 8u * EV, rs = mw + 8u, plane_elems = mh * rs;
    const uint32_t nmcu = im.mcu_xmax * im.mcu_ymax, pw = im.blk_xmax * 8, img_x = im.img_x, img_y = im.img_y;
    const bool want_planes = im.want_planes != 0;
    constexpr uint32_t quads = mw / 4, total = quads * mh;
    const uint32_t ly0 = lane / quads, lq0 = lane % quads;
    int best_y = -0x7FFFFFFF;
    // per pair: does this half hold a block, where do the lane's two samples (2l, 2l+1: row l / 4, columns 2 (l % 4) and the next) go in the tile
    bool live[np]; uint32_t toff2[np], cmask[np];
    // DC-only mode: the reference does not run the IDCT at all (:1827); the DC difference (coefficient 0: low half of lane l == 0) is not part of the sum (:2381)
    const uint32_t ac = im.decode_ac != 0 ? (l == 0u ? 0xFFFF0000u : 0xFFFFFFFFu) : 0u;
    #pragma unroll
    for (uint32_t p = 0; p < np; p++) {
        const uint32_t slot = 2u * p + hi; live[p] = slot < nb;
        const uint32_t meta = C.s_meta[live[p] ? slot : nb - 1u], comp0 = meta & 15u;
        toff2[p] = comp0 == 0 ? (((meta >> 20) & 255u) + (l >> 2)) * rs + ((meta >> 12) & 255u) + (l & 3u) * 2u : plane_elems + (comp0 - 1u) * 64u + 2u * l;
        cmask[p] = live[p] ? ac : 0u;
        asm volatile("" : "+v"(cmask[p]));                       // (a mask in a VGPR: v_and_b32 issues at full rate, the v_cndmask the compiler prefers does not)
    }
    const size_t coef_off = im.coef_off;
    uint32_t raw[np]; int dcl[np];
    // A lane fetches the dword that holds its two coefficients (an idle half re-reads block A: nothing is read past the arena) and the
    // cumulative DC of its block.  The NEXT MCU's row of pair p is fetched as soon as this MCU's pair p sits in its lists -- into the very
    // register it came from -- and its DC word behind the pair's tile store: the loads fly under the terms of this and the later pairs and
    // under the colour phase (round 5: issued in front of the colour phase only, one MCU's 768 bytes per wave were in flight for a third of an
    // iteration, profiles/r05_backend_parts.txt).
    // Round 6: the loads are issued and WAITED FOR by hand.  vmcnt counts in order, loads and stores alike, and the compiler's own waits were
    // derived from the loop's entry path: `vmcnt(3)` in front of every tile store and `vmcnt(1)` on the back edge (the sign extension of the
    // three DC words, hoisted there) -- each drained everything but the newest loads, i.e. the rows fetched one pair earlier and the previous
    // MCU's DIB store had to land within a pair's time (s_memtime stamps: ~1000 of the ~10000 cycles of a wave's MCU in each of the three
    // tile phases, profiles/r06_backend_stamps.txt).  Per iteration the wave issues R0 D0 R1 D1 .. (rows, DC words) and one DIB store, always
    // in this order, so behind any load 2 * np younger operations have been issued when its value is needed: every use waits for
    // vmcnt(2 * np) -- a load has a whole iteration to land, the store of the previous MCU as well.  (Operations of the compiler's own in
    // between -- the plane stores -- only make that wait cover more.)  The loop's entry issues one more load in the store's place.  Nothing but these
    // statements may touch raw[] / dcl[] while a load is in flight to them: the compiler does not know (check the ISA for copies after a change).
    uint32_t voff[np], doff[np];
    #pragma unroll
    for (uint32_t p = 0; p < np; p++) { voff[p] = (live[p] ? lane : l) * 4u; doff[p] = (live[p] ? hi : 0u) * 2u; }
    auto ld_row = [&](uint32_t p, const uint32_t* base /*wave-uniform*/) {
        if (HAND) asm volatile("global_load_dword %0, %1, %2" : "+v"(raw[p]) : "v"(voff[p]), "s"(base + p * 64u) : "memory");
        else raw[p] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(base + p * 64u) + voff[p]); };
    auto ld_dc = [&](uint32_t p, const int16_t* base /*wave-uniform*/) {
        if (HAND) asm volatile("global_load_sshort %0, %1, %2" : "+v"(dcl[p]) : "v"(doff[p]), "s"(base + 2u * p) : "memory");
        else dcl[p] = (int)*reinterpret_cast<const int16_t*>(reinterpret_cast<const char*>(base + 2u * p) + doff[p]); };
#define BK_VM_WAIT(X) do { if (HAND) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(X) : "n"(2 * np)); } while (0)
    // A workgroup owns a CONTIGUOUS range of the image's MCUs and its waves step through it side by side (round 5; before: 8-MCU pieces a whole
    // grid stride apart): what a workgroup reads and writes over time is one sequential stream per MCU row.
    const uint32_t per = (nmcu + C.wgs_in_img - 1u) / C.wgs_in_img, m_begin = C.wg_in_img * per, m_end = min(m_begin + per, nmcu);
    const uint32_t wstride = BK_WAVES;
    uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)(m_begin + C.wave));
    const uint32_t xmax = im.mcu_xmax, step_x = wstride % xmax, step_y = wstride / xmax;
    uint32_t mx = m % xmax, my = m / xmax;                       // MCU coordinates, stepped along with m (no division in the loop)
    #pragma unroll
    for (uint32_t p = 0; p < np; p++) { raw[p] = 0; dcl[p] = 0; }
    if (m < m_end) {
        const uint32_t* p32 = reinterpret_cast<const uint32_t*>(C.cbase + (size_t)m * nb * 64);
        const int16_t* d16 = C.dccum + coef_off + (size_t)m * nb;
        #pragma unroll
        for (uint32_t p = 0; p < np; p++) { ld_row(p, p32); ld_dc(p, d16); }
        if (HAND) ld_dc(np - 1u, d16);                           // (in the DIB store's place: the last DC word once more, into the register it is in flight to)
    }
    #pragma nounroll
    for (; m < m_end; m += wstride, mx += step_x, my += step_y) {
        if (mx >= xmax) { mx -= xmax; my++; }
        const uint32_t m_next = m + wstride < m_end ? m + wstride : m;   // (the last round fetches its own MCU again: no branch around the fetches)
        const uint32_t* p32n = reinterpret_cast<const uint32_t*>(C.cbase + (size_t)m_next * nb * 64);
        const int16_t* d16n = C.dccum + coef_off + (size_t)m_next * nb;
        #pragma unroll
        for (uint32_t p = 0; p < np; p++) {
            float acc0, acc1;
            BK_VM_WAIT(raw[p]);
            uint32_t d = raw[p] & cmask[p];
            asm volatile("" : "+v"(d));                              // (the masked copy exists from here on: the row's register is free for the next MCU's row)
            ld_row(p, p32n);
            idct_pair(d, L, lane, acc0, acc1);
            // fp32 sums (x 2: the table holds 2 x the reference's entries) -> samples, to_sample on both; the int16 wrap of the sum is the low half
            BK_VM_WAIT(dcl[p]);
            const uint32_t x0 = (uint32_t)((int)acc0 + dcl[p]), x1 = (uint32_t)((int)acc1 + dcl[p]);
            if (live[p]) *reinterpret_cast<uint32_t*>(tile + toff2[p]) = __builtin_amdgcn_perm(x1, x0, 0x05040100u);
            ld_dc(p, d16n);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        mcu_to_dib_fast<EH, EV>(im, img_x, img_y, tile, plane_elems, rs, quads, total, lane, ly0, lq0, my, mx, mw, mh, C.dibp, C.planes, pw, want_planes, bright, best_y, sum_y);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (HAND) {
        #pragma unroll
        for (uint32_t p = 0; p < np; p++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw[p]), "+v"(dcl[p]) :: "memory");   // (the last round's fetches land in registers that are free from here on)
    }
#undef BK_VM_WAIT
}

#ifndef JS_BK_OCC
#define JS_BK_OCC 8     // 64 VGPRs: with the small tiles of the common layouts four workgroups (32 waves) fit a CU
#endif
// LAYOUT: 0 = any image (the layout is read from the descriptor, every path is in the kernel); 1..4 = every image of the launch has the
// fast layout with chroma expansion (2,2) / (2,1) / (1,2) / (1,1) -- the host checks -- and the kernel holds that one path only: a
// quarter of the code (the four-layout kernel is 20 k instructions, its 4:2:0 loop alone 4.5 k) and registers allocated for it alone.
template <int LAYOUT>
__global__ void __launch_bounds__(BK_THREADS, LAYOUT ? JS_BK_OCC : JS_BK_OCC - 2) k_idct_color(const JsImage* __restrict__ imgs, const uint32_t* __restrict__ wg_base, uint32_t nimg,
                                                           uint32_t tile_bytes, const float* __restrict__ lut_t /*[vu][yx]*/,
                                                           const int16_t* __restrict__ coef, const int16_t* __restrict__ dccum,
                                                           uint8_t* __restrict__ dib, int16_t* __restrict__ planes, uint32_t* __restrict__ side,
                                                           unsigned long long* __restrict__ wg_part)
{
    // dynamic shared memory only: the cosine table must sit at LDS offset 0 (idct_terms addresses its rows through M0)
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    float* s_lut = reinterpret_cast<float*>(s_dyn);                               // 16 KiB transposed cosine table
    uint32_t* s_meta = reinterpret_cast<uint32_t*>(s_dyn + 64 * 64 * sizeof(float)); // per block-in-MCU placement word
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint8_t* wave_mem = s_dyn + 64 * 64 * sizeof(float) + JS_MAX_BLK_PER_MCU * 4 + wave * (LIST_BYTES + tile_bytes);
    uint8_t* tail = s_dyn + 64 * 64 * sizeof(float) + JS_MAX_BLK_PER_MCU * 4 + BK_WAVES * (LIST_BYTES + tile_bytes);
    unsigned long long* s_bright = reinterpret_cast<unsigned long long*>(tail); uint32_t* s_sum = reinterpret_cast<uint32_t*>(tail + BK_WAVES * 8);
    if ((uint32_t)(size_t)s_dyn != 0u) __builtin_trap();

    uint32_t lo = 0, hi = nimg;                                  // wg_base is an exclusive prefix, nimg+1 entries
    const uint32_t bx = blockIdx.x + wg_base[0];
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (wg_base[mid] <= bx) lo = mid; else hi = mid; }
    const JsImage& im = imgs[lo];

    for (uint32_t i = tid; i < 64 * 64; i += BK_THREADS) s_lut[i] = __fmul_rn(lut_t[i], 2.0f);          // 2 x the table: see idct_terms
    if (tid < im.blk_per_mcu) { const uint32_t comp = im.blk_comp[tid];
        s_meta[tid] = (comp - 1) | (im.expand_h[comp] << 4) | (im.expand_v[comp] << 8) | ((uint32_t)im.blk_ch[tid] * 8u << 12) | ((uint32_t)im.blk_cv[tid] * 8u << 20); }
    BackEndCtx C;
    C.im = &im; C.cbase = coef + im.coef_off * 64; C.dccum = dccum; C.dibp = dib + im.dib_off; C.planes = planes;
    C.L = wave_list(wave_mem, lane); C.listmem = wave_mem; C.tile = reinterpret_cast<int16_t*>(wave_mem + LIST_BYTES);
    C.s_meta = s_meta; C.lane = lane; C.wave = wave; C.wg_in_img = bx - wg_base[lo]; C.wgs_in_img = wg_base[lo + 1] - wg_base[lo];
    __syncthreads();

    uint64_t bright = 0; uint32_t sum_y = 0;
    // the common layouts take the short colour path: Y un-expanded, Cb and Cr one block each, both expanded EH x EV with EH, EV in {1, 2}
    if (LAYOUT == 1) back_end_pairs<2, 2, true>(C, bright, sum_y);
    else if (LAYOUT == 2) back_end_pairs<2, 1, true>(C, bright, sum_y);
    else if (LAYOUT == 3) back_end_pairs<1, 2, true>(C, bright, sum_y);
    else if (LAYOUT == 4) back_end_pairs<1, 1, true>(C, bright, sum_y);
    else {
        const uint32_t eh = im.expand_h[2], ev = im.expand_v[2];
        const bool fast = js_fast_layout(im);
        if (fast && eh == 2 && ev == 2) back_end_pairs<2, 2, false>(C, bright, sum_y);
        else if (fast && eh == 2) back_end_pairs<2, 1, false>(C, bright, sum_y);
        else if (fast && ev == 2) back_end_pairs<1, 2, false>(C, bright, sum_y);
        else if (fast) back_end_pairs<1, 1, false>(C, bright, sum_y);
        else back_end_mcus<false, 1, 1>(C, bright, sum_y);
    }

    for (int off = 32; off > 0; off >>= 1) {
        const uint64_t ob = __shfl_down(bright, off); bright = ob > bright ? ob : bright;
        sum_y += __shfl_down(sum_y, off);
    }
    // one pair of atomics per workgroup (they all hit the same line of the image's status words), and the maximum only when it
    // can still raise what is there
    if (lane == 0) { s_bright[wave] = bright; s_sum[wave] = sum_y; }
    __syncthreads();
    if (tid == 0) {
        for (uint32_t w = 1; w < BK_WAVES; w++) { bright = s_bright[w] > bright ? s_bright[w] : bright; sum_y += s_sum[w]; }
        // An image spread over hundreds of workgroups (a single large image: ~1000) would queue them all on these two words -- 42 ns per
        // workgroup, 55 of the 100 us the kernel took for one 3840x2160 image: such launches leave one record per workgroup instead and
        // k_status_reduce folds them.
        if (wg_part) { wg_part[2 * (size_t)bx] = bright; wg_part[2 * (size_t)bx + 1] = sum_y; return; }
        uint32_t* sd = side + im.side_off;
        unsigned long long* bp = reinterpret_cast<unsigned long long*>(sd + 12);
        if ((unsigned long long)bright > __atomic_load_n(bp, __ATOMIC_RELAXED)) atomicMax(bp, (unsigned long long)bright);
        atomicAdd(sd + 15, sum_y);
    }
}
// brightest pixel / luminance sum of an image from the per-workgroup records of k_idct_color (one workgroup per image)
__global__ void __launch_bounds__(256) k_status_reduce(const JsImage* __restrict__ imgs, const uint32_t* __restrict__ wg_base, const unsigned long long* __restrict__ wg_part,
                                                       uint32_t* __restrict__ side)
{
    __shared__ unsigned long long s_b[4]; __shared__ uint32_t s_s[4];
    const JsImage& im = imgs[blockIdx.x];
    unsigned long long bright = 0; uint32_t sum_y = 0;
    for (uint32_t w = wg_base[blockIdx.x] + threadIdx.x; w < wg_base[blockIdx.x + 1]; w += 256) { const unsigned long long b = wg_part[2 * (size_t)w]; bright = b > bright ? b : bright; sum_y += (uint32_t)wg_part[2 * (size_t)w + 1]; }
    for (int off = 32; off > 0; off >>= 1) { const unsigned long long ob = __shfl_down(bright, off); bright = ob > bright ? ob : bright; sum_y += __shfl_down(sum_y, off); }
    if ((threadIdx.x & 63) == 0) { s_b[threadIdx.x >> 6] = bright; s_s[threadIdx.x >> 6] = sum_y; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) { bright = s_b[w] > bright ? s_b[w] : bright; sum_y += s_s[w]; }
        uint32_t* sd = side + im.side_off;
        atomicMax(reinterpret_cast<unsigned long long*>(sd + 12), bright);
        atomicAdd(sd + 15, sum_y);
    }
}

// One block through the device IDCT (known-answer probe for jsnoop_idct_block): the production term loop, then the reference's * 0.25.
__global__ void __launch_bounds__(64) k_idct_probe(const float* __restrict__ lut_t, const int16_t* __restrict__ coef64, float* __restrict__ out64)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    float* s_lut = reinterpret_cast<float*>(s_dyn);
    const WaveList L = wave_list(s_dyn + 64 * 64 * sizeof(float), threadIdx.x);
    const PairList LP = pair_list(s_dyn + 64 * 64 * sizeof(float), threadIdx.x);
    if ((uint32_t)(size_t)s_dyn != 0u) __builtin_trap();
    const uint32_t lane = threadIdx.x, l = lane & 31u;
    for (uint32_t i = lane; i < 64 * 64; i += 64) s_lut[i] = __fmul_rn(lut_t[i], 2.0f);
    __syncthreads();
    // both production forms of the term loop: the block alone (general layouts), and as both halves of a pair (the fast layouts) --
    // a sample on which any of the three results differs comes back as NaN
    const IdctPrep P = idct_prep(lane ? (uint32_t)(uint16_t)coef64[lane] : 0u, L, 0);
    const float one = __fmul_rn(idct_run(P, L, 0, lane), 0.125f);
    __syncthreads();
    const uint32_t d = reinterpret_cast<const uint32_t*>(coef64)[l] & (l == 0u ? 0xFFFF0000u : 0xFFFFFFFFu);
    float acc0, acc1;
    idct_pair(d, LP, lane, acc0, acc1);
    float* s_chk = reinterpret_cast<float*>(s_dyn);                // (the table is no longer needed)
    __syncthreads();
    s_chk[(lane >> 5) * 64u + 2u * l] = __fmul_rn(acc0, 0.125f); s_chk[(lane >> 5) * 64u + 2u * l + 1u] = __fmul_rn(acc1, 0.125f);
    __syncthreads();
    const bool same = __float_as_uint(s_chk[lane]) == __float_as_uint(one) && __float_as_uint(s_chk[64u + lane]) == __float_as_uint(one);
    out64[lane] = same ? one : __uint_as_float(0x7FC00000u);
}

// ConvertYCCtoRGBFastFloat on one triple (the RGB of the brightest pixel, :4805-4811).
__global__ void k_color_probe(int y, int cb, int cr, uint32_t* out) { out[0] = ycc_to_bgra<true>(y, cb, cr, 1); }
// The brightest pixel of image `img` as the report quotes it (:4722-4730, :4805-4811): its chroma samples from the retained planes and its RGB through the colour
// routine, from the key the back end left in side words 12 / 13 -- out[0] = Cb, [1] = Cr, [2] = BGRA, [3] / [4] = the key these belong to.  One thread.
__global__ void k_bright_probe(const JsImage* __restrict__ imgs, uint32_t img, const uint32_t* __restrict__ side, const int16_t* __restrict__ planes, uint32_t* __restrict__ out)
{
    const JsImage& im = imgs[img];
    const uint32_t k_lo = side[im.side_off + 12], k_hi = side[im.side_off + 13];
    int y = -32768, cb = -32768, cr = -32768;
    if (k_hi != 0u) {
        const uint32_t idx = 0xFFFFFFFFu - k_lo, px = idx % im.img_x, py = idx / im.img_x;
        y = (int)k_hi - 32768; cb = 0; cr = 0;
        if (im.ncomp == 3 && planes) {
            const size_t psz = (size_t)im.blk_xmax * 8 * im.blk_ymax * 8, pi = (size_t)py * im.blk_xmax * 8 + px;
            cb = planes[im.plane_off + psz + pi]; cr = planes[im.plane_off + 2 * psz + pi];
        }
    }
    out[0] = (uint32_t)cb; out[1] = (uint32_t)cr; out[2] = ycc_to_bgra<true>(y, cb, cr, 1); out[3] = k_lo; out[4] = k_hi;
}
// Every (y, cb, cr) in [-128, 127]^3 through the device colour conversion: out[(y+128)<<16 | (cb+128)<<8 | (cr+128)] = BGRA.
// Both ways of finishing a pixel (packed bytes straight from the floats; capped integers for the preview modes) must agree:
// a triple on which they do not comes back with a non-zero alpha byte.
__global__ void __launch_bounds__(256) k_color_sweep(uint32_t* __restrict__ out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const int y = ((int)(i >> 16) - 128) * 8, cb = ((int)((i >> 8) & 255u) - 128) * 8, cr = ((int)(i & 255u) - 128) * 8;
    const uint32_t a = ycc_to_bgra<true>(y, cb, cr, 1), b = ycc_to_bgra<false>(y, cb, cr, 1);
    out[i] = a == b ? a : 0xFF000000u | a;
}

// Position-keyed 64-bit checksum of every DIB: sum over 32-bit pixels of mix64(index<<32 | pixel).
// Order independent, so it reduces in parallel; tests recompute it with numpy from the oracle's DIB.
__device__ __forceinline__ uint64_t mix64(uint64_t z)
{ z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

__global__ void __launch_bounds__(256) k_dib_checksum(const JsImage* __restrict__ imgs, uint32_t nimg, uint32_t chunks_per_img,
                                                      const uint8_t* __restrict__ dib, unsigned long long* __restrict__ sums)
{
    __shared__ uint64_t s[4];
    for (uint32_t img = blockIdx.y; img < nimg; img += gridDim.y) {          // (grid rows wrap beyond the grid.y limit)
        const JsImage& im = imgs[img];
        const uint32_t npx = im.img_x * im.img_y;
        const uint32_t* p = reinterpret_cast<const uint32_t*>(dib + im.dib_off);
        uint64_t acc = 0;
        for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < npx; i += chunks_per_img * 256) acc += mix64(((uint64_t)i << 32) | p[i]);
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
        if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(&sums[img], (unsigned long long)(s[0] + s[1] + s[2] + s[3]));
        __syncthreads();
    }
}

// =====================================================================================
//  bHistoEn / bStatClipEn statistics: ConvertYCCtoRGB :4229-4326, CapYccRange :4341-4475,
//  CapRgbRange :4495-4601 evaluated over the retained int16 planes.  The pixels this path produces are the ones
//  ConvertYCCtoRGBFastFloat produces (the DIB comes from k_idct_color either way); what it adds is a set of
//  reductions: min / max / sum records (PixelCcHisto), clip counters (PixelCcClip), a 2048-bin histogram of the
//  raw Y samples and 128-bin histograms of the final R, G, B.
//  stats block (u32): [0..36] PixelCcHisto (36 ints + nCount), [37..49] PixelCcClip, [50..433] R,G,B bins,
//  [434..2481] Y bins, [2482..2487] how many YCC range events the image has in all (Y<0, Y>255, Cb<0, Cb>255,
//  Cr<0, Cr>255): the reference only counts those while fewer than 10 warnings were issued (:4372-4378), which
//  the host resolves from these totals -- or, when the budget is exceeded, with k_clip_order below.
// =====================================================================================
struct StatPix { int pre[3]; int clipv[3]; int fin[3]; int lim[3]; int rgb[3]; };
__device__ __forceinline__ void stat_pixel(const JsImage& im, const int16_t* __restrict__ pl, uint32_t p, uint32_t shift_ind, StatPix& o)
{
    const uint32_t W = im.img_x, pw = im.blk_xmax * 8, py = p / W, px = p - py * W;
    const size_t pi = (size_t)py * pw + px, psz = (size_t)pw * im.blk_ymax * 8;
    o.pre[0] = pl[pi]; o.pre[1] = im.ncomp == 3 ? pl[psz + pi] : 0; o.pre[2] = im.ncomp == 3 ? pl[2 * psz + pi] : 0;   // :4683-4693
    const uint32_t mi = (py / im.mcu_h) * (W / im.mcu_w) + px / im.mcu_w;
    if (mi >= shift_ind) { o.pre[0] += im.shift_y; o.pre[1] += im.shift_cb; o.pre[2] += im.shift_cr; }                   // :4735-4739
    #pragma unroll
    for (int c = 0; c < 3; c++) {
        o.clipv[c] = (o.pre[c] + 1024) / 8;                      // C division, truncates toward zero (:4265-4267)
        o.fin[c] = min(max(o.clipv[c], 0), 255);                 // CapYccRange
    }
    const float kr = 0.299f, kg = 0.587f, kb = 0.114f;
    const float cr_mul = 2 - 2 * kr, cb_mul = 2 - 2 * kb;
    const float fy = (float)(o.fin[0] - 128);
    float r = __fadd_rn(__fmul_rn((float)(o.fin[2] - 128), cr_mul), fy);
    float b = __fadd_rn(__fmul_rn((float)(o.fin[1] - 128), cb_mul), fy);
    float g = __fdiv_rn(__fsub_rn(__fsub_rn(fy, __fmul_rn(kb, b)), __fmul_rn(kr, r)), kg);
    r = __fadd_rn(r, 128.0f); b = __fadd_rn(b, 128.0f); g = __fadd_rn(g, 128.0f);
    o.lim[0] = (int)r; o.lim[1] = (int)g; o.lim[2] = (int)b;      // CapRgbRange truncates first, then range-checks the ints
    #pragma unroll
    for (int c = 0; c < 3; c++) o.rgb[c] = min(max(o.lim[c], 0), 255);
}

#define ST_THREADS 256
__global__ void __launch_bounds__(ST_THREADS) k_color_stats(const JsImage* __restrict__ imgs, uint32_t img, const int16_t* __restrict__ planes,
                                                            uint32_t hist_en, uint32_t* __restrict__ stats)
{
    __shared__ int s_mm[36]; __shared__ uint32_t s_cnt[24]; __shared__ uint32_t s_bins[3 * 128 + 2048];
    const JsImage& im = imgs[img];
    const int16_t* pl = planes + im.plane_off;
    const uint32_t npix = im.img_x * im.img_y, shift_ind = im.shift_mcu_y * (im.img_x / im.mcu_w) + im.shift_mcu_x;
    const uint32_t t = threadIdx.x;
    if (t < 36) s_mm[t] = 0;                                    // the reference's records start from memset(0) (:3146-3147)
    if (t < 24) s_cnt[t] = 0;
    for (uint32_t i = t; i < 3 * 128 + 2048; i += ST_THREADS) s_bins[i] = 0;
    __syncthreads();
    int mn[12], mx[12]; uint32_t sm[12], clip[12], tot[6], n = 0;
    #pragma unroll
    for (int i = 0; i < 12; i++) { mn[i] = 0; mx[i] = 0; sm[i] = 0; clip[i] = 0; }
    #pragma unroll
    for (int i = 0; i < 6; i++) tot[i] = 0;
    for (uint32_t p = blockIdx.x * ST_THREADS + t; p < npix; p += gridDim.x * ST_THREADS) {
        StatPix q; stat_pixel(im, pl, p, shift_ind, q);
        #pragma unroll
        for (int c = 0; c < 3; c++) {
            tot[2 * c] += q.clipv[c] < 0; tot[2 * c + 1] += q.clipv[c] > 255;
            clip[6 + 2 * c] += q.lim[c] < 0; clip[7 + 2 * c] += q.lim[c] > 255;
        }
        if (hist_en) {
            #pragma unroll
            for (int c = 0; c < 3; c++) {                          // PixelCcHisto groups: PreclipYCC 0-2, ClipYCC 3-5, ClipRGB 6-8, PreclipRGB 9-11
                mn[c] = min(mn[c], q.pre[c]); mx[c] = max(mx[c], q.pre[c]); sm[c] += (uint32_t)q.pre[c];
                mn[3 + c] = min(mn[3 + c], q.clipv[c]); mx[3 + c] = max(mx[3 + c], q.clipv[c]); sm[3 + c] += (uint32_t)q.clipv[c];
                mn[6 + c] = min(mn[6 + c], q.rgb[c]); mx[6 + c] = max(mx[6 + c], q.rgb[c]); sm[6 + c] += (uint32_t)q.rgb[c];
                mn[9 + c] = min(mn[9 + c], q.lim[c]); mx[9 + c] = max(mx[9 + c], q.lim[c]); sm[9 + c] += (uint32_t)q.lim[c];
                atomicAdd(&s_bins[c * 128 + (uint32_t)q.rgb[c] / 2u], 1u);        // 256 / HISTO_BINS = 2 (:4313-4317)
            }
            atomicAdd(&s_bins[384 + (uint32_t)(min(max(q.pre[0], -1024), 1023) + 1024)], 1u);   // m_anHistoYFull (:4254-4259)
            n++;
        }
    }
    #pragma unroll
    for (int i = 0; i < 12; i++) {
        if (hist_en) { atomicMin(&s_mm[3 * i], mn[i]); atomicMax(&s_mm[3 * i + 1], mx[i]); atomicAdd(reinterpret_cast<uint32_t*>(&s_mm[3 * i + 2]), sm[i]); }
        if (i >= 6) atomicAdd(&s_cnt[i], clip[i]);
    }
    #pragma unroll
    for (int i = 0; i < 6; i++) atomicAdd(&s_cnt[12 + i], tot[i]);
    atomicAdd(&s_cnt[18], n);
    __syncthreads();
    if (hist_en && t < 36) {
        int* g = reinterpret_cast<int*>(stats);
        if (t % 3 == 0) atomicMin(&g[t], s_mm[t]); else if (t % 3 == 1) atomicMax(&g[t], s_mm[t]); else atomicAdd(&stats[t], (uint32_t)s_mm[t]);
    }
    if (t == 36) atomicAdd(&stats[36], s_cnt[18]);
    if (t >= 6 && t < 12) atomicAdd(&stats[37 + t], s_cnt[t]);               // RGB clip counters are unconditional (:4532-4586)
    if (t < 6) atomicAdd(&stats[2482 + t], s_cnt[12 + t]);
    if (hist_en) for (uint32_t i = t; i < 3 * 128 + 2048; i += ST_THREADS) { const uint32_t v = s_bins[i]; if (v) atomicAdd(&stats[50 + i], v); }
}

// The first `budget` (<= 10) YCC range events of the image in the reference's visiting order: pixels in raster order,
// within a pixel Y over, Y under, Cb over, Cb under, Cr over, Cr under (:4370-4462).  One workgroup, 1024 pixels per
// step, stops as soon as the budget is used up.  out[0..5] is indexed like PixelCcClip (Y<0, Y>255, Cb<0, Cb>255, Cr<0,
// Cr>255); out[6 + 5*k ..] describes event k for the warning text: MCU x | y << 16, kind (0 Y over, 1 Y under, 2 Cb
// over, ...), and the three values as the message prints them (earlier clips of the same pixel already applied).
__global__ void __launch_bounds__(1024) k_clip_order(const JsImage* __restrict__ imgs, uint32_t img, const int16_t* __restrict__ planes,
                                                     uint32_t budget, uint32_t* __restrict__ out)
{
    __shared__ uint32_t s_scan[1024]; __shared__ uint32_t s_out[6]; __shared__ uint32_t s_run;
    const JsImage& im = imgs[img];
    const int16_t* pl = planes + im.plane_off;
    const uint32_t npix = im.img_x * im.img_y, shift_ind = im.shift_mcu_y * (im.img_x / im.mcu_w) + im.shift_mcu_x;
    const uint32_t t = threadIdx.x;
    if (t < 6) s_out[t] = 0;
    if (t == 0) s_run = 0;
    __syncthreads();
    for (uint32_t base = 0; base < npix; base += 1024) {
        uint32_t ev = 0;                                         // bit 2c: component c over, bit 2c+1: component c under
        StatPix q;
        if (base + t < npix) {
            stat_pixel(im, pl, base + t, shift_ind, q);
            for (int c = 0; c < 3; c++) ev |= (q.clipv[c] > 255 ? 1u : 0u) << (2 * c) | (q.clipv[c] < 0 ? 2u : 0u) << (2 * c);
        }
        const uint32_t cnt = (uint32_t)__builtin_popcount(ev);
        s_scan[t] = cnt; __syncthreads();
        for (uint32_t d = 1; d < 1024; d <<= 1) { const uint32_t a = t >= d ? s_scan[t - d] : 0; __syncthreads(); s_scan[t] += a; __syncthreads(); }
        uint32_t ord = s_run + s_scan[t] - cnt;
        if (ev) {
            const uint32_t p = base + t, py = p / im.img_x, px = p - py * im.img_x;
            int cur[3] = { q.clipv[0], q.clipv[1], q.clipv[2] };
            for (uint32_t e = ev; e; e &= e - 1, ord++) {
                const uint32_t bit = (uint32_t)__builtin_ctz(e), c = bit >> 1;
                if (ord < budget) {
                    atomicAdd(&s_out[c * 2 + ((bit & 1) ? 0 : 1)], 1u);
                    uint32_t* r = out + 6 + ord * 5;
                    r[0] = (px / im.mcu_w) | ((py / im.mcu_h) << 16); r[1] = bit; r[2] = (uint32_t)cur[0]; r[3] = (uint32_t)cur[1]; r[4] = (uint32_t)cur[2];
                }
                cur[c] = (bit & 1) ? 0 : 255;
            }
        }
        __syncthreads();
        if (t == 1023) s_run += s_scan[1023];
        __syncthreads();
        if (s_run >= budget) break;
    }
    __syncthreads();
    if (t < 6) out[t] = s_out[t];
}

// TIFF export (OnToolsExporttiff, source/JPEGsnoopDoc.cpp:2110-2180): the pixel strip in file order, top-down.
//   mode 0: R,G,B bytes from the bottom-up BGRA DIB; mode 1: the same as 16-bit samples v << 8, big-endian (bytes v, 0);
//   mode 2: Y,Cb,Cr from the int16 planes, clamped to [-1024, 1023], (1024 + v) >> 3.
__global__ void __launch_bounds__(256) k_tiff_pack(const JsImage* __restrict__ imgs, uint32_t img, const uint8_t* __restrict__ dib,
                                                   const int16_t* __restrict__ planes, int mode, uint8_t* __restrict__ out)
{
    const JsImage& im = imgs[img];
    const uint32_t W = im.img_x, H = im.img_y, npix = W * H;
    const uint8_t* src = dib + im.dib_off;
    const int16_t* pl = planes + im.plane_off;
    const size_t psz = (size_t)im.blk_xmax * 8 * im.blk_ymax * 8;
    for (uint32_t p = blockIdx.x * 256 + threadIdx.x; p < npix; p += gridDim.x * 256) {
        const uint32_t y = p / W, x = p - y * W;
        if (mode == 2) {
            uint8_t* o = out + (size_t)p * 3;
            for (int c = 0; c < 3; c++) { const int v = min(max((int)pl[c * psz + p], -1024), 1023); o[c] = (uint8_t)((1024 + v) >> 3); }
        } else {
            const uint32_t bgra = *reinterpret_cast<const uint32_t*>(src + ((size_t)(H - 1 - y) * W + x) * 4);
            const uint8_t r = (uint8_t)(bgra >> 16), g = (uint8_t)(bgra >> 8), b = (uint8_t)bgra;
            if (mode == 0) { uint8_t* o = out + (size_t)p * 3; o[0] = r; o[1] = g; o[2] = b; }
            else { uint8_t* o = out + (size_t)p * 6; o[0] = r; o[1] = 0; o[2] = g; o[3] = 0; o[4] = b; o[5] = 0; }
        }
    }
}

// ------------------------------------------------------------------------------ launch wrappers
void js_launch_entropy_exact(hipStream_t st, const JsImage* imgs, const uint32_t* sel, uint32_t nsel, const JsTableSet* tables,
                             const uint8_t* raw, int16_t* coef, int16_t* dccum, uint32_t* side, int side_only, uint32_t* events)
{
    if (!nsel) return;
    ExactTail none; none.flags = nullptr; none.seg_tab = nullptr; none.mcu_rst = nullptr; none.mcu_pos = nullptr; none.us_out = nullptr; none.us_threads = 0;
    hipLaunchKernelGGL(k_entropy_exact, dim3(nsel), dim3(64), 0, st, imgs, sel, nsel, tables, raw, coef, dccum, side, side_only, events, none);
}
int js_launch_idct_color(hipStream_t st, const JsImage* imgs, const uint32_t* wg_base, uint32_t nimg, uint32_t total_wgs, uint32_t tile_bytes,
                         const float* lut_t, const int16_t* coef, const int16_t* dccum, uint8_t* dib, int16_t* planes, uint32_t* side, int layout, unsigned long long* wg_part)
{
    if (!total_wgs) return 0;
    // tile_bytes: the largest per-wave tile any image of the launch needs (js_tile_bytes).  Ordinary images leave room for four
    // workgroups per CU; a 4 x 4 sampled image (32 x 32 MCU) needs more than the default 64 KiB limit and is opted in explicitly.
    const size_t lds = 64 * 64 * sizeof(float) + JS_MAX_BLK_PER_MCU * 4 + (size_t)BK_WAVES * (LIST_BYTES + tile_bytes) + BK_WAVES * 12;
    if (lds > 160u * 1024u) return -2;
    if (lds > 64u * 1024u) {
        // The attribute belongs to the function object of the CURRENT device (one host thread per GPU is a supported use of the C ABI):
        // the size already opted in is remembered per device, raised monotonically, and read / written with atomics.
        static std::atomic<size_t> opted[JS_MAX_DEVICES];
        int devi = 0; if (hipGetDevice(&devi) != hipSuccess || devi < 0) return -3;
        layout = 0;                                            // (the fast layouts never need it: their tiles are at most 16 x 16 samples)
        if (devi >= JS_MAX_DEVICES || lds > opted[devi].load(std::memory_order_acquire)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_idct_color<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -3;
            if (devi < JS_MAX_DEVICES) { size_t cur = opted[devi].load(std::memory_order_relaxed); while (cur < lds && !opted[devi].compare_exchange_weak(cur, lds)) {} }
        }
    }
    switch (layout) {
    case 1: hipLaunchKernelGGL(k_idct_color<1>, dim3(total_wgs), dim3(BK_THREADS), lds, st, imgs, wg_base, nimg, tile_bytes, lut_t, coef, dccum, dib, planes, side, wg_part); break;
    case 2: hipLaunchKernelGGL(k_idct_color<2>, dim3(total_wgs), dim3(BK_THREADS), lds, st, imgs, wg_base, nimg, tile_bytes, lut_t, coef, dccum, dib, planes, side, wg_part); break;
    case 3: hipLaunchKernelGGL(k_idct_color<3>, dim3(total_wgs), dim3(BK_THREADS), lds, st, imgs, wg_base, nimg, tile_bytes, lut_t, coef, dccum, dib, planes, side, wg_part); break;
    case 4: hipLaunchKernelGGL(k_idct_color<4>, dim3(total_wgs), dim3(BK_THREADS), lds, st, imgs, wg_base, nimg, tile_bytes, lut_t, coef, dccum, dib, planes, side, wg_part); break;
    default: hipLaunchKernelGGL(k_idct_color<0>, dim3(total_wgs), dim3(BK_THREADS), lds, st, imgs, wg_base, nimg, tile_bytes, lut_t, coef, dccum, dib, planes, side, wg_part); break;
    }
    if (wg_part) hipLaunchKernelGGL(k_status_reduce, dim3(nimg), dim3(256), 0, st, imgs, wg_base, wg_part, side);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
// k_write2 stores eight DC differences as ONE 16-byte vector at an address that is only 2-byte aligned (global memory takes unaligned vector accesses
// in the mode the HIP runtime runs gfx9 devices in).  A device set up to enforce alignment would fault or split it wrongly: this probe does the same
// store once per device when the first batch is created (JsnoopBatch::init) and the library refuses to work if the bytes do not arrive as written.
__global__ void k_unaligned_probe(uint16_t* buf /*>= 24 halves, zeroed*/)
{
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t t; t.x = 0x00020001u; t.y = 0x00040003u; t.z = 0x00060005u; t.w = 0x00080007u;
    *reinterpret_cast<u32x4_t*>(buf + 3) = t;                      // byte offset 6
}
void js_launch_unaligned_probe(hipStream_t st, void* buf) { hipLaunchKernelGGL(k_unaligned_probe, dim3(1), dim3(1), 0, st, (uint16_t*)buf); }
void js_launch_idct_probe(hipStream_t st, const float* lut_t, const int16_t* coef64, float* out64)
{ hipLaunchKernelGGL(k_idct_probe, dim3(1), dim3(64), 64 * 64 * sizeof(float) + LIST_BYTES, st, lut_t, coef64, out64); }
void js_launch_color_probe(hipStream_t st, int y, int cb, int cr, uint32_t* out)
{ hipLaunchKernelGGL(k_color_probe, dim3(1), dim3(1), 0, st, y, cb, cr, out); }
void js_launch_bright_probe(hipStream_t st, const JsImage* imgs, uint32_t img, const uint32_t* side, const int16_t* planes, uint32_t* out8)
{ hipLaunchKernelGGL(k_bright_probe, dim3(1), dim3(1), 0, st, imgs, img, side, planes, out8); }
void js_launch_color_sweep(hipStream_t st, uint32_t* out)
{ hipLaunchKernelGGL(k_color_sweep, dim3(1u << 16), dim3(256), 0, st, out); }
void js_launch_color_stats(hipStream_t st, const JsImage* imgs, uint32_t img, const int16_t* planes, int hist_en, uint32_t* stats)
{ hipLaunchKernelGGL(k_color_stats, dim3(512), dim3(ST_THREADS), 0, st, imgs, img, planes, (uint32_t)(hist_en != 0), stats); }
void js_launch_clip_order(hipStream_t st, const JsImage* imgs, uint32_t img, const int16_t* planes, uint32_t budget, uint32_t* out6)
{ hipLaunchKernelGGL(k_clip_order, dim3(1), dim3(1024), 0, st, imgs, img, planes, budget, out6); }
void js_launch_tiff_pack(hipStream_t st, const JsImage* imgs, uint32_t img, const uint8_t* dib, const int16_t* planes, int mode, uint8_t* out)
{ hipLaunchKernelGGL(k_tiff_pack, dim3(1024), dim3(256), 0, st, imgs, img, dib, planes, mode, out); }
// the three arenas a decode starts from zero with (side outputs, MCU restart marks, flag words) in one launch instead of three fills
__global__ void __launch_bounds__(256) k_clear3(uint4* __restrict__ a, size_t na, uint4* __restrict__ b, size_t nb, uint4* __restrict__ c, size_t nc,
                                                const JsImage* __restrict__ imgs, uint32_t nimg, uint32_t* __restrict__ side)
{
    const uint4 z = make_uint4(0, 0, 0, 0);
    // imgs: only the sixteen status words of every image's side block (what the decode itself reads or accumulates into: scan length, interval count,
    // brightest pixel, luminance sum) -- histogram and maps belong to the side pass, which clears them itself; na == 0 then
    const size_t ns = imgs ? (size_t)nimg * (JS_SIDE_HISTO / 4) : 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < ns + na + nb + nc; i += (size_t)gridDim.x * 256) {
        if (i < ns) reinterpret_cast<uint4*>(side + imgs[i / (JS_SIDE_HISTO / 4)].side_off)[i % (JS_SIDE_HISTO / 4)] = z;
        else if (i < ns + na) a[i - ns] = z; else if (i < ns + na + nb) b[i - ns - na] = z; else c[i - ns - na - nb] = z;
    }
}
void js_launch_clear3(hipStream_t st, void* a, size_t a_bytes, void* b, size_t b_bytes, void* c, size_t c_bytes, const JsImage* imgs, uint32_t nimg)   // sizes rounded UP to 16 bytes: the arenas have the slack
{
    // imgs != nullptr: of the side arena `a` only every image's status words are cleared (a decode of the parallel path); else all of it
    const size_t na = imgs ? 0 : (a_bytes + 15) / 16, nb = (b_bytes + 15) / 16, nc = (c_bytes + 15) / 16, tot = na + nb + nc + (imgs ? (size_t)nimg * (JS_SIDE_HISTO / 4) : 0);
    if (!tot) return;
    const uint32_t wgs = (uint32_t)std::min<size_t>(2048, (tot + 1023) / 1024);
    hipLaunchKernelGGL(k_clear3, dim3(wgs), dim3(256), 0, st, (uint4*)a, na, (uint4*)b, nb, (uint4*)c, nc, imgs, nimg, (uint32_t*)a);
}
// five word ranges to zero in one launch (the outputs of a side pass: js_side_parallel_enqueue)
__global__ void __launch_bounds__(256) k_clear5(uint32_t* __restrict__ a, size_t na, uint32_t* __restrict__ b, size_t nb, uint32_t* __restrict__ c, size_t nc,
                                                uint32_t* __restrict__ d, size_t nd, uint32_t* __restrict__ e, size_t ne)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < na + nb + nc + nd + ne; i += (size_t)gridDim.x * 256) {
        if (i < na) a[i] = 0u; else if (i < na + nb) b[i - na] = 0u; else if (i < na + nb + nc) c[i - na - nb] = 0u;
        else if (i < na + nb + nc + nd) d[i - na - nb - nc] = 0u; else e[i - na - nb - nc - nd] = 0u;
    }
}
void js_launch_clear5(hipStream_t st, uint32_t* a, size_t na, uint32_t* b, size_t nb, uint32_t* c, size_t nc, uint32_t* d, size_t nd, uint32_t* e, size_t ne)   // word counts
{
    const size_t tot = na + nb + nc + nd + ne;
    if (!tot) return;
    hipLaunchKernelGGL(k_clear5, dim3((uint32_t)std::min<size_t>(1024, (tot + 1023) / 1024)), dim3(256), 0, st, a, na, b, nb, c, nc, d, nd, e, ne);
}
void js_launch_dib_checksum(hipStream_t st, const JsImage* imgs, uint32_t nimg, const uint8_t* dib, unsigned long long* sums)
{
    if (!nimg) return;
    const uint32_t chunks = 64;
    hipLaunchKernelGGL(k_dib_checksum, dim3(chunks, nimg < 65535u ? nimg : 65535u), dim3(256), 0, st, imgs, nimg, chunks, dib, sums);
}

// =====================================================================================
//  Parallel entropy path.
//
//  (1) k_unstuff_fused (large jobs) / k_unstuff_write<true> (small jobs) : apply the reference's byte
//      rules (BuffAddByte :1386-1573) once, in parallel and in ONE pass (a chunk's place in its image from a
//      decoupled look-back over the chunks before it): drop the 00 after FF, cut the stream at every
//      RSTn into restart intervals (interval table = start byte of each interval in the
//      compacted stream).  After this the bit cursor can be advanced with plain word loads.
//      (k_unstuff_count / _scan / _write<false> / k_interleave: the multi-pass form, kept as a cross-check.)
//  (2) k_sync : every thread owns one sub-sequence (64 bytes ... 1 KiB).  It first decodes speculatively
//      over its tail, then repeatedly re-decodes from its left neighbour's exit state
//      until the exit states stop changing (Huffman codes self-synchronise after a few
//      symbols).  A fixed point of the chain is exactly the sequential decode.  (Small jobs: k_cand_*.)
//  (3) k_block_scan : exclusive prefix sum of "blocks completed per sub-sequence" = the
//      absolute block index at which every sub-sequence starts writing.
//  (4) k_write2 : final decode from the synchronised entry states; dequantise + de-zigzag
//      (DecodeIdctSet :2270-2303) straight into the coefficient arena; verifies the chain.
//  (5) k_dc_scan : int16 wrapping prefix sum of the DC differences per component, reset at
//      every interval boundary actually present in the stream (:3280, :2693-2703, :1660).
//  Anything that deviates from a well-formed scan raises a JSNOOP_FLAG_* bit for the image;
//  what the walks followed the reference's way stays (bad codes, restarts off an MCU boundary, a decode that ends
//  at an over-read: js_parallel_fixup), the rest is re-decoded by k_entropy_exact from the first anomaly on, so
//  malformed streams stay reference-exact.
// =====================================================================================
#define US_THREADS 256
#define US_CHUNK   (US_THREADS * 16)
#define SY_THREADS 256
#define SY_HALO    2                   // k_sync: lanes of a workgroup that walk the sub-sequences in front of its first one
static_assert(SY_THREADS == JS_SY_THREADS && SY_HALO == JS_SY_HALO, "the host deals the sub-sequences to the workgroups with these");
// sub-sequence length is a per-batch choice: WL = log2(32-bit words per sub-sequence) = 4 (64 B, a handful of images), 5 (128 B) or 7 (512 B, large batches);
// 6 and 8 are instantiated for experiments (JSNOOP_SUB_WL)
#define SUB_BITS   (32u << WL)
#ifndef JS_SPEC_TAIL7
#define JS_SPEC_TAIL7 1536u
#endif
#define SYNC_SPEC_TAIL (WL >= 7 ? JS_SPEC_TAIL7 : 1024u)   // bits at the end of a sub-sequence the first (speculative) walk covers

// flag arena: two words per image -- [2 * img] the F_* bits, [2 * img + 1] the first block (decode order) at which something other than a
// coefficient-index overflow was seen (stored complemented; 0 = none): everything before it is what the reference decodes, and the exact-mirror
// reader can take over from the MCU that holds it (k_entropy_exact, tail mode) instead of from the first byte of the scan.
#define FLAG_OR(flags, img, bits) atomicOr(&(flags)[2u * (img)], (bits))
// The anomaly word is a KEY, block << 4 | kind, so that the first anomaly of an image also says what it was: kind 0 = "the mirror takes over here";
// 1..8 = the reference's decode ENDS in this block (value bits of a symbol ran past the end of a restart interval: its register over-reads,
// scan_end and scan_bad are set for good, :1229-1282 and :3623-3625): AK_DEAD + bit 0 "behind the block's DC symbol (its DC difference stands)"
// + bit 1 "a restart was handled INSIDE this block before that (its mark on the MCU is the reference's; otherwise a mark on this block comes from
// the walk going on over bits the reference never reads)" + bit 2 "reported by a lane that entered the block in its middle" -- the lane that owns
// the block saw all of it, reports the same symbol, and sorts first.  An anomaly of kind 0 in the same block sorts in front of both.
#define AK_MIRROR  0u
#define AK_DEAD    1u
#define ANOM_KEY(blk, kind) (((uint32_t)(blk) << 4) | (kind))
#define ANOM_MIN(flags, img, key) atomicMax(&(flags)[2u * (img) + 1u], ~(uint32_t)(key))   // kept as the maximum of ~key: "none" is 0, one memset clears the arena
#define F_BAD_CODE      0x0001u
#define F_OVERRUN       0x0002u
#define F_COEF_OVERFLOW 0x0004u
#define F_RST_MISALIGN  0x0008u
#define F_SHORT         0x0010u
#define F_NOSYNC        0x0080u
#define F_BAD_EDGE      0x0100u          // a code that matches nothing within 64 bits of an interval end: what the reference does there depends on its look-ahead

// Physical layout of the compacted stream: 64 consecutive sub-sequences (64 x 128 B = 8 KiB, or 64 x 512 B = 32 KiB) form a group
// stored word-interleaved -- word w of sub-sequence l sits at 32-bit index (group*32 + w)*64 + l -- so that
// the 64 lanes of a wave, each walking its own sub-sequence at roughly the same pace, read one coalesced
// 256-byte row per refill instead of 64 different cache lines.
// 64-byte pieces (WL = 4: small jobs, where nothing is bandwidth-bound and every launch is 1-2 % of the decode) read the linear stream as it is: no transpose pass.
template <int WL> __device__ __forceinline__ uint32_t phys_word(uint32_t W) { return WL == 4 ? W : (W & ~((64u << WL) - 1u)) | ((W & ((1u << WL) - 1u)) << 6) | ((W >> WL) & 63u); }
template <int WL> __device__ __forceinline__ uint32_t phys_byte(uint32_t B) { return (phys_word<WL>(B >> 2) << 2) | (B & 3u); }

struct UsBytes { uint32_t keep_mask, rst_mask; };

// Classifies the 16 bytes at absolute raw offset `o16` of image `im` (scan range [s,e)).
// Four "byte is zero" flags of a word, gathered into bits 0..3 (exact: no borrow crosses a byte; the multiply lines the four flag bits
// up in the top nibble, every partial product lands on a bit of its own).
__device__ __forceinline__ uint32_t us_zero_bytes(uint32_t x)
{
    const uint32_t nz = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x;          // bit 7 of every byte: byte != 0
    return (((~nz & 0x80808080u) >> 7) * 0x10204080u) >> 28;
}
__device__ __forceinline__ UsBytes us_classify(const uint8_t* __restrict__ raw, uint64_t o16, uint64_t s, uint64_t e, uint4* words = nullptr)
{
    UsBytes r; r.keep_mask = 0; r.rst_mask = 0;
    if (o16 + 16 <= s || o16 >= e) return r;
    const uint4 v = *reinterpret_cast<const uint4*>(raw + o16);
    if (words) *words = v;
    const uint32_t w[4] = { v.x, v.y, v.z, v.w };
    uint32_t prev = o16 > s ? raw[o16 - 1] : 0u;             // bytes before the scan start never count as FF
    const uint32_t next16 = (o16 + 16 < e) ? raw[o16 + 16] : 0u;
    if (o16 > s && o16 + 17 <= e) {
        // All sixteen bytes, the one before and the one after lie inside the scan (everything but the first and last few threads of an
        // image): the byte rules as word arithmetic.  ff = bytes that are FF, rc = bytes D0..D7 (bit j = byte j of the sixteen).
        uint32_t ff = 0, rc = 0, zb = 0;
        #pragma unroll
        for (int q = 0; q < 4; q++) { ff |= us_zero_bytes(~w[q]) << (4 * q); rc |= us_zero_bytes((w[q] & 0xF8F8F8F8u) ^ 0xD0D0D0D0u) << (4 * q); zb |= us_zero_bytes(w[q]) << (4 * q); }
        const uint32_t rc_next = (next16 & 0xF8u) == 0xD0u ? 1u : 0u;
        const uint32_t is_rst = ff & ((rc >> 1) | (rc_next << 15));            // FF followed by D0..D7
        // the byte after an FF is dropped when it is the stuffed 00 or the RSTn code; anything else behind an FF (another FF: :1486-1525, a
        // marker that is no RSTn: :1527-1561) stays in the stream as data, exactly as BuffAddByte keeps it ("skip 1") -- inside a well-formed
        // scan such bytes do not occur; a scan decoded THROUGH its stray markers (js_parallel_fixup) relies on it
        const uint32_t after_ff = ((ff << 1) | (prev == 0xFFu ? 1u : 0u)) & 0xFFFFu & (zb | rc);
        r.keep_mask = ~(after_ff | is_rst) & 0xFFFFu; r.rst_mask = is_rst;
        return r;
    }
    #pragma unroll
    for (int j = 0; j < 16; j++) {
        const uint64_t o = o16 + j;
        const uint32_t b = (w[j >> 2] >> ((j & 3) * 8)) & 255u;
        const uint32_t nb = j < 15 ? ((w[(j + 1) >> 2] >> (((j + 1) & 3) * 8)) & 255u) : next16;
        const bool in = o >= s && o < e;
        const bool is_rst = in && b == 0xFF && (o + 1 < e) && nb >= 0xD0 && nb <= 0xD7;
        const bool drop = (prev == 0xFF && o > s && (b == 0x00 || (b >= 0xD0 && b <= 0xD7))) || is_rst;    // the byte after an FF: dropped when it is the stuffed 00 or the RSTn code
        if (in && !drop) r.keep_mask |= 1u << j;
        if (is_rst) r.rst_mask |= 1u << j;
        prev = b;
    }
    return r;
}

// Finds (image, chunk) of a workgroup from an exclusive prefix table.
__device__ __forceinline__ uint32_t find_image(const uint32_t* __restrict__ base, uint32_t nimg, uint32_t wg)
{
    uint32_t lo = 0, hi = nimg;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (base[mid] <= wg) lo = mid; else hi = mid; }
    return lo;
}

__global__ void __launch_bounds__(US_THREADS) k_unstuff_count(const JsImage* __restrict__ imgs, const uint32_t* __restrict__ us_base, uint32_t nimg,
                                                              const uint8_t* __restrict__ raw, uint32_t* __restrict__ chunk_keep, uint32_t* __restrict__ chunk_rst)
{
    // (a launch may cover a sub-range of the batch's images: imgs / us_base then point at its first image, the prefix values stay the
    // batch's own, and the first workgroup of the launch is workgroup us_base[0] of the batch -- the same in every kernel below)
    const uint32_t wg = blockIdx.x + us_base[0];
    const uint32_t img = find_image(us_base, nimg, wg);
    const JsImage& im = imgs[img];
    const uint64_t s = im.file_off + im.scan_start, e = s + im.scan_len;
    const uint64_t o16 = (s & ~15ull) + (uint64_t)(wg - us_base[img]) * US_CHUNK + threadIdx.x * 16;
    const UsBytes c = us_classify(raw, o16, s, e);
    uint32_t nk = __popc(c.keep_mask), nr = __popc(c.rst_mask);
    for (int off = 32; off > 0; off >>= 1) { nk += __shfl_down(nk, off); nr += __shfl_down(nr, off); }
    __shared__ uint32_t sk[US_THREADS / 64], sr[US_THREADS / 64];
    if ((threadIdx.x & 63) == 0) { sk[threadIdx.x >> 6] = nk; sr[threadIdx.x >> 6] = nr; }
    __syncthreads();
    if (threadIdx.x == 0) { chunk_keep[wg] = sk[0] + sk[1] + sk[2] + sk[3]; chunk_rst[wg] = sr[0] + sr[1] + sr[2] + sr[3]; }
}

// One workgroup per image: exclusive scan over its chunks (in place), totals into the side block,
// interval table entry 0 and the end sentinel.
__global__ void __launch_bounds__(256) k_unstuff_scan(const JsImage* __restrict__ imgs, const uint32_t* __restrict__ us_base,
                                                      uint32_t* __restrict__ chunk_keep, uint32_t* __restrict__ chunk_rst,
                                                      uint32_t* __restrict__ seg_tab, uint32_t* __restrict__ side, uint32_t* __restrict__ flags)
{
    const uint32_t img = blockIdx.x; const JsImage& im = imgs[img];
    const uint32_t c0 = us_base[img], nc = us_base[img + 1] - c0;
    __shared__ uint32_t s_k[256], s_r[256]; __shared__ uint32_t run_k, run_r;
    if (threadIdx.x == 0) { run_k = 0; run_r = 0; }
    __syncthreads();
    for (uint32_t b = 0; b < nc; b += 256) {
        const uint32_t i = b + threadIdx.x;
        uint32_t k = i < nc ? chunk_keep[c0 + i] : 0, r = i < nc ? chunk_rst[c0 + i] : 0;
        s_k[threadIdx.x] = k; s_r[threadIdx.x] = r; __syncthreads();
        for (uint32_t d = 1; d < 256; d <<= 1) {                           // Hillis-Steele inclusive scan
            uint32_t ak = threadIdx.x >= d ? s_k[threadIdx.x - d] : 0, ar = threadIdx.x >= d ? s_r[threadIdx.x - d] : 0;
            __syncthreads(); s_k[threadIdx.x] += ak; s_r[threadIdx.x] += ar; __syncthreads();
        }
        if (i < nc) { chunk_keep[c0 + i] = run_k + s_k[threadIdx.x] - k; chunk_rst[c0 + i] = run_r + s_r[threadIdx.x] - r; }
        __syncthreads();
        if (threadIdx.x == 255) { run_k += s_k[255]; run_r += s_r[255]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        uint32_t* sd = side + im.side_off; uint32_t* st = seg_tab + im.seg_off;
        sd[10] = run_k; sd[11] = run_r + 1;
        st[0] = 0;
        if (run_r + 2 <= im.seg_cap) st[run_r + 1] = run_k; else { FLAG_OR(flags, img, F_OVERRUN); ANOM_MIN(flags, img, ANOM_KEY(0u, AK_MIRROR)); }
    }
}

// Chained scan state of the fused un-stuffing pass: one 64-bit word per chunk, self-contained (a reader needs nothing else, so plain relaxed
// device-scope atomics carry it -- no fence, no cache write-back):  [63:56] epoch of the decode that wrote it (1..255; the arena is cleared
// at upload and every decode rewrites every word, so a word of another epoch is simply "not there yet"), [55:54] 1 = the chunk's own counts,
// 2 = the inclusive prefix over the image's chunks up to and including it, [53:32] RSTn markers (saturating: anything near 2^22 is far
// beyond every interval table and ends in the overflow flag), [31:0] kept bytes.
#define US_ST_AGG 1u
#define US_ST_INC 2u
__device__ __forceinline__ uint64_t us_pack(uint32_t epoch, uint32_t kind, uint32_t keep, uint32_t rst)
{ return ((uint64_t)epoch << 56) | ((uint64_t)kind << 54) | ((uint64_t)min(rst, 0x3FFFFFu) << 32) | keep; }
__device__ __forceinline__ uint32_t us_sat_add(uint32_t a, uint32_t b) { return min(a + b, 0x3FFFFFu); }

// Decoupled look-back of the fused un-stuffing passes (one wave): the sums (kept bytes, RSTn markers) over the `ci` state words in front of
// word `wg`, all of the same image.  64 words per trip: lane j reads word wg - back - j, and the wave waits until every word between itself and
// the closest inclusive prefix has been published in this decode's epoch.
__device__ __forceinline__ void us_lookback(unsigned long long* __restrict__ us_state, uint32_t wg, uint32_t ci, uint32_t epoch, uint32_t lane, uint32_t& ek, uint32_t& er)
{
    ek = 0; er = 0;
    if (!ci) return;
    uint32_t back = 1;                                           // distance of lane 0's word
    for (;;) {
        const bool valid = back + lane <= ci;
        uint64_t v = 0; bool ready = false;
        for (;;) {
            if (valid) { v = __hip_atomic_load(&us_state[wg - back - lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ready = (uint32_t)(v >> 56) == epoch && ((v >> 54) & 3u) != 0u; }
            const uint64_t m_inc = WBALLOT(valid && ready && ((v >> 54) & 3u) == US_ST_INC), m_wait = WBALLOT(valid && !ready);
            const uint64_t first_inc = m_inc & (0 - m_inc), below = m_inc ? first_inc - 1ull : ~0ull;   // lanes closer than the closest inclusive prefix (all, if there is none)
            if (!(m_wait & (below | first_inc))) {
                const bool take = valid && (((1ull << lane) & (below | first_inc)) != 0ull);
                uint32_t tk = take ? (uint32_t)v : 0u, tr = take ? (uint32_t)(v >> 32) & 0x3FFFFFu : 0u;
                for (int off = 32; off > 0; off >>= 1) { tk += __shfl_xor(tk, off); tr = us_sat_add(tr, __shfl_xor(tr, off)); }
                ek += tk; er = us_sat_add(er, tr);
                back = m_inc ? 0u : back + 64u;                  // 0: done
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (back == 0 || back > ci) break;                       // (past the image's first word without an inclusive prefix cannot happen: word 0 publishes one)
    }
}

// The un-stuffing pass proper: classifies the chunk's bytes, finds where its kept bytes go, writes them (and the interval table entries of its
// RSTn markers).  FUSED (the decode): ONE pass over the file bytes -- the chunk's place in its image comes from a decoupled look-back over the
// chunks before it (us_state; chunks are taken by ticket, so every chunk a workgroup waits for is running or done), the last
// chunk of an image also leaves the totals where k_unstuff_scan used to (side block words 10 / 11, interval 0 and the end sentinel), and the
// exclusive prefixes are kept in chunk_keep / chunk_rst for the side passes.  !FUSED: prefixes are read from chunk_keep / chunk_rst
// (k_unstuff_count + k_unstuff_scan before it: the three-pass form, kept as a cross-check; and the side passes).
// Side-output pass (us_out != nullptr, grid = the chunks of one image starting at workgroup wg0): instead of writing the
// stream again, every thread records the compacted-stream index of its first kept byte -- the inverse map
// "compacted byte -> file offset" that the MCU file map needs (k_side_maps).
template <bool FUSED>
__global__ void __launch_bounds__(US_THREADS) k_unstuff_write(const JsImage* __restrict__ imgs, const uint32_t* __restrict__ us_base, uint32_t nimg,
                                                              const uint8_t* __restrict__ raw, uint32_t* __restrict__ chunk_keep,
                                                              uint32_t* __restrict__ chunk_rst, uint8_t* __restrict__ ustr, uint32_t* __restrict__ seg_tab,
                                                              uint32_t wg0, uint32_t* __restrict__ us_out,
                                                              unsigned long long* __restrict__ us_state, uint32_t epoch, uint32_t* __restrict__ side, uint32_t* __restrict__ flags,
                                                              uint32_t* __restrict__ ticket, uint32_t ticket_base)
{
    // FUSED: which chunk a workgroup takes is decided by the order in which the workgroups START (a ticket), not by blockIdx.x: the look-back below waits
    // for the chunks in front of its own, and a chunk with a smaller ticket is held by a workgroup that is running or done -- whatever order the
    // hardware dispatches block indices in, and whatever else shares the device (two streams of a split decode, other processes).  The counter is
    // never cleared between decodes: the host hands in its value before this launch.
    __shared__ uint32_t s_bid;
    if (FUSED) { if (threadIdx.x == 0) s_bid = atomicAdd(ticket, 1u) - ticket_base; __syncthreads(); }
    const uint32_t bid = FUSED ? s_bid : blockIdx.x;
    const uint32_t wg = bid + wg0 + us_base[0];
    const uint32_t img = find_image(us_base, nimg, wg);
    const JsImage& im = imgs[img];
    const uint64_t s = im.file_off + im.scan_start, e = s + im.scan_len;
    const uint64_t o16 = (s & ~15ull) + (uint64_t)(wg - us_base[img]) * US_CHUNK + threadIdx.x * 16;
    const UsBytes c = us_classify(raw, o16, s, e);
    const uint32_t nk = __popc(c.keep_mask), nr = __popc(c.rst_mask), lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t pk = nk, pr = nr;                                             // inclusive wave scans
    for (int off = 1; off < 64; off <<= 1) { uint32_t a = __shfl_up(pk, off), b = __shfl_up(pr, off); if (lane >= (uint32_t)off) { pk += a; pr += b; } }
    __shared__ uint32_t wk[US_THREADS / 64], wr[US_THREADS / 64];
    __shared__ uint32_t s_excl[2];
    if (lane == 63) { wk[wave] = pk; wr[wave] = pr; }
    __syncthreads();
    const uint32_t agg_k = wk[0] + wk[1] + wk[2] + wk[3], agg_r = us_sat_add(us_sat_add(wr[0], wr[1]), us_sat_add(wr[2], wr[3]));
    const uint32_t ci = wg - us_base[img], nc = us_base[img + 1] - us_base[img];
    // FUSED: the chunk's own counts go out FIRST (the chunks behind this one may be waiting for them) ...
    if (FUSED && ci && threadIdx.x == 0) __hip_atomic_store(&us_state[wg], us_pack(epoch, US_ST_AGG, agg_k, agg_r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t lk = pk - nk, lr = pr - nr;                                  // exclusive prefixes of this thread INSIDE the chunk
    for (uint32_t w = 0; w < wave; w++) { lk += wk[w]; lr += wr[w]; }
    // ... then the kept bytes of the chunk are gathered in LDS (chunk-relative: where the chunk lands in the stream is not needed for that), and only
    // then does the first wave look back for the chunk's base: by now the chunks before it have long published, the look-back costs its reads.
    __shared__ __attribute__((aligned(4))) uint8_t s_out[US_CHUNK + 8];
    if (!us_out && c.keep_mask) {
        const uint4 v = *reinterpret_cast<const uint4*>(raw + o16);
        const uint32_t w4[4] = { v.x, v.y, v.z, v.w };
        uint32_t lo = lk;
        #pragma unroll
        for (int j = 0; j < 16; j++) if (c.keep_mask & (1u << j)) s_out[lo++] = (uint8_t)(w4[j >> 2] >> ((j & 3) * 8));
    }
    uint32_t bk, br;
    if (FUSED) {
        if (wave == 0) {
            uint32_t ek = 0, er = 0;
            us_lookback(us_state, wg, ci, epoch, lane, ek, er);
            if (lane == 0) {
                __hip_atomic_store(&us_state[wg], us_pack(epoch, US_ST_INC, ek + agg_k, us_sat_add(er, agg_r)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_excl[0] = ek; s_excl[1] = er;
                chunk_keep[wg] = ek; chunk_rst[wg] = er;               // (the side passes of this decode read them)
                if (ci + 1 == nc) {                                      // the image's totals: un-stuffed length, intervals, interval 0 and the end sentinel
                    const uint32_t run_k = ek + agg_k, run_r = us_sat_add(er, agg_r);
                    uint32_t* sd = side + im.side_off; uint32_t* st = seg_tab + im.seg_off;
                    sd[10] = run_k; sd[11] = run_r + 1;
                    st[0] = 0;
                    if (run_r + 2 <= im.seg_cap) st[run_r + 1] = run_k; else { FLAG_OR(flags, img, F_OVERRUN); ANOM_MIN(flags, img, ANOM_KEY(0u, AK_MIRROR)); }
                }
            }
        }
        __syncthreads();
        bk = s_excl[0]; br = s_excl[1];
    } else { bk = chunk_keep[wg]; br = chunk_rst[wg]; __syncthreads(); }
    const uint32_t cbase_out = bk, phase = cbase_out & 3u;               // the image's stream starts 16-byte aligned
    if (us_out) { us_out[blockIdx.x * US_THREADS + threadIdx.x] = bk + lk; return; }
    if (c.rst_mask) {                                                      // interval table: interval `seg` starts at the next kept byte behind its marker
        uint32_t* st = seg_tab + im.seg_off; uint32_t seg = br + lr;
        #pragma unroll
        for (int j = 0; j < 16; j++) if (c.rst_mask & (1u << j)) { seg++; if (seg + 1 < im.seg_cap) st[seg] = cbase_out + lk + __popc(c.keep_mask & ((1u << j) - 1u)); }
    }
    const uint32_t total = agg_k;                                          // bytes this chunk keeps
    if (!total) return;
    // The chunk leaves as whole aligned 32-bit words of the stream: output word w holds the gathered bytes [4w - phase, 4w - phase + 4)
    // (phase = chunk base & 3: two LDS words and a byte alignment); only the ragged first / last word goes out byte by byte.
    uint8_t* dst = ustr + im.ustr_off + (cbase_out - phase);               // 4-byte aligned
    const uint32_t lo_b = phase, hi_b = phase + total;                     // valid byte range inside dst
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(s_out);
    for (uint32_t w = threadIdx.x; w * 4 < hi_b; w += US_THREADS) {
        if (w * 4 >= lo_b && w * 4 + 4 <= hi_b) reinterpret_cast<uint32_t*>(dst)[w] = phase ? __builtin_amdgcn_alignbyte(s32[w], s32[w - 1], 4u - phase) : s32[w];
        else for (uint32_t b = max(w * 4, lo_b); b < min(w * 4 + 4, hi_b); b++) dst[b] = s_out[b - phase];
    }
}

// The un-stuffing pass of large jobs (sub-sequences of 128 bytes and more): ONE pass from the file bytes to the word-interleaved sub-sequence layout
// (phys_word) -- no linear copy, no transposition pass.  A workgroup takes a SUPER-chunk of four consecutive 4 KiB chunks of one image (four 16-byte
// loads per thread in flight: a 4 KiB workgroup lived on one), scans them as k_unstuff_write does, publishes its counts, gathers up to 16 KiB of kept
// bytes in LDS (one pad word per sub-sequence length, so that the transposed read-out below walks the banks), looks back for its base, and writes the
// bytes where the walks read them: for every word row w of the sub-sequences it touches, the words of consecutive sub-sequences are consecutive in
// memory (33 x 4 bytes per row for 512-byte pieces).  The exclusive prefixes of its 4 KiB chunks are kept in chunk_keep / chunk_rst for the side
// passes (k_unstuff_write<false>); us4_base: the images' prefix over super-chunks (the state words are indexed by super-chunk).
#define US_SUPER 4
template <int WL>
__global__ void __launch_bounds__(US_THREADS) k_unstuff_fused(const JsImage* __restrict__ imgs, const uint32_t* __restrict__ us_base, const uint32_t* __restrict__ us4_base, uint32_t nimg,
                                                              const uint8_t* __restrict__ raw, uint32_t* __restrict__ chunk_keep, uint32_t* __restrict__ chunk_rst,
                                                              uint8_t* __restrict__ ustr, uint32_t* __restrict__ seg_tab, unsigned long long* __restrict__ us_state, uint32_t epoch,
                                                              uint32_t* __restrict__ side, uint32_t* __restrict__ flags, uint32_t* __restrict__ ticket, uint32_t ticket_base)
{
    constexpr uint32_t PADSH = WL + 2;                            // one pad word per 4 << WL gathered bytes
    __shared__ __attribute__((aligned(4))) uint8_t s_out[US_SUPER * US_CHUNK + ((US_SUPER * US_CHUNK) >> WL) + 16];
    __shared__ uint32_t wk[US_SUPER][US_THREADS / 64], wr[US_SUPER][US_THREADS / 64];
    __shared__ uint32_t s_excl[2];
    __shared__ uint32_t s_bid;                                    // the super-chunk is taken by ticket (k_unstuff_write): the look-back only ever waits for workgroups that have started
    if (threadIdx.x == 0) s_bid = atomicAdd(ticket, 1u) - ticket_base;
    __syncthreads();
    const uint32_t sw = s_bid + us4_base[0];
    const uint32_t img = find_image(us4_base, nimg, sw);
    const JsImage& im = imgs[img];
    const uint32_t sc = sw - us4_base[img], nsc = us4_base[img + 1] - us4_base[img];
    const uint32_t c0 = sc * US_SUPER, nc = us_base[img + 1] - us_base[img], wgc = us_base[img] + c0;     // first 4 KiB chunk of the super-chunk: in the image / in the batch
    const uint64_t s = im.file_off + im.scan_start, e = s + im.scan_len;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    UsBytes c[US_SUPER]; uint4 v[US_SUPER]; uint32_t pk[US_SUPER], pr[US_SUPER];
    #pragma unroll
    for (uint32_t q = 0; q < US_SUPER; q++) {
        v[q] = make_uint4(0u, 0u, 0u, 0u); c[q].keep_mask = 0; c[q].rst_mask = 0;
        if (c0 + q < nc) c[q] = us_classify(raw, (s & ~15ull) + (uint64_t)(c0 + q) * US_CHUNK + threadIdx.x * 16, s, e, &v[q]);
    }
    #pragma unroll
    for (uint32_t q = 0; q < US_SUPER; q++) {                    // inclusive wave scans, kept bytes in the low half, markers in the high half (<= 1024 / <= 512 per wave)
        uint32_t x = __popc(c[q].keep_mask) | (__popc(c[q].rst_mask) << 16);
        for (int off = 1; off < 64; off <<= 1) { const uint32_t a = __shfl_up(x, off); if (lane >= (uint32_t)off) x += a; }
        pk[q] = x & 0xFFFFu; pr[q] = x >> 16;
        if (lane == 63) { wk[q][wave] = pk[q]; wr[q][wave] = pr[q]; }
    }
    __syncthreads();
    uint32_t agg_k = 0, agg_r = 0, lk[US_SUPER], lr[US_SUPER], ck[US_SUPER], cr[US_SUPER];       // exclusive prefixes inside the super-chunk: of the thread (lk, lr), of the 4 KiB chunk (ck, cr)
    #pragma unroll
    for (uint32_t q = 0; q < US_SUPER; q++) {
        ck[q] = agg_k; cr[q] = agg_r;
        lk[q] = agg_k + pk[q] - __popc(c[q].keep_mask); lr[q] = agg_r + pr[q] - __popc(c[q].rst_mask);
        #pragma unroll
        for (uint32_t w = 0; w < US_THREADS / 64; w++) { if (w < wave) { lk[q] += wk[q][w]; lr[q] += wr[q][w]; } agg_k += wk[q][w]; agg_r = us_sat_add(agg_r, wr[q][w]); }
    }
    if (sc && threadIdx.x == 0) __hip_atomic_store(&us_state[sw], us_pack(epoch, US_ST_AGG, agg_k, agg_r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    #pragma unroll
    for (uint32_t q = 0; q < US_SUPER; q++) if (c[q].keep_mask) {
        const uint32_t w4[4] = { v[q].x, v[q].y, v[q].z, v[q].w };
        uint32_t g = lk[q];
        #pragma unroll
        for (int j = 0; j < 16; j++) if (c[q].keep_mask & (1u << j)) { s_out[g + ((g >> PADSH) << 2)] = (uint8_t)(w4[j >> 2] >> ((j & 3) * 8)); g++; }
    }
    if (wave == 0) {
        uint32_t ek, er;
        us_lookback(us_state, sw, sc, epoch, lane, ek, er);
        if (lane == 0) {
            __hip_atomic_store(&us_state[sw], us_pack(epoch, US_ST_INC, ek + agg_k, us_sat_add(er, agg_r)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_excl[0] = ek; s_excl[1] = er;
            if (sc + 1 == nsc) {                                     // the image's totals: un-stuffed length, intervals, interval 0 and the end sentinel
                const uint32_t run_k = ek + agg_k, run_r = us_sat_add(er, agg_r);
                uint32_t* sd = side + im.side_off; uint32_t* st = seg_tab + im.seg_off;
                sd[10] = run_k; sd[11] = run_r + 1;
                st[0] = 0;
                if (run_r + 2 <= im.seg_cap) st[run_r + 1] = run_k; else { FLAG_OR(flags, img, F_OVERRUN); ANOM_MIN(flags, img, ANOM_KEY(0u, AK_MIRROR)); }
            }
        }
    }
    __syncthreads();
    const uint32_t cb = s_excl[0], br = s_excl[1], phase = cb & 3u, total = agg_k;
    if (threadIdx.x < US_SUPER && c0 + threadIdx.x < nc) {         // (the side passes of this decode read them)
        uint32_t k_ = 0, r_ = 0;
        #pragma unroll
        for (uint32_t q = 0; q < US_SUPER; q++) if (q == threadIdx.x) { k_ = ck[q]; r_ = cr[q]; }
        chunk_keep[wgc + threadIdx.x] = cb + k_; chunk_rst[wgc + threadIdx.x] = us_sat_add(br, r_);
    }
    #pragma unroll
    for (uint32_t q = 0; q < US_SUPER; q++) if (c[q].rst_mask) {    // interval table: interval `seg` starts at the next kept byte behind its marker
        uint32_t* st = seg_tab + im.seg_off; uint32_t seg = br + lr[q];
        #pragma unroll
        for (int j = 0; j < 16; j++) if (c[q].rst_mask & (1u << j)) { seg++; if (seg + 1 < im.seg_cap) st[seg] = cb + lk[q] + __popc(c[q].keep_mask & ((1u << j) - 1u)); }
    }
    if (!total) return;
    // read-out: stream word W = W0 + j holds the gathered bytes [4j - phase, 4j - phase + 4); it goes to phys_word(W).  Rows of the transposition:
    // for word index w inside a sub-sequence, the sub-sequences l0 .. l0 + nl - 1 this super-chunk touches -- consecutive lanes, consecutive words.
    const uint32_t W0 = cb >> 2, Wend = (cb + total + 3u) >> 2, jl = (phase + total) >> 2, jf = phase ? 1u : 0u;     // full words: j in [jf, jl)
    const uint32_t l0 = W0 >> WL, nl = ((Wend - 1u) >> WL) - l0 + 1u;
    uint32_t* dst32 = reinterpret_cast<uint32_t*>(ustr + im.ustr_off); uint8_t* dst8 = ustr + im.ustr_off;
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(s_out);
    const uint32_t dq = US_THREADS / nl, dr = US_THREADS % nl;                  // (w, l) of item i = w * nl + l, i = tid, tid + 256, ...: all lanes busy whatever nl is
    for (uint32_t w = threadIdx.x / nl, l = threadIdx.x % nl; w < (1u << WL); w += dq, l += dr) {
            if (l >= nl) { l -= nl; w++; if (w >> WL) break; }
            const uint32_t W = ((l0 + l) << WL) + w;
            if (W < W0 || W >= Wend) continue;
            const uint32_t j = W - W0, pw = phys_word<WL>(W);
            if (j >= jf && j < jl) dst32[pw] = phase ? __builtin_amdgcn_alignbyte(s32[j + (j >> WL)], s32[(j - 1u) + ((j - 1u) >> WL)], 4u - phase) : s32[j + (j >> WL)];
            else for (uint32_t b = max(4u * j, phase); b < min(4u * j + 4u, phase + total); b++) { const uint32_t g = b - phase; dst8[(pw << 2) + (b & 3u)] = s_out[g + ((g >> PADSH) << 2)]; }
        }
}

// Linear compacted stream -> word-interleaved sub-sequence layout (see phys_word).  One workgroup per group of 64
// sub-sequences: the group's bytes are read linearly (coalesced) into LDS and leave as 256-byte rows (word w of the
// 64 sub-sequences), so both sides of the permutation stream at full line width.
template <int WL>
__global__ void __launch_bounds__(256) k_interleave(const JsImage* __restrict__ imgs, const uint32_t* __restrict__ sy_base, uint32_t nimg,
                                                    const uint32_t* __restrict__ side, const uint8_t* __restrict__ lin, uint8_t* __restrict__ ustr)
{
    constexpr uint32_t WPS = 1u << WL, GW = 64u * WPS;               // words per sub-sequence / per group
    __shared__ uint32_t s_w[64 * (WPS + 1)];                          // +1: the transposed read walks a column without bank conflicts
    const uint32_t bx = blockIdx.x + sy_base[0] * 4;
    const uint32_t img = find_image(sy_base, nimg, bx / 4);
    const JsImage& im = imgs[img];
    const uint32_t g = bx - sy_base[img] * 4;                         // group index inside the image (4 groups per 256 sub-sequences)
    if (g * 64 >= im.n_subseq) return;
    const uint32_t len_words = (side[im.side_off + 10] + 3) / 4 + 4;  // un-stuffed length (+ cursor look-ahead)
    if (g * GW >= len_words) return;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(lin + im.ustr_off) + (size_t)g * GW;
    uint32_t* dst = reinterpret_cast<uint32_t*>(ustr + im.ustr_off) + (size_t)g * GW;
    for (uint32_t i = threadIdx.x; i < GW; i += 256) s_w[(i >> WL) * (WPS + 1) + (i & (WPS - 1))] = src[i];   // i = l*WPS + w
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < GW; i += 256) dst[i] = s_w[(i & 63u) * (WPS + 1) + (i >> 6)];          // i = w*64 + l
}

// ---- Huffman symbol walk shared by the sync and write passes ------------------------------
// Per-workgroup LDS view of one image's decode tables (dynamic shared memory, sized by the batch).
struct SubTabs {
    const uint16_t* lut1;              // n_rows x 2048 entries
    const uint16_t* lut2;              // second level (codes longer than JS_L1_BITS bits)
    const uint32_t* lutp;              // state-only pair entries (k_sync only; nullptr elsewhere)
    const uint32_t* lut2p;             // ... and the single-symbol entries behind their escapes
    uint32_t rb0, rb1, rb2;            // per component: byte offset of its DC row in lutp | of its AC row << 16
    const uint32_t* qz;                // 3 x 64: quantiser entry (zig-zag order) | natural index of that position << 16 -- one read per coefficient
    uint32_t rows01, rows2;            // per component 16 bits: first-level row of its DC table | AC table << 8
    uint32_t n1, n2, nb;               // block-in-MCU index where Cb / Cr blocks start; blocks per MCU
};

// Write-pass view: DC tables as 16-bit single-symbol rows, AC tables as 32-bit value-pair rows (both from JsTableSet::lutw), the shared
// second level, the q|zz table.  tabw = DC rows | AC rows << 8 of the largest table set of the batch.
struct WriteTabs {
    const char* rows;                  // DC rows, then AC rows
    const uint16_t* lut2;
    const uint32_t* qz;
    uint32_t wb0, wb1, wb2;            // per component: byte offset of its DC row | of its AC row << 16 (from `rows`)
};
__host__ __device__ __forceinline__ size_t wtabs_bytes(uint32_t tabw, uint32_t tab_lut2)
{ return (size_t)(tabw & 255u) * (2u << JS_L1_BITS) + (size_t)(tabw >> 8) * (4u << JS_L1_BITS) + (((size_t)tab_lut2 * 2 + 15) & ~15ull) + 3 * 64 * 4; }
__device__ __forceinline__ void load_wtabs(WriteTabs& W, uint8_t* lds, const JsTableSet& ts, uint32_t tabw, uint32_t tab_lut2, uint32_t ncomp, uint32_t tid, uint32_t nthreads)
{
    const uint32_t dc_bytes = (tabw & 255u) * (2u << JS_L1_BITS), ac_bytes = (tabw >> 8) * (4u << JS_L1_BITS);   // DC rows: 16-bit entries (low halves of JsTableSet::lutw), AC rows: 32-bit pair entries
    uint16_t* l2 = reinterpret_cast<uint16_t*>(lds + dc_bytes + ac_bytes);
    uint32_t* q = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(l2) + (((size_t)tab_lut2 * 2 + 15) & ~15ull));
    uint32_t off[6];                                             // byte offset of the table of slot (comp-1)*2 + class
    for (uint32_t slot = 0; slot < 6; slot++) {
        const uint32_t row = ts.slot_row[slot], sub = ts.row_sub[row];
        if (slot >= ncomp * 2) { off[slot] = 0; continue; }
        if (slot & 1) {
            off[slot] = dc_bytes + sub * (4u << JS_L1_BITS);
            uint32_t* dst = reinterpret_cast<uint32_t*>(lds + off[slot]);
            for (uint32_t i = tid; i < (1u << JS_L1_BITS); i += nthreads) dst[i] = ts.lutw[row][i];        // (shared rows are copied once per slot: same bytes)
        } else {                                                 // DC rows: the 16-bit form of the same entries (code length | size << 4; bit 15: escape)
            off[slot] = sub * (2u << JS_L1_BITS);
            uint16_t* dst = reinterpret_cast<uint16_t*>(lds + off[slot]);
            for (uint32_t i = tid; i < (1u << JS_L1_BITS); i += nthreads) dst[i] = (uint16_t)ts.lutw[row][i];
        }
    }
    for (uint32_t i = tid; i < ts.lut2_used; i += nthreads) l2[i] = ts.lut2[i];
    for (uint32_t i = tid; i < 3 * 64; i += nthreads) q[i] = (uint32_t)(&ts.qzz[0][0])[i] | ((uint32_t)c_zigzag[i & 63u] << 16);
    W.rows = reinterpret_cast<const char*>(lds); W.lut2 = l2; W.qz = q;
    W.wb0 = off[0] | (off[1] << 16); W.wb1 = off[2] | (off[3] << 16); W.wb2 = off[4] | (off[5] << 16);
}

// PAIRS (every user today: k_sync and the candidate kernels): LDS holds the state-only pair rows and their second level ONLY -- 2 KiB per row.  The single-symbol
// tables (lut1 / lut2) are read where the batch's table sets lie in global memory: only walk_sync's careful step (an interval ends, a code matches nothing)
// looks there, once per restart interval or so.  Round 6: a k_sync workgroup is a chain of ~2900 dependent steps (one workgroup alone on the chip 0.65 ms,
// seven per CU 0.79 ms each: the kernel is the number of workgroup rounds times that latency, profiles/r06_experiments.txt 16); without the 5.4 KiB of tables
// nothing reads in the common step ten workgroups fit a CU where seven did.
template <bool PAIRS>
__device__ __forceinline__ void load_subtabs(SubTabs& T, uint8_t* lds, const JsImage& im, const JsTableSet& ts, uint32_t tab_rows, uint32_t tab_lut2,
                                             uint32_t tid, uint32_t nthreads)
{
    T.lutp = nullptr; T.lut2p = nullptr; T.rb0 = T.rb1 = T.rb2 = 0;
    if (!PAIRS) {
        uint16_t* l1 = reinterpret_cast<uint16_t*>(lds);
        uint16_t* l2 = l1 + (size_t)tab_rows * (1u << JS_L1_BITS);
        uint32_t* q = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(l2) + (((size_t)tab_lut2 * 2 + 15) & ~15ull));
        const uint32_t* src1 = reinterpret_cast<const uint32_t*>(&ts.lut1[0][0]);
        uint32_t* dst1 = reinterpret_cast<uint32_t*>(l1);
        for (uint32_t i = tid; i < ts.n_rows * (1u << JS_L1_BITS) / 2; i += nthreads) dst1[i] = src1[i];
        for (uint32_t i = tid; i < ts.lut2_used; i += nthreads) l2[i] = ts.lut2[i];
        for (uint32_t i = tid; i < 3 * 64; i += nthreads) q[i] = (uint32_t)(&ts.qzz[0][0])[i] | ((uint32_t)c_zigzag[i & 63u] << 16);
        T.lut1 = l1; T.lut2 = l2; T.qz = q;
    } else {
        T.lut1 = &ts.lut1[0][0]; T.lut2 = &ts.lut2[0]; T.qz = nullptr;
        uint32_t* lp = reinterpret_cast<uint32_t*>(lds);
        uint32_t* lp2 = lp + (size_t)tab_rows * (1u << JS_L1_BITS);
        const uint32_t* srcp = &ts.lutp[0][0];
        for (uint32_t i = tid; i < ts.n_rows * (1u << JS_L1_BITS); i += nthreads) lp[i] = srcp[i];
        for (uint32_t i = tid; i < ts.lut2_used; i += nthreads) lp2[i] = ts.lut2p[i];
        T.lutp = lp; T.lut2p = lp2;
        const uint32_t rsh = JS_L1_BITS + 2;                     // a row of lutp is 4 << JS_L1_BITS bytes
        T.rb0 = (ts.slot_row[0] << rsh) | (ts.slot_row[1] << (rsh + 16)); T.rb1 = (ts.slot_row[2] << rsh) | (ts.slot_row[3] << (rsh + 16));
        T.rb2 = (ts.slot_row[4] << rsh) | (ts.slot_row[5] << (rsh + 16));
    }
    T.rows01 = ts.slot_row[0] | (ts.slot_row[1] << 8) | (ts.slot_row[2] << 16) | (ts.slot_row[3] << 24); T.rows2 = ts.slot_row[4] | (ts.slot_row[5] << 8);
    T.nb = im.blk_per_mcu; T.n1 = im.samp_h[1] * im.samp_v[1]; T.n2 = im.ncomp == 3 ? T.n1 + im.samp_h[2] * im.samp_v[2] : T.nb;
}

struct Cursor {                        // MSB-first bit cursor: two byte-swapped words, one raw word prefetched.  `sh` = 32 - (bits of w0 already
    const uint32_t* words; uint32_t widx, w0, w1, nxt; int32_t sh; uint32_t p;   // consumed), kept in [0, 31]: at least one bit of w0 is always consumed,
    uint32_t poff;                     // so the window is ONE v_alignbit_b32 with no special case (sh == 0: the window is w1 itself)
};                                     // poff: byte offset of word `widx` in the interleaved layout, carried along (see cur_fetch)
__device__ __forceinline__ uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }
// The word behind `nxt`.  Its place in the interleaved layout is not computed from the word index every time (phys_word: six vector instructions, four of them at
// half rate, in a step that every walker of the chip is bound by -- ~15 % of k_sync's vector cycles): consecutive words of a sub-sequence lie one 256-byte row apart,
// the offset is stepped, and computed afresh only where a sub-sequence ends (a lane crosses that once, at the end of its walk).
template <int WL> __device__ __forceinline__ uint32_t cur_fetch(Cursor& c)
{
    const uint32_t v = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(c.words) + c.poff);
    c.widx++;
    if (WL == 4) c.poff += 4u;
    else { c.poff += 256u; if ((c.widx & ((1u << WL) - 1u)) == 0u) c.poff = phys_word<WL>(c.widx) << 2; }
    return v;
}
template <int WL> __device__ __forceinline__ void cur_init(Cursor& c, const uint32_t* words, uint32_t p)
{
    const uint32_t wi = p ? ((p - 1u) >> 5) + 1u : 0u;           // index of w1; w0 is the word before it (nothing before bit 0)
    c.words = words; c.p = p; c.sh = (int32_t)(31u - ((p - 1u) & 31u));
    c.w0 = p ? bswap32(words[phys_word<WL>(wi - 1u)]) : 0u; c.w1 = bswap32(words[phys_word<WL>(wi)]);
    c.nxt = words[phys_word<WL>(wi + 1u)]; c.widx = wi + 2u; c.poff = phys_word<WL>(wi + 2u) << 2;
}
// the next 32 bits of the stream
__device__ __forceinline__ uint32_t cur_peek(const Cursor& c) { return __builtin_amdgcn_alignbit(c.w0, c.w1, (uint32_t)c.sh); }
template <int WL> __device__ __forceinline__ void cur_skip(Cursor& c, uint32_t n)      // n <= 32
{
    c.sh -= (int32_t)n; c.p += n;
    if (c.sh < 0) { c.sh += 32; c.w0 = c.w1; c.w1 = bswap32(c.nxt); c.nxt = cur_fetch<WL>(c); }
}

// state word: [31:12] interval index (up to 2^20 - 1 restart intervals), [11:6] block-in-MCU (< 48), [5:0] next coefficient index (0 = DC)
#define ST_SEG(s) ((s) >> 12)
#define ST_C(s)   (((s) >> 6) & 63u)
#define ST_K(s)   ((s) & 63u)
#define ST_MAKE(seg, c, k) (((seg) << 12) | ((c) << 6) | (k))
#define P_END 0xFFFFFFFFu
#define WR_STRIDE 64                   // int16 per thread-private LDS block buffer: 32 KiB per workgroup + the decode tables (DC rows 1 KiB,
                                       // AC pair rows 2 KiB each) fit a CU four times

// component (0..2) of block-in-MCU index c, without a table lookup
__device__ __forceinline__ uint32_t comp_of(const SubTabs& T, uint32_t c) { return (c >= T.n1 ? 1u : 0u) + (c >= T.n2 ? 1u : 0u); }
__device__ __forceinline__ uint32_t rows_of(const SubTabs& T, uint32_t comp) { return comp == 2 ? T.rows2 : (T.rows01 >> (comp * 16u)) & 0xFFFFu; }

// One symbol: table entry for the 32-bit window `win` under the DC (k == 0) or AC table of `rp`.
__device__ __forceinline__ uint32_t sym_lookup(const SubTabs& T, uint32_t win, uint32_t rp, uint32_t k)
{
    const uint32_t row = (k ? rp >> 8 : rp) & 255u;
    uint32_t e = T.lut1[(row << JS_L1_BITS) + (win >> (32 - JS_L1_BITS))];
    if (__builtin_expect(e & 0x8000u, 0)) {                      // a few % of symbols: code longer than JS_L1_BITS bits
        const uint32_t nbx = (e >> 12) & 7u;
        e = T.lut2[(e & 0xFFFu) + ((win >> (32 - JS_L1_BITS - nbx)) & ((1u << nbx) - 1u))];
    }
    return e;
}

// The restart mark of MCU m: 0 = none, j + 1 (low six bits) = the DC predictors are cleared in front of block j of the MCU.  One byte per MCU, set by
// compare-and-swap on its word (lanes of different sub-sequences may meet markers in neighbouring MCUs); false when the MCU already carries
// a different mark (two markers inside one MCU: hostile, left to the mirror) -- except for the pair that a stray RSTn shortly behind a regular
// one makes: bit 6 = "and in front of block 0 as well" beside a mark j + 1 > 1 (the regular marker on the MCU boundary, the stray one further in).
// Bit 7 of a mark: the marker was met INSIDE the block (after its DC symbol), not in front of it -- the predictors are cleared in front of the block
// either way (the DC scan looks at the low seven bits), but when the reference's decode ENDS in that block (ANOM_KEY) a mark the walk set inside it may
// stem from bits the reference never read, while one set in front of it cannot: "in front" wins when both set the same mark.
__device__ __forceinline__ bool mark_reset(uint8_t* __restrict__ mcu_rst, uint32_t m, uint32_t v, bool inside)
{
    uint32_t* w = reinterpret_cast<uint32_t*>(mcu_rst + (m & ~3u)); const uint32_t sh = (m & 3u) * 8u;      // (the arena is 16-byte aligned per image)
    uint32_t old = *reinterpret_cast<volatile uint32_t*>(w);
    for (;;) {
        const uint32_t cur = (old >> sh) & 255u;
        if ((cur & 63u) == v) { if (!inside && (cur & 128u)) atomicAnd(w, ~(128u << sh)); return true; }
        uint32_t nv;
        if (cur == 0u) nv = v | (inside ? 128u : 0u);
        else if (v == 1u && (cur & 63u) > 1u && !inside) { if (cur & 64u) return true; nv = cur | 64u; }        // the boundary mark joins one further in
        else if (v > 1u && cur == 1u) nv = v | 64u | (inside ? 128u : 0u);                                       // a mark further in joins the (plain) boundary mark
        else return false;
        const uint32_t seen = atomicCAS(w, old, (old & ~(255u << sh)) | (nv << sh));
        if (seen == old) return true;
        old = seen;
    }
}

// What the reference sees as RSV_RST_TERM (:1167-1176) -- no code fits in what is left of the interval --
// or a code that matches nothing.  Returns WS_OVER when the walk is over (end of the entropy data), WS_BAD_CODE when a code that matches
// nothing ended the block the reference's way, else WS_GO_ON.
// A code that matches nothing, far from any marker (the reader's 32-bit register holds no restart marker's shadow: 64 bits of margin):
// ReadScanVal consumes ONE bit and reports RSV_UNDERFLOW (:1178-1186, :1270-1282), DecodeScanComp gives the block up there and then
// without running the IDCT (:1737-1757, m_afIdctBlock still holds DecodeIdctClear's zeros :2243) -- the block keeps its DC difference
// if it had decoded one, its AC part counts as zero, and the next block starts at the next bit.  The walks do exactly that (the write
// pass empties the block's AC part before it leaves); what the flag then stands for is bookkeeping (scan_bad, messages, the histogram
// slot of "one bit"), which the mirror's side-only pass produces on request.  Closer to an interval end the outcome depends on what the
// reader's look-ahead has already seen: F_BAD_EDGE, the mirror takes over as before.
enum { WS_OVER = 0, WS_GO_ON = 1, WS_BAD_CODE = 2 };
template <bool WRITE, int WL>
__device__ __forceinline__ int walk_slow(const JsImage& im, const uint32_t* __restrict__ words, const uint32_t* __restrict__ st, uint32_t nseg,
                                       Cursor& cur, uint32_t len, uint32_t& seg, uint32_t& seg_end, uint32_t& c, uint32_t& k,
                                       uint32_t blk, bool mark, uint8_t* __restrict__ mcu_rst, uint32_t& flags, uint32_t& anom, bool spec = false)
{
    const uint32_t remain = seg_end > cur.p ? seg_end - cur.p : 0u;
    // What the reference does with "no code here" depends on whether its look-ahead has met the marker behind the interval (m_bRestartRead):
    // BuffTopup :1292-1323 runs right in front of every match and loads whole bytes while eight bits of the register are vacant, so the RSTn has
    // been met exactly when at most 24 bits of the interval are left -- then "nothing fits" IS the end of the interval (RSV_RST_TERM :1167-1176),
    // with 25 or more it is a code that matches nothing (one bit consumed, :1178-1186).  In the LAST interval no RSTn follows (the reference reads
    // on through whatever marker ends the scan): there the margin of 64 bits stays, and anything closer is the mirror's.
    const bool more = seg + 1 < nseg;
    if (len == 0 && remain >= (more ? 25u : 16u)) {
        // No code matches although a whole code could still fit: a corrupt stream -- or simply a speculative walk that is not synchronised yet.
        const bool native = more || remain >= 64;
        if (WRITE && blk < im.total_blocks) { flags |= native ? F_BAD_CODE : (F_BAD_CODE | F_BAD_EDGE); if (!native) anom = min(anom, ANOM_KEY(blk, AK_MIRROR)); }
        cur_skip<WL>(cur, 1);
        return native ? WS_BAD_CODE : WS_GO_ON;
    }
    if (seg + 1 < nseg) {
        // The reference's restart is marker-driven and happens INSIDE DecodeScanComp (:1644-1680): whatever is left of the interval is dropped,
        // the three DC predictors are cleared, and the block in progress simply goes on with the new interval's bits -- coefficient index and
        // position in the MCU are kept.  On an MCU boundary that is the well-formed case; anywhere else (a damaged interval that lost or gained
        // blocks) the walks follow all the same: the reset is recorded for the block in progress (block-in-MCU index + 1 in the MCU's mark; the
        // DC scan clears its sums in front of that block), and F_RST_MISALIGN then only stands for bookkeeping (messages).  Two resets inside one
        // MCU other than {on its boundary, further in} (mark_reset), or two markers back to back (the reference meets the second one inside its
        // retry and files the DC value under index 1): F_BAD_EDGE.
        if (WRITE) {
            // An interval that is left before ONE bit of it was consumed -- a few bits that hold no code, e.g. FF FF kept as data between two markers: the
            // reference entered it inside the retry of :1644-1680, and that retry's RSV_RST_TERM is not handled as a restart (the ASSERT of :1676): the
            // symbol counts as a coefficient and the restart happens one read later -- like two markers back to back (below), the mirror's
            if (seg > 0 && cur.p == st[seg] * 8 && blk < im.total_blocks) { flags |= F_RST_MISALIGN | F_BAD_EDGE; anom = min(anom, ANOM_KEY(blk, AK_MIRROR)); }
            if (k != 0 || c != 0 || remain >= 8) flags |= F_RST_MISALIGN;                       // well-formed: < 8 pad bits, on an MCU boundary
            if (mark && blk < im.total_blocks && !mark_reset(mcu_rst, blk / im.blk_per_mcu, c + 1u, k != 0u)) { flags |= F_BAD_EDGE; anom = min(anom, ANOM_KEY(blk, AK_MIRROR)); }
        }
        seg++; const uint32_t np = seg_end; seg_end = st[seg + 1] * 8;
        if (spec) { c = 0; k = 0; }                              // a speculative walk (its exit state is only a guess): in a well-formed stream an MCU starts here
        if (WRITE && seg_end == np && seg + 1 < nseg) { flags |= F_RST_MISALIGN | F_BAD_EDGE; anom = min(anom, ANOM_KEY(blk, AK_MIRROR)); }       // back-to-back RSTn
        cur_init<WL>(cur, words, np);
        return WS_GO_ON;
    }
    if (WRITE && blk < im.total_blocks) { flags |= F_SHORT; anom = min(anom, ANOM_KEY(blk, AK_MIRROR)); }
    cur.p = P_END; c = 0; k = 0; seg = 0;
    return WS_OVER;
}


// SYNC flavour: state only.  Walks the symbols that start inside [entry position, own_end).
// The lanes of a wave step together (one table entry per lane and step); everything a lane rarely needs -- a code longer than
// the first-level window, the end of a restart interval, a code that matches nothing, the end of its range -- sits behind a wave-level
// vote, and leaves the lane in a state in which the straight-line code of the step does nothing for it: the common step has no
// "is this lane still going" selects.  Votes are compares written into scalar pairs and combined there; a lane reads its bit back as
// a predicate.  a_ctab: LDS address of the per-block-of-the-MCU table {address of its DC row, address of its AC row} in lutp.
// MID (the memo walks of the candidate form): the state at the first symbol boundary at or behind bit `mid_bits` is reported too -- {position,
// state word, blocks completed so far} in mid3 -- so that the write pass can put a second lane on the second half of the sub-sequence; a pair
// of symbols is not taken across that bit (the lane of the first half stops at the first boundary behind it: both must name the same one).
template <int WL, bool MID = false>
__device__ __forceinline__ void walk_sync(const JsImage& im, const SubTabs T, uint32_t a_ctab, const uint32_t* __restrict__ words, const uint32_t* __restrict__ st,
                                          uint32_t nseg, uint32_t total_bits, uint32_t own_end, uint32_t& p_io, uint32_t& s_io, uint32_t& nblk_out,
                                          uint32_t mid_bits = 0, uint32_t* mid3 = nullptr, bool spec = false)
{
    uint32_t seg = ST_SEG(s_io), c = ST_C(s_io), k = ST_K(s_io), nblk = 0, fl = 0;
    if (p_io == P_END || (p_io >= total_bits && seg + 1 >= nseg)) { p_io = P_END; s_io = 0; nblk_out = 0; if (MID) { mid3[0] = P_END; mid3[1] = 0; mid3[2] = 0; } return; }
    uint32_t mid_p = 0, mid_s = 0, mid_n = 0; uint64_t m_midc = 0ull;            // MID: lanes whose mid state is taken
    const uint32_t mid_lim = min(mid_bits, own_end);
    uint32_t seg_end = st[seg + 1] * 8;
    Cursor cur; cur_init<WL>(cur, words, p_io);
    const uint32_t a_lp = lds_addr(T.lutp), a_l2p = lds_addr(T.lut2p);
    uint2 ct = lds_r64(a_ctab + (c < T.nb ? c : 0u) * 8u);
    uint32_t row = k ? ct.y : ct.x;                              // LDS address of the table row the next symbol is read from
    uint32_t res_p = cur.p, res_s = s_io, res_n = 0;
    uint64_t m_live = WBALLOT(true);                             // lanes still inside their range
    for (;;) {
        if (MID) {
            const uint64_t m_m = WBALLOT(cur.p >= mid_bits) & m_live & ~m_midc;
            if (m_m) { if (__builtin_amdgcn_inverse_ballot_w64(m_m)) { mid_p = cur.p; mid_s = cur.p == P_END ? 0u : ST_MAKE(seg, c, k); mid_n = nblk; } m_midc |= m_m; }
        }
        // a lane that has left its range reports the state it left with; from then on it computes on whatever it holds
        const uint64_t m_act = m_live & WBALLOT(cur.p < own_end);
        if (m_act != m_live) {
            if (__builtin_amdgcn_inverse_ballot_w64(m_live & ~m_act)) { res_p = cur.p; res_s = cur.p == P_END ? 0u : ST_MAKE(seg, c, k); res_n = nblk; }
            m_live = m_act;
        }
        if (!m_act) break;
        const uint32_t win = cur_peek(cur);
        // One table entry describes the symbol at the cursor and, where its code was visible in the same window, the AC symbol
        // behind it: byte 0 = bits of symbol 1 (code + value), byte 1 = its advance of the coefficient index (64 for EOB: ends the
        // block from any index), byte 2 = bits of both (0: no second symbol), byte 3 = index advance of both.
        uint32_t pe = lds_r32(row + ((win >> (32 - JS_L1_BITS)) << 2));
        // A code longer than the window: a few % of symbols, but SOME lane of the wave holds one nearly every step.  One read of
        // the second level replaces the entry by a single-symbol one and the lane stays on the common path.
        const uint64_t m_esc = WBALLOT((int32_t)pe < (int32_t)0xC0000000u) & m_act;       // top bits 10
        if (m_esc) {
            if (__builtin_amdgcn_inverse_ballot_w64(m_esc)) { const uint32_t nbx = (pe >> 12) & 7u; pe = lds_r32(a_l2p + (((pe & 0xFFFu) + __builtin_amdgcn_ubfe(win, 32u - JS_L1_BITS - nbx, nbx)) << 2)); }
        }
        uint32_t b1 = pe & 255u, k1 = k + ((pe >> 8) & 255u);
        const uint32_t b12 = (pe >> 16) & 255u;
        // both symbols together when the first one does not end the block and the second one starts inside this lane's own range
        uint64_t m_two = WBALLOT(b12 != 0u) & WBALLOT(k1 < 64u) & WBALLOT(cur.p + b1 < (MID && !__builtin_amdgcn_inverse_ballot_w64(m_midc) ? mid_lim : own_end));
        const uint64_t m_slow = (WBALLOT((int32_t)pe < 0) | WBALLOT(cur.p + (__builtin_amdgcn_inverse_ballot_w64(m_two) ? b12 : b1) > seg_end)) & m_act;
        if (m_slow) {
            if (__builtin_amdgcn_inverse_ballot_w64(m_slow)) {   // no code here, or the end of the restart interval / of the data is near:
                const uint32_t lrow = (row - a_lp) >> (JS_L1_BITS + 2);                     // one symbol the careful way
                const uint32_t e = sym_lookup(T, win, lrow, 0u);
                const uint32_t len = (e >> 8) & 31u, run = (e >> 4) & 15u, size = e & 15u;
                if (len == 0 || cur.p + len > seg_end) {
                    const int ws = walk_slow<false, WL>(im, words, st, nseg, cur, len, seg, seg_end, c, k, 0, false, nullptr, fl, fl, spec);   // end of the data: p = P_END
                    if (ws == WS_BAD_CODE) { k = 0u; c = c + 1 == T.nb ? 0u : c + 1; nblk++; }                                        // the block ends with the bad code
                } else {
                    cur_skip<WL>(cur, len + size);
                    const uint32_t kn = k == 0 ? 1u : k + run + 1u;
                    const bool dn = k != 0 && ((e & 255u) == 0 || kn >= 64u);
                    k = dn ? 0u : kn;
                    if (dn) { c = c + 1 == T.nb ? 0u : c + 1; nblk++; }
                }
                ct = lds_r64(a_ctab + c * 8u);
                b1 = 0; k1 = k;                                  // the rest of the step does nothing for this lane
            }
            m_two &= ~m_slow;
        }
        const bool two = __builtin_amdgcn_inverse_ballot_w64(m_two);
        { const uint32_t adv = two ? b12 : b1; cur.sh -= (int32_t)adv; cur.p += adv; }
        if (__builtin_amdgcn_inverse_ballot_w64(WBALLOT(cur.sh < 0) & m_act)) {          // (lanes out of their range must not load)
            cur.sh += 32; cur.w0 = cur.w1; cur.w1 = bswap32(cur.nxt);
            cur.nxt = cur_fetch<WL>(cur);
        }
        const uint32_t kn = two ? k + (pe >> 24) : k1;
        const uint64_t m_dn = WBALLOT(kn >= 64u) & m_act;
        k = __builtin_amdgcn_inverse_ballot_w64(m_dn) ? 0u : kn;
        if (m_dn) {
            if (__builtin_amdgcn_inverse_ballot_w64(m_dn)) { c = c + 1 == T.nb ? 0u : c + 1; nblk++; ct = lds_r64(a_ctab + c * 8u); }
        }
        row = k ? ct.y : ct.x;
    }
    p_io = res_p; s_io = res_s; nblk_out = res_n;
    if (MID) {                                                   // (a lane whose range ends before the middle: its exit state)
        const bool got = __builtin_amdgcn_inverse_ballot_w64(m_midc);
        mid3[0] = got ? mid_p : res_p; mid3[1] = got ? mid_s : res_s; mid3[2] = got ? mid_n : res_n;
    }
}

// upper_bound(seg table, byte) - 1 : the interval a speculative start position lies in
__device__ __forceinline__ uint32_t find_interval(const uint32_t* __restrict__ st, uint32_t nseg, uint32_t byte)
{
    uint32_t lo = 0, hi = nseg;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (st[mid] <= byte) lo = mid; else hi = mid; }
    return lo;
}

// sub-sequence state arrays (SoA, one u32 each per sub-sequence slot)
struct SubArrays { uint32_t *out_p, *out_s, *in_p, *in_s, *nblk, *base; };

template <int WL>
__global__ void __launch_bounds__(SY_THREADS) k_sync(const JsImage* __restrict__ imgs, const uint32_t* __restrict__ sy_base, uint32_t nimg,
                                                     const JsTableSet* __restrict__ tables, const uint8_t* __restrict__ ustr,
                                                     const uint32_t* __restrict__ seg_tab, const uint32_t* __restrict__ side, SubArrays A, int first_pass,
                                                     uint32_t tab_rows, uint32_t tab_lut2, uint32_t it_max /* rounds at most (the list rounds of k_sync_round follow); 0: to the fixed point */)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    __shared__ uint32_t s_inp[SY_THREADS], s_ins[SY_THREADS], s_outp[SY_THREADS], s_outs[SY_THREADS], s_nblk[SY_THREADS];
    __shared__ uint16_t s_act[SY_THREADS];
    __shared__ uint32_t s_wcount[SY_THREADS / 64];
    __shared__ int s_changed;
    __shared__ __attribute__((aligned(8))) uint2 s_ctab[JS_MAX_BLK_PER_MCU];   // per block of the MCU: LDS address of its DC row, of its AC row (lutp)
    const uint32_t bx = blockIdx.x + sy_base[0];
    const uint32_t img = find_image(sy_base, nimg, bx);
    const JsImage& im = imgs[img];
    if (!tables[im.tableset].lut_ok) return;
    const uint32_t* sd = side + im.side_off;
    const uint32_t total_bits = sd[10] * 8, nseg = min(sd[11], im.seg_cap - 1);
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // Threads 2..255 own one sub-sequence each; threads 1 and 0 walk the TWO before the workgroup's first, in the first launch only: thread 0 the tail of its
    // sub-sequence in the first iteration, thread 1 the tail of its own and then -- like every owned lane -- the whole of it from thread 0's exit state.  A
    // guess that comes out of a whole sub-sequence walked from a guess fails when the tail walk AND 4 Kbit of re-synchronisation fail: the later launches
    // (which carry the true states across workgroup boundaries) find nothing to do.  (One halo lane that walked a tail only was wrong in a few % of the 20 000
    // workgroups of a large batch: every second launch had workgroups that re-loaded their tables and walked again, 0.25 ms of a 0.3 ms launch.)
    const uint32_t own0 = (bx - sy_base[img]) * (SY_THREADS - SY_HALO);
    if (own0 * SUB_BITS >= total_bits && own0) return;          // whole workgroup lies past the end of the data
    const uint32_t i_last = total_bits ? (total_bits - 1u) / SUB_BITS : 0u;     // the last sub-sequence that holds data
    const size_t g0 = im.subseq_off + own0;                      // slot of the first owned sub-sequence
    const bool halo = t < SY_HALO;
    const uint32_t i = own0 + t - SY_HALO;                       // this thread's sub-sequence (the halo threads of the first workgroup: none)
    const bool valid = halo ? own0 != 0 : i < im.n_subseq;
    // Later launches only carry exit states across workgroup boundaries: if the state entering this workgroup is the
    // one its first sub-sequence was last walked from, the whole workgroup is already at its fixed point.
    // The left neighbour writes those slots at the end of this same launch, unordered: ONE thread reads them and the whole
    // workgroup follows its verdict (waves that disagreed would leave the others with a half-loaded table).
    if (first_pass == 0) {
        if (t == 0) { const uint32_t lp = own0 ? A.out_p[g0 - 1] : 0u, ls = own0 ? A.out_s[g0 - 1] : 0u; s_changed = (lp == A.in_p[g0] && ls == A.in_s[g0]) ? 0 : 1; }
        __syncthreads();
        const int go = s_changed;
        __syncthreads();
        if (!go) return;
    } else if (first_pass == 2) {
        // Verification mode (after the candidate chain, k_cand_*): the arrays hold a chain that is a fixed point except at the
        // sub-sequences marked with an entry state nothing equals -- every thread checks its own link.
        bool open = false;
        if (valid && !halo) { const size_t gq = g0 + t - SY_HALO; const uint32_t lp = i ? A.out_p[gq - 1] : 0u, ls = i ? A.out_s[gq - 1] : 0u; open = lp != A.in_p[gq] || ls != A.in_s[gq]; }
        if (!__syncthreads_or(open ? 1 : 0)) return;
    }
    SubTabs T; load_subtabs<true>(T, s_dyn, im, tables[im.tableset], tab_rows, tab_lut2, t, SY_THREADS);
    if (t < T.nb) { const uint32_t rbc = t < T.n1 ? T.rb0 : (t < T.n2 ? T.rb1 : T.rb2); s_ctab[t] = make_uint2(lds_addr(T.lutp) + (rbc & 0xFFFFu), lds_addr(T.lutp) + (rbc >> 16)); }
    const uint32_t a_ctab = lds_addr(s_ctab);                    // (visible after the barrier below)
    const uint32_t* words = reinterpret_cast<const uint32_t*>(ustr + im.ustr_off);
    const uint32_t* st = seg_tab + im.seg_off;
    const size_t gs = g0 + t - SY_HALO;                          // this thread's slot (not for the halo threads of the first workgroup)
    const bool spec_pass = first_pass == 1;
    if (spec_pass || !valid) { s_inp[t] = 0xFFFFFFFEu; s_ins[t] = 0; s_outp[t] = valid ? 0u : P_END; s_outs[t] = 0; s_nblk[t] = 0; }
    else if (halo) { s_inp[t] = 0xFFFFFFFEu; s_ins[t] = 0; s_outp[t] = A.out_p[gs]; s_outs[t] = A.out_s[gs]; s_nblk[t] = 0; }   // the true state left of the workgroup
    else { s_inp[t] = A.in_p[gs]; s_ins[t] = A.in_s[gs]; s_outp[t] = A.out_p[gs]; s_outs[t] = A.out_s[gs]; s_nblk[t] = A.nblk[gs]; }
    __syncthreads();
    const int it_end = it_max ? (int)it_max : SY_THREADS + 2;
    for (int it = 0; it < it_end; it++) {
        // ---- phase A: which sub-sequences see a new entry state?  (reads last iteration's exit states only)
        uint32_t ip = 0, is = 0;
        bool active;
        if (!valid || (halo && !(spec_pass && (it == 0 || t == SY_HALO - 1)))) active = false;
        else {
            if (spec_pass && it == 0 && i != 0) {
                // Speculative start.  This first walk only has to hand an exit state to the right neighbour (every lane walks
                // again from its true entry state in the next round), so it covers the TAIL of the sub-sequence only: JPEG codes
                // resynchronise within a few hundred bits, and the few lanes whose tail was too short are redone one round later.
                const uint32_t sp = i * SUB_BITS + (SUB_BITS - SYNC_SPEC_TAIL);
                const bool in_data = i * SUB_BITS < total_bits;
                ip = !in_data ? P_END : (sp < total_bits ? sp : i * SUB_BITS);
                is = in_data ? ST_MAKE(find_interval(st, nseg, ip / 8), 0u, 0u) : 0u;
            } else if (i == 0) { ip = 0; is = 0; }               // true start of the scan: interval 0, block 0, DC
            else {
                // a sub-sequence behind the last one that holds data owns no symbol: the state passes through, so it takes the exit state of
                // that last one directly (handed on one sub-sequence per round, a scan that ends on the last bit of a byte -- one image in
                // eight -- kept its last workgroup busy for up to 63 rounds more)
                const uint32_t tl = (i > i_last + 1u && i_last >= own0) ? i_last - own0 + SY_HALO : t - 1u;
                ip = s_outp[tl]; is = s_outs[tl];
            }
            active = ip != s_inp[t] || is != s_ins[t];
        }
        if (t == 0) s_changed = 0;
        const uint64_t bal = WBALLOT(active);
        if (lane == 0) s_wcount[wave] = (uint32_t)__builtin_popcountll(bal);
        __syncthreads();                                         // everybody has read the old exit states
        if (active) {
            s_inp[t] = ip; s_ins[t] = is;
            uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            for (uint32_t w = 0; w < wave; w++) r += s_wcount[w];
            s_act[r] = (uint16_t)t;                              // compacted list of sub-sequences to walk again
        }
        uint32_t nact = 0;
        for (uint32_t w = 0; w < SY_THREADS / 64; w++) nact += s_wcount[w];
        __syncthreads();
        // ---- phase B: the first `nact` threads each walk one of them (whole waves drop out as the chain converges)
        if (t < nact) {
            const uint32_t u = s_act[t], iu = own0 + u - SY_HALO;
            uint32_t p = s_inp[u], s = s_ins[u], nblk = 0;
            const uint32_t own_end = min((iu + 1) * SUB_BITS, total_bits);
            if (!(p != P_END && p >= own_end))                   // else: owns no symbol, the state passes through
                walk_sync<WL>(im, T, a_ctab, words, st, nseg, total_bits, own_end, p, s, nblk, 0u, nullptr, spec_pass && it == 0);
            if (p != s_outp[u] || s != s_outs[u]) { s_outp[u] = p; s_outs[u] = s; s_changed = 1; }
            s_nblk[u] = nblk;
        }
        __syncthreads();
        if (!s_changed) break;
        __syncthreads();
    }
    if (valid && !halo) { A.out_p[gs] = s_outp[t]; A.out_s[gs] = s_outs[t]; A.in_p[gs] = s_inp[t]; A.in_s[gs] = s_ins[t]; A.nblk[gs] = s_nblk[t]; }
}

// ---- list rounds (round 6): the large-job form of the rounds behind the first two.
// A k_sync workgroup is a chain of dependent steps -- one alone on the chip takes 0.65 ms, seven per CU 0.79 ms each (profiles/r06_experiments.txt 16): the
// kernel's time is its workgroup rounds times that latency, and a workgroup holds its place (LDS, four waves) through rounds in which a third, a twentieth, a
// two-hundredth of its lanes walk.  So the first launch stops after the tail walks and ONE whole walk (it_max = 2), k_sync_links lists per image the
// sub-sequences whose entry state is not the exit state of their left neighbour, and k_sync_round walks a list with dense waves (256 entries per workgroup,
// workgroups without entries leave at once) and lists the right neighbours of the sub-sequences whose exit state changed -- one launch per round, the chip
// shared by whatever still has work.  Links across workgroup boundaries are links like any other here.  Same fixed point as k_sync's own rounds (walk_sync is a
// function of the entry state); what a bounded number of rounds leaves open, the verification mode of k_sync behind them closes.
#define SYR_SLOTS JS_SYR_SLOTS          // list counters per image: round r reads [r], appends to [r + 1]
__device__ __forceinline__ void list_append(bool put, uint32_t value, uint32_t* __restrict__ list, uint32_t* __restrict__ counter, uint32_t cap)
{
    const uint64_t m = WBALLOT(put);
    if (!m) return;
    const uint32_t lane = threadIdx.x & 63u, n = (uint32_t)__builtin_popcountll(m);
    uint32_t base = 0;
    if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(counter, n);
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(m));
    const uint32_t at = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (put && at < cap) list[at] = value;
}
template <int WL>
__global__ void __launch_bounds__(256) k_sync_links(const JsImage* __restrict__ imgs, const uint32_t* __restrict__ sy_base, uint32_t nimg, const JsTableSet* __restrict__ tables,
                                                    const uint32_t* __restrict__ side, SubArrays A, uint32_t* __restrict__ list, uint32_t* __restrict__ rcnt)
{
    const uint32_t wg = blockIdx.x + sy_base[0];
    const uint32_t img = find_image(sy_base, nimg, wg);
    const JsImage& im = imgs[img];
    if (!tables[im.tableset].lut_ok) return;
    const uint32_t total_bits = side[im.side_off + 10] * 8;
    const uint32_t i = (wg - sy_base[img]) * 256u + threadIdx.x, i_last = total_bits ? (total_bits - 1u) / SUB_BITS : 0u;
    const size_t g = im.subseq_off + i;
    bool open = false;
    if (i >= 1u && i <= i_last && i < im.n_subseq) open = A.out_p[g - 1] != A.in_p[g] || A.out_s[g - 1] != A.in_s[g];
    list_append(open, i, list + im.subseq_off, rcnt + (size_t)img * SYR_SLOTS, im.n_subseq);
}
template <int WL>
__global__ void __launch_bounds__(SY_THREADS) k_sync_round(const JsImage* __restrict__ imgs, const uint32_t* __restrict__ sy_base, uint32_t nimg,
                                                           const JsTableSet* __restrict__ tables, const uint8_t* __restrict__ ustr, const uint32_t* __restrict__ seg_tab,
                                                           const uint32_t* __restrict__ side, SubArrays A, const uint32_t* __restrict__ lin, uint32_t* __restrict__ lout,
                                                           uint32_t* __restrict__ rcnt, uint32_t r, uint32_t tab_rows, uint32_t tab_lut2)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    __shared__ __attribute__((aligned(8))) uint2 s_ctab[JS_MAX_BLK_PER_MCU];
    const uint32_t wg = blockIdx.x + sy_base[0];
    const uint32_t img = find_image(sy_base, nimg, wg);
    const JsImage& im = imgs[img];
    if (!tables[im.tableset].lut_ok) return;
    uint32_t* cnt = rcnt + (size_t)img * SYR_SLOTS;
    const uint32_t n_ent = min(cnt[r], im.n_subseq), j = wg - sy_base[img], wgs_img = sy_base[img + 1] - sy_base[img], t = threadIdx.x;
    if (j * SY_THREADS >= n_ent) return;
    const uint32_t* sd = side + im.side_off;
    const uint32_t total_bits = sd[10] * 8, nseg = min(sd[11], im.seg_cap - 1), i_last = total_bits ? (total_bits - 1u) / SUB_BITS : 0u;
    SubTabs T; load_subtabs<true>(T, s_dyn, im, tables[im.tableset], tab_rows, tab_lut2, t, SY_THREADS);
    if (t < T.nb) { const uint32_t rbc = t < T.n1 ? T.rb0 : (t < T.n2 ? T.rb1 : T.rb2); s_ctab[t] = make_uint2(lds_addr(T.lutp) + (rbc & 0xFFFFu), lds_addr(T.lutp) + (rbc >> 16)); }
    __syncthreads();
    const uint32_t* words = reinterpret_cast<const uint32_t*>(ustr + im.ustr_off);
    const uint32_t* st = seg_tab + im.seg_off;
    const uint32_t* li = lin + im.subseq_off; uint32_t* lo = lout + im.subseq_off;
    for (uint32_t e0 = j * SY_THREADS; e0 < n_ent; e0 += wgs_img * SY_THREADS) {
        const uint32_t e = e0 + t;
        bool changed = false; uint32_t i = 0;
        if (e < n_ent) {
            i = li[e];
            const size_t g = im.subseq_off + i;
            // (the left neighbour may be walked in this very launch: whichever of its exit states is read here, a changed one lists this sub-sequence again)
            const uint32_t ip = A.out_p[g - 1], is = A.out_s[g - 1];
            if (ip != A.in_p[g] || is != A.in_s[g]) {
                uint32_t p = ip, s = is, nblk = 0;
                const uint32_t own_end = min((i + 1u) * SUB_BITS, total_bits);
                if (!(p != P_END && p >= own_end)) walk_sync<WL>(im, T, lds_addr(s_ctab), words, st, nseg, total_bits, own_end, p, s, nblk);
                changed = p != A.out_p[g] || s != A.out_s[g];
                A.in_p[g] = ip; A.in_s[g] = is; A.nblk[g] = nblk;
                if (changed) { A.out_p[g] = p; A.out_s[g] = s; }
            }
        }
        if (r + 1u < SYR_SLOTS) list_append(changed && i + 1u <= i_last, i + 1u, lo, cnt + r + 1u, im.n_subseq);
    }
}

// =====================================================================================
//  Candidate synchronisation: the small-job form of stage (2).
//
//  k_sync's rounds are a serial chain -- a sub-sequence learns its true entry state only after its left neighbour has walked from
//  ITS true entry state -- and in a 4:2:0 stream the chain is long: a walk that has the bit position right but the block-in-MCU index
//  wrong keeps disagreeing with its neighbours for thousands of bits (luma and chroma blocks share code prefixes; nothing tells a
//  walker which block of the MCU it is in).  A batch of a thousand images hides that behind its other images; ONE image waits for
//  it (24 rounds in a workgroup of the 3840x2160 picture of BASELINE config 2 = 0.56 of its 0.94 ms).  When the job is small enough
//  that the chip has lanes to spare, the chain is cut by hypotheses instead (tools/mhsync_sim.c is the CPU model of what follows):
//   k_cand_spec   one walk of every sub-sequence per block-in-MCU index h, from its first bit in state (h, DC): the exit states X[h][i]
//                 are the CANDIDATES for the entry state of sub-sequence i + 1 -- the true one is among them for all but ~1 % of them;
//   k_cand_walk   one walk of sub-sequence i from every distinct candidate: a memo (entry state -> exit state, blocks) of up to six
//                 entries per sub-sequence, and per entry the slot of sub-sequence i + 1's memo that continues it (its exit state IS that
//                 candidate) -- a map over slot numbers, eight bytes per sub-sequence (byte 7: where to go on when the entry state
//                 is in no slot: the exit most speculative walks agree on);
//   k_cand_chain  one workgroup per image: following the chain from the true start of the scan is now a composition of those maps --
//                 a prefix scan (two v_perm_b32 per composition, DPP shifts between the lanes) instead of walks.  Where the chain arrives
//                 at a sub-sequence with a state that is in no slot, the sub-sequence is queued with that state; the same workgroup walks
//                 the queued ones (slot 6 of their memos), patches the maps around them and follows the chain again.  Two such rounds
//                 resolve a typical image (313 and 11 walks for the 34 533 sub-sequences of the 3840x2160 picture);
//   k_cand_apply  writes the selected entry / exit states and block counts where k_block_scan and k_write2 expect them.
//  What is still open after the last round the host allows is marked and left to k_sync in its verification mode,
//  which walks exactly the marked sub-sequences and what depends on them; k_write2 verifies the whole chain in any case.
// =====================================================================================
#define CD_H       6                   // candidate slots = hypotheses (images with more blocks per MCU: k_sync)
#define CD_FILL    6                   // the memo slot a queued walk fills
#define CD_NONE    7
#define CD_SLOTS   7
#define CD_EMPTY   0xFFFFFFFDu         // entry position of an unused memo slot (no state has it: positions are < 2^32 - 3)
#define CD_DIAG_WORDS 12                // per image: diagnostics of the chain (JSNOOP_DEBUG_CAND)
struct CandArrays {
    uint32_t *xp, *xs;                 // [CD_H][n]  speculative exit states
    uint32_t *mep, *mes, *mxp, *mxs, *mnb;   // [CD_SLOTS][n]  memo: entry state, exit state, blocks completed
    uint32_t *mmp, *mms, *mmn;         // [CD_SLOTS][n]  memo: state at the middle of the sub-sequence (position, state word, blocks completed before it)
    uint32_t *hp, *hs, *hn;            // [n]  the middle state of the selected entry (k_cand_apply): where the second write lane of the sub-sequence starts
    uint2* map;                        // [n]  byte e (0..6): slot of the next sub-sequence's memo that holds the exit state of entry e (7: none); byte 7: the guess
    uint8_t *sel;                      // [n]  slot the chain selected
    uint64_t n;
};
__device__ __forceinline__ uint32_t cd_byte(uint2 m, uint32_t e) { return ((e < 4u ? m.x : m.y) >> ((e & 3u) * 8u)) & 255u; }
// (a then b)[e] = b[a[e]]: the bytes of a are selectors into the eight bytes of b
__device__ __forceinline__ uint2 cd_compose(uint2 a, uint2 b) { return make_uint2(__builtin_amdgcn_perm(b.y, b.x, a.x), __builtin_amdgcn_perm(b.y, b.x, a.y)); }
#define CD_IDENT make_uint2(0x03020100u, 0x07060504u)

// common prologue of the grid kernels: image and sub-sequence of this thread (workgroups are dealt to the images by sy_base, 256 sub-sequences each)
#define CD_PROLOGUE \
    const uint32_t wg = blockIdx.x + sy_base[0]; \
    const uint32_t img = find_image(sy_base, nimg, wg); \
    const JsImage& im = imgs[img]; \
    if (!tables[im.tableset].lut_ok) return; \
    const uint32_t* sd = side + im.side_off; \
    const uint32_t total_bits = sd[10] * 8, nseg = min(sd[11], im.seg_cap - 1); \
    const uint32_t t = threadIdx.x, i = (wg - sy_base[img]) * SY_THREADS + t; \
    const bool valid = i < im.n_subseq; \
    const size_t g = im.subseq_off + i, n = C.n; (void)total_bits; (void)nseg; (void)valid; (void)g; (void)n;

template <int WL>
__global__ void __launch_bounds__(SY_THREADS) k_cand_spec(const JsImage* __restrict__ imgs, const uint32_t* __restrict__ sy_base, uint32_t nimg,
                                                          const JsTableSet* __restrict__ tables, const uint8_t* __restrict__ ustr,
                                                          const uint32_t* __restrict__ seg_tab, const uint32_t* __restrict__ side, CandArrays C,
                                                          uint32_t tab_rows, uint32_t tab_lut2)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    __shared__ __attribute__((aligned(8))) uint2 s_ctab[JS_MAX_BLK_PER_MCU];
    CD_PROLOGUE
    const uint32_t h = blockIdx.y;
    if (h >= im.blk_per_mcu) return;
    SubTabs T; load_subtabs<true>(T, s_dyn, im, tables[im.tableset], tab_rows, tab_lut2, t, SY_THREADS);
    if (t < T.nb) { const uint32_t rbc = t < T.n1 ? T.rb0 : (t < T.n2 ? T.rb1 : T.rb2); s_ctab[t] = make_uint2(lds_addr(T.lutp) + (rbc & 0xFFFFu), lds_addr(T.lutp) + (rbc >> 16)); }
    __syncthreads();
    if (!valid) return;
    const uint32_t* words = reinterpret_cast<const uint32_t*>(ustr + im.ustr_off);
    const uint32_t* st = seg_tab + im.seg_off;
    const bool in_data = i * SUB_BITS < total_bits;
    uint32_t p = in_data ? i * SUB_BITS : P_END, s = in_data ? ST_MAKE(find_interval(st, nseg, p / 8), i ? h : 0u, 0u) : 0u, nblk = 0;
    const uint32_t own_end = min((i + 1) * SUB_BITS, total_bits);
    walk_sync<WL>(im, T, lds_addr(s_ctab), words, st, nseg, total_bits, own_end, p, s, nblk, 0u, nullptr, true);      // (speculative: the exit states are candidates)
    C.xp[h * n + g] = p; C.xs[h * n + g] = s;
}

// slot (< nb) of the first speculative exit of sub-sequence slot g that equals (p, s); CD_NONE if none does
__device__ __forceinline__ uint32_t cd_match(const CandArrays& C, size_t g, uint32_t nb, uint32_t p, uint32_t s)
{
    uint32_t r = CD_NONE;
    for (uint32_t q = nb; q-- > 0;) if (C.xp[q * C.n + g] == p && C.xs[q * C.n + g] == s) r = q;
    return r;
}

template <int WL, bool MID>
__global__ void __launch_bounds__(SY_THREADS) k_cand_walk(const JsImage* __restrict__ imgs, const uint32_t* __restrict__ sy_base, uint32_t nimg,
                                                          const JsTableSet* __restrict__ tables, const uint8_t* __restrict__ ustr,
                                                          const uint32_t* __restrict__ seg_tab, const uint32_t* __restrict__ side, CandArrays C,
                                                          uint32_t tab_rows, uint32_t tab_lut2)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    __shared__ __attribute__((aligned(8))) uint2 s_ctab[JS_MAX_BLK_PER_MCU];
    CD_PROLOGUE
    const uint32_t h = blockIdx.y, nb = im.blk_per_mcu;
    uint8_t* mapb = reinterpret_cast<uint8_t*>(C.map);
    // Sub-sequences behind the last one that holds data own no symbol: whatever state arrives passes through (it is P_END unless the last symbol
    // ended on the very last bit).  The chain takes them as resolved -- every map byte points at slot 0 -- and k_cand_apply hands the exit
    // state of the last data sub-sequence through them.
    const uint32_t i_last = total_bits ? (total_bits - 1u) / SUB_BITS : 0u;
    if (valid && i > i_last) { C.mep[h * n + g] = CD_EMPTY; mapb[g * 8 + h] = 0; if (h == 0) { C.mep[CD_FILL * n + g] = CD_EMPTY; mapb[g * 8 + CD_FILL] = 0; mapb[g * 8 + 7] = 0; } }
    if (h >= nb) { if (valid && i <= i_last) { C.mep[h * n + g] = CD_EMPTY; mapb[g * 8 + h] = CD_NONE; } return; }
    SubTabs T; load_subtabs<true>(T, s_dyn, im, tables[im.tableset], tab_rows, tab_lut2, t, SY_THREADS);
    if (t < T.nb) { const uint32_t rbc = t < T.n1 ? T.rb0 : (t < T.n2 ? T.rb1 : T.rb2); s_ctab[t] = make_uint2(lds_addr(T.lutp) + (rbc & 0xFFFFu), lds_addr(T.lutp) + (rbc >> 16)); }
    __syncthreads();
    if (!valid || i > i_last) return;
    const uint32_t* words = reinterpret_cast<const uint32_t*>(ustr + im.ustr_off);
    const uint32_t* st = seg_tab + im.seg_off;
    // entry state: candidate h of the left neighbour (the true start of the scan for the first sub-sequence); a candidate equal to one in a lower slot is walked there
    uint32_t p = 0, s = 0; bool use = h == 0;
    if (i) {
        p = C.xp[h * n + g - 1]; s = C.xs[h * n + g - 1]; use = true;
        for (uint32_t q = 0; q < h; q++) if (C.xp[q * n + g - 1] == p && C.xs[q * n + g - 1] == s) use = false;
    }
    uint32_t succ = CD_NONE;
    if (use) {
        const uint32_t own_end = min((i + 1) * SUB_BITS, total_bits);
        uint32_t xp = p, xs = s, nblk = 0, mid3[3] = { p, s, 0u };                  // (a state that passes through: the middle state is the entry state)
        if (!(xp != P_END && xp >= own_end)) walk_sync<WL, MID>(im, T, lds_addr(s_ctab), words, st, nseg, total_bits, own_end, xp, xs, nblk, i * SUB_BITS + SUB_BITS / 2, mid3);
        C.mep[h * n + g] = p; C.mes[h * n + g] = s; C.mxp[h * n + g] = xp; C.mxs[h * n + g] = xs; C.mnb[h * n + g] = nblk;
        if (MID) { C.mmp[h * n + g] = mid3[0]; C.mms[h * n + g] = mid3[1]; C.mmn[h * n + g] = mid3[2]; }
        if (i + 1 < im.n_subseq) succ = i == i_last ? 0u : cd_match(C, g, nb, xp, xs);
    } else C.mep[h * n + g] = CD_EMPTY;
    mapb[g * 8 + h] = (uint8_t)succ;
    if (h == 0) {
        // where to go on from a sub-sequence whose entry state is in no slot: the exit state most of its speculative walks ended in
        uint32_t best = 0, guess = CD_NONE;
        for (uint32_t q = 0; q < nb; q++) {
            const uint32_t qp = C.xp[q * n + g], qs = C.xs[q * n + g]; uint32_t cnt = 0;
            for (uint32_t r = 0; r < nb; r++) cnt += (C.xp[r * n + g] == qp && C.xs[r * n + g] == qs) ? 1u : 0u;
            if (cnt > best) { best = cnt; guess = q; }
        }
        if (i + 1 >= im.n_subseq) guess = CD_NONE; else if (i == i_last) guess = 0;
        C.mep[CD_FILL * n + g] = CD_EMPTY; mapb[g * 8 + CD_FILL] = CD_NONE; mapb[g * 8 + 7] = (uint8_t)guess;
    }
}

// One workgroup of 8 waves per image: the chain, and the walks it asks for, in rounds until it runs through (or max_rounds walk rounds
// are spent).  Every wave owns a contiguous segment of the image's sub-sequences and goes through it in tiles of 512: a lane takes eight
// consecutive maps (64 contiguous bytes: the wave reads 4 KiB at a stretch, the next tile's already in flight), composes them, the wave scans
// the 64 lane products with shuffles and carries the product of the tiles before along; no barrier inside a segment.  The 8 segment
// products meet in LDS; the second trip over the tiles applies what came before the segment and writes the selections.  A sub-sequence the
// chain reaches with a state in none of its slots, from a left neighbour that HAS a selection, is queued; the first threads of the
// workgroup walk the queued ones (slot 6 of their memos) and patch the two maps around each, and the chain is followed again.
#define CC_THREADS 512                 // (eight waves: the walk inside wants more registers than sixteen waves leave)
#define CC_PER_LANE 8
#define CC_TILE (64 * CC_PER_LANE)
#define CC_REQ_CAP 4096                // walks queued per round and image (a lane takes every 512th); what does not fit is queued again by the next chain
// Scan of the lanes' maps (lane order) without the LDS crossbar: row_shr 1 / 2 / 4 / 8 inside the rows of 16 (a lane no source reaches reads
// the identity map: composing with it changes nothing, so no step needs a select), the four row totals through v_readlane and composed as
// wave-uniform values.
template <int CTRL> __device__ __forceinline__ uint2 cd_dpp(uint2 v)
{
    return make_uint2((uint32_t)__builtin_amdgcn_update_dpp((int)0x03020100, (int)v.x, CTRL, 0xF, 0xF, false),
                      (uint32_t)__builtin_amdgcn_update_dpp((int)0x07060504, (int)v.y, CTRL, 0xF, 0xF, false));
}
template <int LANE> __device__ __forceinline__ uint2 cd_lane(uint2 P) { return make_uint2((uint32_t)__builtin_amdgcn_readlane((int)P.x, LANE), (uint32_t)__builtin_amdgcn_readlane((int)P.y, LANE)); }
struct CdScan { uint2 excl, total; };                            // product of the lanes before this one; product of all 64
__device__ __forceinline__ CdScan cd_scan(uint2 P, uint32_t lane)
{
    P = cd_compose(cd_dpp<0x111>(P), P);                         // row_shr:1
    P = cd_compose(cd_dpp<0x112>(P), P);                         // row_shr:2
    P = cd_compose(cd_dpp<0x114>(P), P);                         // row_shr:4
    P = cd_compose(cd_dpp<0x118>(P), P);                         // row_shr:8 -- inclusive inside each row
    const uint2 S = cd_dpp<0x111>(P);                            // exclusive inside each row (first lane of a row: identity)
    const uint2 t0 = cd_lane<15>(P), t1 = cd_lane<31>(P), t2 = cd_lane<47>(P), t3 = cd_lane<63>(P);
    const uint2 q2 = cd_compose(t0, t1), q3 = cd_compose(q2, t2);
    const uint32_t row = lane >> 4;
    uint2 pre = CD_IDENT;
    if (row == 1) pre = t0; else if (row == 2) pre = q2; else if (row == 3) pre = q3;
    CdScan r; r.excl = cd_compose(pre, S); r.total = cd_compose(q3, t3);
    return r;
}
__device__ __forceinline__ void cd_queue(uint32_t* s_req, uint32_t* s_nreq, uint32_t i, uint32_t prev)
{
    const uint32_t slot = atomicAdd(s_nreq, 1u);
    if (slot < CC_REQ_CAP) s_req[slot] = i | (prev << 24);
}
template <int WL, bool MID>
__global__ void __launch_bounds__(CC_THREADS) k_cand_chain(const JsImage* __restrict__ imgs, const JsTableSet* __restrict__ tables, const uint8_t* __restrict__ ustr,
                                                           const uint32_t* __restrict__ seg_tab, const uint32_t* __restrict__ side, CandArrays C,
                                                           uint32_t* __restrict__ diag_all, int max_rounds, uint32_t tab_rows, uint32_t tab_lut2)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    __shared__ __attribute__((aligned(8))) uint2 s_ctab[JS_MAX_BLK_PER_MCU];
    __shared__ uint2 s_wtot[CC_THREADS / 64];
    __shared__ uint8_t s_wlast[CC_THREADS / 64], s_wfirst_open[CC_THREADS / 64];
    __shared__ uint32_t s_nreq, s_open;
    __shared__ uint32_t s_req[CC_REQ_CAP];
    const uint32_t img = blockIdx.x; const JsImage& im = imgs[img];
    if (!tables[im.tableset].lut_ok) return;
    uint32_t* diag = diag_all + (size_t)img * CD_DIAG_WORDS;     // [0] walks still queued at the end, [4 + r] / [8 + r]: sub-sequences open / walks queued after chain r (r < 4)
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t nsub = im.n_subseq;
    const uint32_t seg = (((nsub + CC_THREADS / 64 - 1) / (CC_THREADS / 64)) + CC_TILE - 1) / CC_TILE * CC_TILE, ntiles = seg / CC_TILE;   // per wave, whole tiles
    const size_t g0 = im.subseq_off, n = C.n;
    const uint2* maps = C.map + g0 + wave * seg + lane * CC_PER_LANE;            // this lane's four maps of tile 0
    const bool has0 = ntiles > 0 && wave * seg < nsub;
    SubTabs T = {}; bool have_tabs = false;
    uint32_t nreq = 0;
    for (int round = 0;; round++) {
        if (t == 0) { s_nreq = 0; s_open = 0; }
        __syncthreads();
        // ---- first trip: R = product of the maps of this wave's segment
        uint2 R = CD_IDENT;
        uint4 nx[CC_PER_LANE / 2];
        #pragma unroll
        for (int q = 0; q < CC_PER_LANE / 2; q++) nx[q] = has0 ? *reinterpret_cast<const uint4*>(maps + 2 * q) : make_uint4(0, 0, 0, 0);
        #pragma unroll 1
        for (uint32_t tl = 0; tl < ntiles && wave * seg + tl * CC_TILE < nsub; tl++) {
            const uint32_t i0 = wave * seg + tl * CC_TILE + lane * CC_PER_LANE;
            uint4 v[CC_PER_LANE / 2];
            #pragma unroll
            for (int q = 0; q < CC_PER_LANE / 2; q++) v[q] = nx[q];
            if (tl + 1 < ntiles && wave * seg + (tl + 1) * CC_TILE < nsub) {
                #pragma unroll
                for (int q = 0; q < CC_PER_LANE / 2; q++) nx[q] = *reinterpret_cast<const uint4*>(maps + (tl + 1) * CC_TILE + 2 * q);
            }
            uint2 L = CD_IDENT;
            #pragma unroll
            for (int q = 0; q < CC_PER_LANE / 2; q++) {
                if (i0 + 2 * q < nsub) L = cd_compose(L, make_uint2(v[q].x, v[q].y));
                if (i0 + 2 * q + 1 < nsub) L = cd_compose(L, make_uint2(v[q].z, v[q].w));
            }
            R = cd_compose(R, cd_scan(L, lane).total);            // product over the 64 lanes, in lane order
        }
        if (lane == 0) s_wtot[wave] = R;
        __syncthreads();
        uint32_t rv = 0;                                         // slot 0 of the first sub-sequence holds the true start of the scan
        for (uint32_t w = 0; w < wave; w++) rv = cd_byte(s_wtot[w], rv);
        // ---- second trip: the selections (rv: the selection of the tile's first sub-sequence)
        uint32_t nopen = 0, tile_prev = CD_NONE; bool first_open = false;
        #pragma unroll
        for (int q = 0; q < CC_PER_LANE / 2; q++) nx[q] = has0 ? *reinterpret_cast<const uint4*>(maps + 2 * q) : make_uint4(0, 0, 0, 0);
        #pragma unroll 1
        for (uint32_t tl = 0; tl < ntiles && wave * seg + tl * CC_TILE < nsub; tl++) {
            const uint32_t i0 = wave * seg + tl * CC_TILE + lane * CC_PER_LANE;
            uint4 v[CC_PER_LANE / 2];
            #pragma unroll
            for (int q = 0; q < CC_PER_LANE / 2; q++) v[q] = nx[q];
            if (tl + 1 < ntiles && wave * seg + (tl + 1) * CC_TILE < nsub) {
                #pragma unroll
                for (int q = 0; q < CC_PER_LANE / 2; q++) nx[q] = *reinterpret_cast<const uint4*>(maps + (tl + 1) * CC_TILE + 2 * q);
            }
            uint2 L = CD_IDENT;
            #pragma unroll
            for (int q = 0; q < CC_PER_LANE / 2; q++) {
                if (i0 + 2 * q < nsub) L = cd_compose(L, make_uint2(v[q].x, v[q].y));
                if (i0 + 2 * q + 1 < nsub) L = cd_compose(L, make_uint2(v[q].z, v[q].w));
            }
            const CdScan sc = cd_scan(L, lane);
            const uint2 X = sc.excl;                             // the product of the lanes before
            uint32_t sv[CC_PER_LANE];                            // the selections of this lane's sub-sequences
            sv[0] = cd_byte(X, rv);
            #pragma unroll
            for (int q = 1; q < CC_PER_LANE; q++) sv[q] = cd_byte((q - 1) & 1 ? make_uint2(v[(q - 1) / 2].z, v[(q - 1) / 2].w) : make_uint2(v[(q - 1) / 2].x, v[(q - 1) / 2].y), sv[q - 1]);
            rv = cd_byte(sc.total, rv);
            uint32_t pk0 = 0, pk1 = 0;
            #pragma unroll
            for (int q = 0; q < 4; q++) { pk0 |= sv[q] << (8 * q); pk1 |= sv[4 + q] << (8 * q); }
            if (i0 < nsub) *reinterpret_cast<uint2*>(C.sel + g0 + i0) = make_uint2(pk0, pk1);   // (a tile may reach past the image's slots -- they are 256-aligned, a tile is 512 --: a lane stores only what starts inside the image; its eight bytes then end inside the image's slots)
            // last selection of the sub-sequences this lane has: the lane after needs it as `prev`
            const uint32_t nval = i0 >= nsub ? 0u : min(nsub - i0, (uint32_t)CC_PER_LANE);
            uint32_t mylast = CD_NONE;
            #pragma unroll
            for (int q = 0; q < CC_PER_LANE; q++) if ((uint32_t)q < nval) mylast = sv[q];
            uint32_t prev = (uint32_t)__shfl_up((int)mylast, 1);
            if (lane == 0) prev = tile_prev;
            #pragma unroll
            for (int q = 0; q < CC_PER_LANE; q++) {
                if ((uint32_t)q < nval && sv[q] == CD_NONE) {
                    nopen++;
                    if (q == 0 && lane == 0 && tl == 0) first_open = true;
                    else if (prev != CD_NONE && i0 + q > 0) cd_queue(s_req, &s_nreq, i0 + q, prev);
                }
                prev = sv[q];
            }
            tile_prev = (uint32_t)__builtin_amdgcn_readlane((int)mylast, 63);
        }
        if (lane == 0) { s_wlast[wave] = (uint8_t)tile_prev; s_wfirst_open[wave] = first_open ? 1 : 0; }
        if (nopen) atomicAdd(&s_open, nopen);
        __syncthreads();
        if (lane == 0 && wave > 0 && s_wfirst_open[wave] && s_wlast[wave - 1] != CD_NONE) cd_queue(s_req, &s_nreq, wave * seg, s_wlast[wave - 1]);   // first sub-sequence of a segment: its left neighbour is the wave before's
        __syncthreads();
        nreq = s_nreq;
        if (t == 0 && round < 4) { diag[4 + round] = s_open; diag[8 + round] = nreq; }
        if (nreq == 0 || round >= max_rounds) break;
        // ---- the queued walks
        if (!have_tabs) {
            load_subtabs<true>(T, s_dyn, im, tables[im.tableset], tab_rows, tab_lut2, t, CC_THREADS);
            if (t < T.nb) { const uint32_t rbc = t < T.n1 ? T.rb0 : (t < T.n2 ? T.rb1 : T.rb2); s_ctab[t] = make_uint2(lds_addr(T.lutp) + (rbc & 0xFFFFu), lds_addr(T.lutp) + (rbc >> 16)); }
            have_tabs = true;
            __syncthreads();
        }
        for (uint32_t r = t; r < min(nreq, (uint32_t)CC_REQ_CAP); r += CC_THREADS) {
            const uint32_t* sd = side + im.side_off;
            const uint32_t total_bits = sd[10] * 8, nseg = min(sd[11], im.seg_cap - 1);
            const uint32_t i = s_req[r] & 0xFFFFFFu, prev = s_req[r] >> 24;
            const size_t g = g0 + i;
            const uint32_t p = C.mxp[prev * n + g - 1], s = C.mxs[prev * n + g - 1];
            const uint32_t* words = reinterpret_cast<const uint32_t*>(ustr + im.ustr_off);
            const uint32_t* st = seg_tab + im.seg_off;
            const uint32_t own_end = min((i + 1) * SUB_BITS, total_bits);
            uint32_t xp = p, xs = s, nblk = 0, mid3[3] = { p, s, 0u };
            if (!(xp != P_END && xp >= own_end)) walk_sync<WL, MID>(im, T, lds_addr(s_ctab), words, st, nseg, total_bits, own_end, xp, xs, nblk, i * SUB_BITS + SUB_BITS / 2, mid3);
            C.mep[CD_FILL * n + g] = p; C.mes[CD_FILL * n + g] = s; C.mxp[CD_FILL * n + g] = xp; C.mxs[CD_FILL * n + g] = xs; C.mnb[CD_FILL * n + g] = nblk;
            if (MID) { C.mmp[CD_FILL * n + g] = mid3[0]; C.mms[CD_FILL * n + g] = mid3[1]; C.mmn[CD_FILL * n + g] = mid3[2]; }
            // The maps around the new entry.  No two queued sub-sequences are neighbours (a queued one has no selection, its right neighbour
            // got the guess), so nobody else writes these two words in this round.
            uint32_t sc = CD_NONE;
            if (i + 1 < nsub && (i + 1) * SUB_BITS >= total_bits) sc = 0;          // the last data sub-sequence: what follows passes any state through
            else if (i + 1 < nsub) {
                sc = cd_match(C, g, im.blk_per_mcu, xp, xs);
                if (sc == CD_NONE && C.mep[CD_FILL * n + g + 1] == xp && C.mes[CD_FILL * n + g + 1] == xs) sc = CD_FILL;   // a walk of an earlier round next door
            }
            uint2 m = C.map[g]; m.y = (m.y & 0xFF00FFFFu) | (sc << 16); C.map[g] = m;
            // left neighbour: the entry the chain came through continues here; entries that pointed at what this slot held before do so no longer
            uint2 l = C.map[g - 1];
            #pragma unroll
            for (uint32_t e = 0; e < CD_SLOTS; e++) {
                const uint32_t sh = (e & 3u) * 8u; uint32_t& w = e < 4 ? l.x : l.y;
                if (e == prev) w = (w & ~(255u << sh)) | ((uint32_t)CD_FILL << sh);
                else if (((w >> sh) & 255u) == CD_FILL) w = (w & ~(255u << sh)) | ((uint32_t)CD_NONE << sh);
            }
            C.map[g - 1] = l;
        }
        __threadfence();                                         // the maps go to L2, this CU's L1 forgets them: the next chain reads what was patched
        __syncthreads();
    }
    if (t == 0) diag[0] = nreq;
}

// The selected memo entries, where k_sync / k_block_scan / k_write2 read them.  A sub-sequence the chain reached with a state in none of its
// slots gets an entry state no exit state equals (k_sync's verification mode walks it) and, as exit state, the guess its right neighbours were
// selected from -- if the guess was right, the walk changes nothing further.
template <int WL>
__global__ void __launch_bounds__(SY_THREADS) k_cand_apply(const JsImage* __restrict__ imgs, const uint32_t* __restrict__ sy_base, uint32_t nimg,
                                                           const JsTableSet* __restrict__ tables, const uint32_t* __restrict__ side, CandArrays C, SubArrays A, int mid)
{
    CD_PROLOGUE
    if (!valid) return;
    const uint32_t i_last = total_bits ? (total_bits - 1u) / SUB_BITS : 0u;
    if (i > i_last) {                                            // behind the data: the exit state of the last data sub-sequence passes through
        const size_t gl = g - (i - i_last); const uint32_t sl = C.sel[gl];
        const uint32_t ep = sl < CD_NONE ? C.mxp[sl * n + gl] : P_END, es = sl < CD_NONE ? C.mxs[sl * n + gl] : 0u;
        A.in_p[g] = sl < CD_NONE ? ep : 0xFFFFFFFEu; A.in_s[g] = sl < CD_NONE ? es : 0u; A.out_p[g] = ep; A.out_s[g] = es; A.nblk[g] = 0;
        if (mid) { C.hp[g] = ep; C.hs[g] = es; C.hn[g] = 0; }
        return;
    }
    const uint32_t s = C.sel[g];
    if (s < CD_NONE) {
        A.in_p[g] = C.mep[s * n + g]; A.in_s[g] = C.mes[s * n + g]; A.out_p[g] = C.mxp[s * n + g]; A.out_s[g] = C.mxs[s * n + g]; A.nblk[g] = C.mnb[s * n + g];
        if (mid) { C.hp[g] = C.mmp[s * n + g]; C.hs[g] = C.mms[s * n + g]; C.hn[g] = C.mmn[s * n + g]; }
    } else {
        if (mid) { C.hp[g] = 0xFFFFFFFEu; C.hs[g] = 0; C.hn[g] = 0; }   // (no lane of the write pass will agree with this: the image takes the resume path)
        const uint32_t gq = cd_byte(C.map[g], 7u);
        A.in_p[g] = 0xFFFFFFFEu; A.in_s[g] = 0; A.nblk[g] = 0;
        A.out_p[g] = gq < CD_NONE ? C.xp[gq * n + g] : P_END; A.out_s[g] = gq < CD_NONE ? C.xs[gq * n + g] : 0u;
    }
}

// One workgroup per image: exclusive scan of blocks-per-sub-sequence.  THREADS = 256 for batches (one workgroup per image, many
// images), 1024 for small jobs.  Every wave owns a contiguous segment and goes through it in tiles of 256 (a lane: four consecutive
// counts, one 16-byte load, the next tile's already in flight), carrying its running sum along -- no barrier inside a segment; the
// segment sums meet in LDS once, the second trip adds what lies before the segment.  (The first form took one barrier per THREADS
// sub-sequences: 31 of the 940 us of a single 3840x2160 image.)
template <int WL, int THREADS>
__global__ void __launch_bounds__(THREADS) k_block_scan(const JsImage* __restrict__ imgs, const JsTableSet* __restrict__ tables,
                                                        SubArrays A, uint32_t* side, uint32_t* __restrict__ flags)
{
    const uint32_t img = blockIdx.x; const JsImage& im = imgs[img];
    if (!tables[im.tableset].lut_ok) { if (threadIdx.x == 0) { FLAG_OR(flags, img, 0x0020u); ANOM_MIN(flags, img, ANOM_KEY(0u, AK_MIRROR)); } return; }
    const uint32_t total_bits = side[im.side_off + 10] * 8;
    const uint32_t n = min(im.n_subseq, (total_bits + SUB_BITS - 1) / SUB_BITS);
    constexpr uint32_t NW = THREADS / 64;
    __shared__ uint32_t s_w[NW];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t seg = (((n + NW - 1) / NW) + 255u) & ~255u;   // per wave, whole tiles: a tile that starts below n ends inside the image's (256-aligned) slot range
    const uint32_t* nb = A.nblk + im.subseq_off; uint32_t* base = A.base + im.subseq_off;
    const uint32_t w0 = wave * seg, w1 = min(w0 + seg, n);
    uint32_t run = 0;
    if (w0 < n) {
        uint4 nx = *reinterpret_cast<const uint4*>(nb + w0 + lane * 4);
        for (uint32_t b = w0; b < w1; b += 256) {
            const uint4 v = nx; const uint32_t i = b + lane * 4;
            if (b + 256 < w1) nx = *reinterpret_cast<const uint4*>(nb + i + 256);
            run += (i < n ? v.x : 0u) + (i + 1 < n ? v.y : 0u) + (i + 2 < n ? v.z : 0u) + (i + 3 < n ? v.w : 0u);
        }
    }
    #pragma unroll
    for (int off = 32; off >= 1; off >>= 1) run += __shfl_xor(run, off);
    if (lane == 0) s_w[wave] = run;
    __syncthreads();
    uint32_t carry = 0, tot = 0;
    for (uint32_t w = 0; w < NW; w++) { const uint32_t x = s_w[w]; if (w < wave) carry += x; tot += x; }
    if (w0 < n) {
        uint4 nx = *reinterpret_cast<const uint4*>(nb + w0 + lane * 4);
        for (uint32_t b = w0; b < w1; b += 256) {
            const uint4 v = nx; const uint32_t i = b + lane * 4;
            if (b + 256 < w1) nx = *reinterpret_cast<const uint4*>(nb + i + 256);
            const uint32_t a0 = i < n ? v.x : 0u, a1 = i + 1 < n ? v.y : 0u, a2 = i + 2 < n ? v.z : 0u, a3 = i + 3 < n ? v.w : 0u;
            const uint32_t mine = a0 + a1 + a2 + a3;
            uint32_t inc = mine;
            #pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const uint32_t a = __shfl_up(inc, off); if (lane >= (uint32_t)off) inc += a; }
            const uint32_t ex = carry + inc - mine;
            if (i < n) *reinterpret_cast<uint4*>(base + i) = make_uint4(ex, ex + a0, ex + a0 + a1, ex + a0 + a1 + a2);   // (slots past n inside the tile belong to this image: never read)
            carry += __shfl(inc, 63);
        }
    }
    if (threadIdx.x == 0) { side[im.side_off + 14] = tot; if (tot < im.total_blocks) { FLAG_OR(flags, img, F_SHORT); ANOM_MIN(flags, img, ANOM_KEY(tot, AK_MIRROR)); } }
}

// WRITE pass.  Every 8x8 block is written by exactly one lane -- the one that decodes its DC symbol.  A lane entering
// mid-block (k > 0) parses the rest of that block without output; a lane whose last block is unfinished at the end
// of its sub-sequence keeps decoding past it until the block completes.  Coefficients are dequantised
// (DecodeIdctSet :2270-2303) and de-zigzagged into a lane-private LDS block.  The symbol loop is wave-uniform
// (it runs while any lane is active), so that when some lanes complete a block in an iteration the WHOLE wave
// moves each finished block out: 64 lanes x 2 bytes = one coalesced 128-byte line per block.  HBM therefore sees
// whole lines (no scattered 2-byte read-modify-writes) and the coefficient arena needs no memset.
// SIDE variant (side-output pass over one already decoded image, grid = its workgroups starting at wg0): the same
// walk, but instead of coefficients it produces what the reference records while it decodes -- the bit position at
// which every MCU starts (-> m_pMcuFileMap :3229) and the Huffman code-length histogram (m_anDhtHisto :1190).  A
// symbol is counted by the lane in whose own range it starts; an MCU start is recorded by the lane that completes
// the last block of the MCU before it.
template <int WL, bool SIDE>
__global__ void __launch_bounds__(SY_THREADS) k_write(const JsImage* __restrict__ imgs, const uint32_t* __restrict__ sy_base, uint32_t nimg,
                                                      const JsTableSet* __restrict__ tables, const uint8_t* __restrict__ ustr,
                                                      const uint32_t* __restrict__ seg_tab, uint32_t* __restrict__ side, SubArrays A,
                                                      int16_t* __restrict__ coef, int16_t* __restrict__ dccum, uint8_t* __restrict__ mcu_rst, uint32_t* __restrict__ flags,
                                                      uint32_t tab_rows, uint32_t tab_lut2, uint32_t wg0, uint32_t* __restrict__ mcu_pos_in,
                                                      const uint8_t* __restrict__ img_mask = nullptr /* SIDE, a whole batch: the images to walk; positions at mcu_pos_in + rec_off of the image */)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    __shared__ __attribute__((aligned(16))) int16_t s_blk[SY_THREADS][WR_STRIDE];
    __shared__ uint32_t s_histo[SIDE ? 2 * 4 * 17 : 1];
    const uint32_t wg = blockIdx.x + wg0 + sy_base[0];
    const uint32_t img = find_image(sy_base, nimg, wg);
    const JsImage& im = imgs[img];
    if (!tables[im.tableset].lut_ok) return;
    if (SIDE && img_mask && !img_mask[img]) return;
    uint32_t* mcu_pos = (SIDE && img_mask) ? mcu_pos_in + im.rec_off : mcu_pos_in;
    const uint32_t* sd = side + im.side_off;
    const uint32_t total_bits = sd[10] * 8, nseg = min(sd[11], im.seg_cap - 1);
    const uint32_t lane = threadIdx.x & 63, wave0 = threadIdx.x & ~63u;
    const uint32_t sub0 = (wg - sy_base[img]) * SY_THREADS, i = sub0 + threadIdx.x;
    if (sub0 * SUB_BITS >= total_bits) return;
    if (SIDE) for (uint32_t q = threadIdx.x; q < 2 * 4 * 17; q += SY_THREADS) s_histo[q] = 0;
    const JsTableSet& tset = tables[im.tableset];
    SubTabs T; T.nb = im.blk_per_mcu; T.n1 = im.samp_h[1] * im.samp_v[1]; T.n2 = im.ncomp == 3 ? T.n1 + im.samp_h[2] * im.samp_v[2] : T.nb;
    WriteTabs W; load_wtabs(W, s_dyn, tset, tab_rows, tab_lut2, im.ncomp, threadIdx.x, SY_THREADS);
    { uint32_t* z = reinterpret_cast<uint32_t*>(s_blk[threadIdx.x]); for (int j = 0; j < WR_STRIDE / 2; j++) z[j] = 0u; }
    __syncthreads();

    const uint32_t* words = reinterpret_cast<const uint32_t*>(ustr + im.ustr_off);
    const uint32_t* st = seg_tab + im.seg_off;
    int16_t* cbase = coef + im.coef_off * 64; int16_t* dbase = dccum + im.coef_off; uint8_t* rstf = mcu_rst + im.mcu_off;
    int16_t* lbuf = s_blk[threadIdx.x];
    const uint32_t nblocks = im.total_blocks, decode_ac = im.decode_ac, prec_shift = im.precision >= 8 ? ((im.precision - 8) & 31) : 0;
    const bool in_data = i * SUB_BITS < total_bits;
    const size_t g = im.subseq_off + i;
    uint32_t fl = 0, an = 0xFFFFFFFFu, nblk = 0, seg = 0, c = 0, k = 0, seg_end = 0, blk = 0;
    uint32_t res_p = 0, res_s = 0, res_n = 0;                // what this lane reports for verification
    bool verify = false, check_n = false, active = false, captured = false, skip = false;
    const uint32_t own_end = min((i + 1) * SUB_BITS, total_bits);
    Cursor cur; cur.words = words; cur.widx = 0; cur.poff = 0; cur.w0 = cur.w1 = cur.nxt = 0; cur.sh = 0; cur.p = 0;
    if (in_data) {
        const uint32_t p0 = i ? A.out_p[g - 1] : 0u, s0 = i ? A.out_s[g - 1] : 0u;
        blk = A.base[g]; seg = ST_SEG(s0); c = ST_C(s0); k = ST_K(s0);
        verify = true;
        if (p0 != P_END && p0 >= own_end) { res_p = p0; res_s = s0; }                          // owns no symbol: passes through
        else if (p0 == P_END || (p0 >= total_bits && seg + 1 >= nseg)) { res_p = P_END; res_s = 0; check_n = true; }
        else if (blk >= nblocks) verify = false;                                                // everything owned lies past the last MCU
        else { active = true; check_n = true; seg_end = st[seg + 1] * 8; skip = k != 0; cur_init<WL>(cur, words, p0); }
    }
    int16_t dq0 = 0;
    const uint32_t wb0 = W.wb0, wb1 = W.wb1, wb2 = W.wb2;       // per component: byte offset of its DC row | of its AC row << 16
    const char* l1b = W.rows;
    uint32_t comp = comp_of(T, c), wb = comp == 0 ? wb0 : (comp == 1 ? wb1 : wb2);
    uint64_t amask = ~0ull;
    for (;;) {
        // ---- end of the owned range: report the state there; keep going only to finish a block this lane started
        // Votes: the lane mask of `active` is formed once per step (amask) and ANDed with votes on plain register compares -- a vote on
        // "active && ..." makes the compiler materialise the loop-carried flag as 0 / 1 and compare it again, two half-rate VALU
        // instructions per vote.  amask may still hold lanes that went inactive late in the previous step: such a vote only enters a
        // block in which no lane acts.
        const bool at_end = active && cur.p >= own_end;
        if (WBALLOT(cur.p >= own_end) & amask) {
            if (at_end) {
                if (!captured) { res_p = cur.p; res_s = ST_MAKE(seg, c, k); res_n = nblk; captured = true; }
                if (k == 0 || skip) active = false;
            }
        }
        amask = WBALLOT(active);
        if (!amask) break;
        // ---- one or two symbols per lane: straight-line select code on the common path
        const uint32_t win = cur_peek(cur);
        const bool isdc = k == 0;
        const uint32_t widx = win >> (32 - JS_L1_BITS);
        // Both tables are read unconditionally and side by side (some lane of the wave is at a DC symbol in nearly every step: a DC
        // read behind a branch is a second, dependent LDS round trip per step); the DC entry has the AC entry's format in 16 bits
        // (code length | size << 4, bit 15 = escape to the second level, 0 = no code), widened without a branch.
        const uint32_t e_dc = *reinterpret_cast<const uint16_t*>(l1b + ((wb & 0xFFFFu) + (widx << 1)));
        const uint32_t e_ac = *reinterpret_cast<const uint32_t*>(l1b + ((wb >> 16) + (widx << 2)));   // AC: this symbol and, where visible, the one behind it
        const uint32_t e_dcw = (e_dc & 0x7FFFu) | ((e_dc & 0x8000u) << 16);
        uint32_t e = isdc ? e_dcw : e_ac;
        uint32_t len = e & 15u, size = (e >> 4) & 15u, run = (e >> 8) & 15u;
        if (WBALLOT((int32_t)e < 0) & amask) {                // some lane holds a code longer than JS_L1_BITS bits (a few % of symbols), or no code
            if ((int32_t)e < 0) {
                if (e & 0x40000000u) { len = 0; size = 0; run = 0; }
                else { const uint32_t nbx = (e >> 12) & 7u; const uint32_t e2 = W.lut2[(e & 0xFFFu) + ((win >> (32 - JS_L1_BITS - nbx)) & ((1u << nbx) - 1u))];
                       len = (e2 >> 8) & 31u; run = (e2 >> 4) & 15u; size = e2 & 15u; }
                e = 0;                                           // a single symbol
            }
        }
        const uint32_t tot = len + size;
        const uint32_t k2 = isdc ? 1u : k + run + 1u;            // coefficient index behind this symbol
        // the AC symbol behind it goes along when the first one neither ends the block nor the lane's own range, and nothing
        // out of the ordinary can happen on the way (interval end, coefficient overflow)
        const uint32_t len2 = (e >> 12) & 15u, size2 = (e >> 16) & 15u, run2 = (e >> 20) & 15u, tot2 = len2 + size2, k3 = k2 + run2 + 1u;
        bool two = !SIDE && active && ((e >> 24) & 1u) && (run | size) != 0u && k2 < 64u && k3 <= 64u && cur.p + tot < own_end && cur.p + tot + tot2 <= seg_end;
        // ---- anything out of the ordinary sits behind one vote: no code, the end of the interval inside the code or its value
        //      bits, a run past the 64th coefficient
        bool norm = active, bad = false;                         // bad: a code that matches nothing ended the block (walk_slow)
        if (WBALLOT(len == 0 || cur.p + tot > seg_end || k2 > 64u) & amask) {
            if (active && (len == 0 || cur.p + len > seg_end)) {  // interval / stream end, or a code that matches nothing
                // (the side pass repeats a walk whose restart marks are set -- and, for an image whose decode ends early, pruned behind that end, k_dead_fill: it sets none)
                const int ws = walk_slow<true, WL>(im, words, st, nseg, cur, len, seg, seg_end, c, k, blk, !SIDE && !captured, rstf, fl, an);
                if (ws == WS_OVER) { if (!captured) { captured = true; res_p = P_END; res_s = 0; res_n = nblk; } active = false; }
                bad = ws == WS_BAD_CODE;
                comp = comp_of(T, c); wb = comp == 0 ? wb0 : (comp == 1 ? wb1 : wb2);
                norm = false;
            } else if (active && blk < nblocks) {
                if (cur.p + tot > seg_end) { fl |= F_OVERRUN; an = min(an, ANOM_KEY(blk, AK_MIRROR)); }
                if (k2 > 64u) {
                    fl |= F_COEF_OVERFLOW;
                    // side pass: what the reference's two messages about this block quote (:1723-1735 "nNumCoeffs>64", CheckScanErrors :2605) --
                    // the block, where the offending symbol starts, the index it ran to, where the block ends.  `flags` is the record list here.
                    if (SIDE && flags && !captured) {
                        const uint32_t slot = atomicAdd(&flags[0], 1u);
                        if (slot < JS_ANOM_MAX) { uint32_t* e = flags + 4 + 4 * slot; e[0] = blk; e[1] = cur.p; e[2] = k2; e[3] = cur.p + tot; }
                    }
                }
            }
        }
        // value bits: EXTEND (HuffmanDc2Signed :859), precision divide (:1234-1238), dequantise (:2278)
        const uint32_t vraw = __builtin_amdgcn_ubfe(win, 32u - tot, size);           // size == 0: 0
        const uint32_t lim = (1u << size) - 1u;
        int32_t val = (int32_t)(win << len) < 0 ? (int32_t)vraw : (int32_t)(vraw - lim);   // first value bit clear: negative (size == 0: 0 - 0)
        if (prec_shift) val /= (int32_t)(1u << prec_shift);
        const uint32_t ind = k2 - 1u;                            // DC: 0, AC: k + run
        const uint32_t qz = W.qz[comp * 64 + (ind & 63u)];
        const int16_t dq = (int16_t)((int32_t)(int16_t)val * (int32_t)(qz & 0xFFFFu));
        if (!SIDE && norm && !skip && (isdc || (decode_ac && size)) && ind < 64) lbuf[qz >> 16] = dq;
        if (SIDE && norm && !captured && blk < nblocks) atomicAdd(&s_histo[((isdc ? 0u : 4u) + tset.dest_id[comp * 2 + (isdc ? 0u : 1u)]) * 17u + len], 1u);
        dq0 = (norm && isdc) ? dq : dq0;
        if (WBALLOT(bad) & amask) {
            if (bad) {                                           // the block keeps its DC difference (none decoded yet: 0) and nothing else
                if (isdc) dq0 = 0;
                if (!SIDE && !skip) { uint32_t* z = reinterpret_cast<uint32_t*>(lbuf); for (int j = 0; j < WR_STRIDE / 2; j++) z[j] = 0u; lbuf[0] = dq0; }
                if (SIDE && !captured && blk < nblocks) atomicAdd(&s_histo[((isdc ? 0u : 4u) + tset.dest_id[comp * 2 + (isdc ? 0u : 1u)]) * 17u + 1u], 1u);   // (one bit "used": m_anDhtHisto[..][1])
            }
        }
        two = two && norm;
        if (!SIDE) {                                             // second symbol: bits [tot, tot + tot2) of the same window (tot + tot2 <= 24)
            const uint32_t vraw2 = __builtin_amdgcn_ubfe(win, 32u - tot - tot2, size2);
            int32_t val2 = (int32_t)(win << (tot + len2)) < 0 ? (int32_t)vraw2 : (int32_t)(vraw2 - ((1u << size2) - 1u));
            if (prec_shift) val2 /= (int32_t)(1u << prec_shift);
            const uint32_t qz2 = W.qz[comp * 64 + ((k3 - 1u) & 63u)];
            const int16_t dq2 = (int16_t)((int32_t)(int16_t)val2 * (int32_t)(qz2 & 0xFFFFu));
            if (two && !skip && decode_ac && size2) lbuf[qz2 >> 16] = dq2;             // k3 - 1 < 64 by construction
        }
        cur_skip<WL>(cur, norm ? (two ? tot + tot2 : tot) : 0u);
        const uint32_t kn = two ? k3 : k2;
        const bool done = bad || (norm && !isdc && (two ? ((run2 | size2) == 0u || k3 >= 64u) : ((run | size) == 0u || k2 >= 64u)));
        k = norm ? (done ? 0u : kn) : (bad ? 0u : k);
        const bool flush = done && !skip && blk < nblocks;
        const uint32_t fblk = blk;
        if (done) {
            c = c + 1 == T.nb ? 0u : c + 1; comp = comp_of(T, c); wb = comp == 0 ? wb0 : (comp == 1 ? wb1 : wb2);
            nblk += captured ? 0u : 1u;
            if (!SIDE && flush) dbase[blk] = dq0;
            if (SIDE && flush && c == 0) mcu_pos[(blk + 1) / T.nb] = cur.p;      // the next MCU starts here (before any restart handling)
            if (SIDE && flush && blk + 2u == nblocks) mcu_pos[nblocks / T.nb + 1u] = cur.p;   // ... and the image's LAST block here: where k_side_maps starts its reader for the end of the scan
            skip = false; blk++;
        }
        // ---- the whole wave moves every block that completed in this iteration: one 128-byte line each ----
        // Up to FOUR finished blocks per trip: 16 lanes move one block (8 bytes each), so a trip is one LDS read and one
        // 512-byte store instruction -- few, wide stores keep the count of outstanding memory operations (which the next
        // bit-window refill has to wait for) low.
        // The flushing lanes are ranked (mbcnt) and one wave permute hands lane j the id of the j-th flushing lane; per trip a group of
        // 16 lanes fetches "its" source lane and that lane's block number with two more permutes -- no scalar loop over the vote mask
        // (find-first-set, readlane and per-group selects cost 25 scalar + 21 vector instructions per trip).
        const uint64_t fm = SIDE ? 0ull : WBALLOT(flush);
        if (fm) {
            const uint32_t nfl = (uint32_t)__builtin_popcountll(fm);
            const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
            const uint32_t dst = flush ? rk : nfl + lane - rk;                       // a full permutation: flushing lanes first, in lane order
            const uint32_t ent = (uint32_t)__builtin_amdgcn_ds_permute((int)(dst << 2), (int)lane);
            for (uint32_t t0 = 0; t0 < nfl; t0 += 4u) {
                const uint32_t idx = t0 + (lane >> 4);
                const uint32_t src = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(idx << 2), (int)ent);
                const uint32_t b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)fblk);
                if (idx < nfl) {
                    uint32_t* sb = reinterpret_cast<uint32_t*>(s_blk[wave0 + src]) + (lane & 15u) * 2u;
                    const uint32_t lo = sb[0], hi = sb[1];
                    sb[0] = 0u; sb[1] = 0u;
                    *reinterpret_cast<uint2*>(cbase + (size_t)b * 64 + (lane & 15u) * 4u) = make_uint2(lo, hi);
                }
            }
        }
    }
    if (verify) {
        if (check_n && !captured && res_p != P_END) { res_p = cur.p; res_s = cur.p == P_END ? 0u : ST_MAKE(seg, c, k); res_n = nblk; }
        // the chain must be at its fixed point, and the block count that fed the prefix sum must be the real one
        if (res_p != A.out_p[g] || res_s != A.out_s[g] || (check_n && res_n != A.nblk[g])) fl |= F_NOSYNC;
    }
    if (SIDE) {
        __syncthreads();
        uint32_t* ho = side + im.side_off + JS_SIDE_HISTO;
        for (uint32_t q = threadIdx.x; q < 2 * 4 * 17; q += SY_THREADS) { const uint32_t v = s_histo[q]; if (v) atomicAdd(&ho[q], v); }
    } else if (fl) { FLAG_OR(flags, img, fl); if (an != 0xFFFFFFFFu) ANOM_MIN(flags, img, an); }
}

// WRITE pass, second form (the one the main path launches; k_write<., true> above stays the side-output pass and k_write<., false> the
// cross-check, JSNOOP_WRITE_V1=1).  Same walk, same results -- written against the instruction-cost table of DESIGN.md §4.7:
//  * every per-lane flag that lives across steps (active, skip, captured) is a 64-bit lane MASK in scalar registers; votes are compares
//    written straight into a scalar pair and combined there, a lane reads its bit back as a predicate (inverse ballot) -- the compiler
//    never has to turn a flag into 0 / 1 in a vector register and compare it again;
//  * the rare events (end of the own range, codes longer than the first level, interval ends and errors) each sit behind ONE vote, and
//    they leave the lane in a state in which the straight-line code of the step does nothing for it (no bits consumed, index kept);
//  * zero-valued coefficients (ZRL, EOB) are simply stored: they land on positions that are zero and are never written twice;
//  * the flush has no scalar loop (ranks + wave permutes), its block rows leave as non-temporal stores through a scalar base and a
//    32-bit offset, and the DC differences of finished blocks wait in registers and leave eight at a time: what the vector memory
//    pipe is charged for is the number of store instructions, not their bytes.
#define IBAL(m) __builtin_amdgcn_inverse_ballot_w64(m)
// EXTEND (HuffmanDc2Signed :859) of the `size` bits that follow `skipbits` bits of the window; size == 0 -> 0
__device__ __forceinline__ int32_t extend_bits(uint32_t win, uint32_t skipbits, uint32_t size)
{
    const uint32_t x = win << skipbits;
    const uint32_t vraw = __builtin_amdgcn_ubfe(x, 32u - size, size);
    const uint32_t neglim = (0xFFFFFFFFu << size) + 1u;          // -(2^size - 1)
    return (int32_t)x < 0 ? (int32_t)vraw : (int32_t)(vraw + neglim);
}
// HALF (small jobs after the candidate synchronisation): two lanes per sub-sequence -- the second one enters at the state the selected memo
// walk reported for the middle of the sub-sequence (half_*: position, state word, blocks completed before it) -- twice the lanes, half the steps of
// the chain a wave is; the lane of the first half verifies against that middle state, the other one against the exit state as before.
// REC (a single-image call that will ask for the side outputs anyway, js_side_prepare): the pass also records what the side walk (k_write<., true>) would -- the bit
// position of every MCU top and of the image's last block top in rec_pos, the code-length histogram (m_anDhtHisto) into the side block -- and that walk is not run.
template <int WL, bool HALF = false, bool REC = false>
__global__ void __launch_bounds__(SY_THREADS) k_write2(const JsImage* __restrict__ imgs, const uint32_t* __restrict__ sy_base, uint32_t nimg,
                                                       const JsTableSet* __restrict__ tables, const uint8_t* __restrict__ ustr,
                                                       const uint32_t* __restrict__ seg_tab, uint32_t* __restrict__ side, SubArrays A,
                                                       int16_t* __restrict__ coef, int16_t* __restrict__ dccum, uint8_t* __restrict__ mcu_rst, uint32_t* __restrict__ flags,
                                                       uint32_t tab_rows, uint32_t tab_lut2, const uint32_t* __restrict__ half_p = nullptr, const uint32_t* __restrict__ half_s = nullptr,
                                                       const uint32_t* __restrict__ half_n = nullptr, uint32_t* __restrict__ rec_pos = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    __shared__ __attribute__((aligned(16))) int16_t s_blk[SY_THREADS][WR_STRIDE];
    __shared__ uint32_t s_histo[REC ? 2 * 4 * 17 : 1];
    const uint32_t wg = (HALF ? blockIdx.x >> 1 : blockIdx.x) + sy_base[0];
    const uint32_t img = find_image(sy_base, nimg, wg);
    const JsImage& im = imgs[img];
    if (!tables[im.tableset].lut_ok) return;
    const uint32_t* sd = side + im.side_off;
    const uint32_t total_bits = sd[10] * 8, nseg = min(sd[11], im.seg_cap - 1);
    const uint32_t lane = threadIdx.x & 63, wave0 = threadIdx.x & ~63u;
    const uint32_t hi = HALF ? threadIdx.x & 1u : 0u;            // HALF: which half of the sub-sequence this lane has
    const uint32_t sub0 = (wg - sy_base[img]) * SY_THREADS + (HALF ? (blockIdx.x & 1u) * (SY_THREADS / 2) : 0u), i = sub0 + (HALF ? threadIdx.x >> 1 : threadIdx.x);
    if (sub0 * SUB_BITS >= total_bits) return;
    // block rows are addressed with 32-bit byte offsets from the image's first row: an image of 2^25 blocks or more (2 Gpixel of
    // grayscale) is left to the exact kernel
    if (im.total_blocks >= (1u << 25)) { if (threadIdx.x == 0) { FLAG_OR(flags, img, F_SHORT); ANOM_MIN(flags, img, ANOM_KEY(0u, AK_MIRROR)); } return; }
    const JsTableSet& tset = tables[im.tableset];
    SubTabs T; T.nb = im.blk_per_mcu; T.n1 = im.samp_h[1] * im.samp_v[1]; T.n2 = im.ncomp == 3 ? T.n1 + im.samp_h[2] * im.samp_v[2] : T.nb;
    WriteTabs W; load_wtabs(W, s_dyn, tset, tab_rows, tab_lut2, im.ncomp, threadIdx.x, SY_THREADS);
    { uint32_t* z = reinterpret_cast<uint32_t*>(s_blk[threadIdx.x]); for (int j = 0; j < WR_STRIDE / 2; j++) z[j] = 0u; }
    if (REC) for (uint32_t q = threadIdx.x; q < 2 * 4 * 17; q += SY_THREADS) s_histo[q] = 0;
    __syncthreads();

    const uint32_t* words = reinterpret_cast<const uint32_t*>(ustr + im.ustr_off);
    const uint32_t* st = seg_tab + im.seg_off;
    int16_t* cbase = coef + im.coef_off * 64; int16_t* dbase = dccum + im.coef_off; uint8_t* rstf = mcu_rst + im.mcu_off;
    char* lbuf = reinterpret_cast<char*>(s_blk[threadIdx.x]);
    const uint32_t nblocks = im.total_blocks, prec_shift = im.precision >= 8 ? ((im.precision - 8) & 31) : 0;
    const uint64_t acmask = im.decode_ac ? ~0ull : 0ull;        // DC-only mode: AC coefficients are parsed, not stored
    const bool in_data = i * SUB_BITS + hi * (SUB_BITS / 2) < total_bits;
    const size_t g = im.subseq_off + i;
    uint32_t fl = 0, an = 0xFFFFFFFFu, nblk = 0, seg = 0, c = 0, k = 0, seg_end = 0, blk = 0;
    uint32_t res_p = 0, res_s = 0, res_n = 0;                // what this lane reports for verification
    bool verify = false, check_n = false, active0 = false, skip0 = false;
    const uint32_t own_end = min(HALF && !hi ? i * SUB_BITS + SUB_BITS / 2 : (i + 1) * SUB_BITS, total_bits);
    Cursor cur; cur.words = words; cur.widx = 0; cur.poff = 0; cur.w0 = cur.w1 = cur.nxt = 0; cur.sh = 0; cur.p = 0;
    if (in_data) {
        const uint32_t p0 = hi ? half_p[g] : (i ? A.out_p[g - 1] : 0u), s0 = hi ? half_s[g] : (i ? A.out_s[g - 1] : 0u);
        blk = A.base[g] + (hi ? half_n[g] : 0u); seg = ST_SEG(s0); c = ST_C(s0); k = ST_K(s0);
        verify = true;
        if (p0 != P_END && p0 >= own_end) { res_p = p0; res_s = s0; }                          // owns no symbol: passes through
        else if (p0 == P_END || (p0 >= total_bits && seg + 1 >= nseg)) { res_p = P_END; res_s = 0; check_n = true; }
        else if (blk >= nblocks) verify = false;                                                // everything owned lies past the last MCU
        else { active0 = true; check_n = true; seg_end = st[seg + 1] * 8; skip0 = k != 0; cur_init<WL>(cur, words, p0); }
    }
    const uint32_t wb0 = W.wb0, wb1 = W.wb1, wb2 = W.wb2;       // per component: byte offset of its DC row | of its AC row << 16
    const char* l1b = W.rows;
    const char* qzb = reinterpret_cast<const char*>(W.qz);
    uint32_t comp = comp_of(T, c), wb = comp == 0 ? wb0 : (comp == 1 ? wb1 : wb2);
    uint64_t m_act = WBALLOT(active0), m_skip = WBALLOT(skip0), m_cap = 0ull;
    // DC differences of finished blocks wait in registers -- eight per lane, in memory order: the oldest in the low half of dcq0 (a lane's
    // finished blocks have consecutive numbers) -- and leave as ONE 16-byte store of the lanes that hold eight: what the vector memory pipe
    // is charged for is the store instruction, not its bytes (a 2-byte store per block: +0.45 ms per 1024 images; eight 2-byte stores per
    // lane and eight blocks: +0.4).  The address is only 2-byte aligned: global memory takes unaligned 16-byte accesses (checked once per device:
    // k_unaligned_probe).  Consecutive by construction: a lane's block number goes up by one with every block it ends, it flushes every block it ends
    // except a first one it entered in the middle (skip) and blocks past the image's last -- after which it flushes none.
    uint32_t dcq0 = 0, dcq1 = 0, dcq2 = 0, dcq3 = 0, dccnt = 0, dclast = 0, dq0 = 0;
    uint32_t rst_blk = 0xFFFFFFFFu;                              // the block in progress when a restart was last followed (see ANOM_KEY): written on the slow path only
    auto dc_store_rest = [&]() {                                 // what is left at the end: the newest dccnt values sit in the top halves
        #pragma unroll
        for (uint32_t h = 0; h < 8; h++) {
            const uint32_t q = h < 2 ? dcq0 : (h < 4 ? dcq1 : (h < 6 ? dcq2 : dcq3));
            if (h + dccnt >= 8u) dbase[dclast - 7u + h] = (int16_t)((h & 1u) ? q >> 16 : q);
        }
        dccnt = 0;
    };
    for (;;) {
        // ---- end of the owned range: report the state there; keep going only to finish a block this lane started
        const uint64_t m_end = WBALLOT(cur.p >= own_end) & m_act;
        if (m_end) {
            if (IBAL(m_end & ~m_cap)) { res_p = cur.p; res_s = ST_MAKE(seg, c, k); res_n = nblk; }
            m_cap |= m_end;
            m_act &= ~(m_end & (WBALLOT(k == 0) | m_skip));
        }
        if (!m_act) break;
        // ---- table entry: both tables read side by side (some lane of the wave is at a DC symbol in nearly every step)
        const uint32_t win = cur_peek(cur);
        const uint32_t widx = win >> (32 - JS_L1_BITS);
        const uint64_t m_dc = WBALLOT(k == 0);
        const uint32_t e_dc = (uint32_t)(int32_t)*reinterpret_cast<const int16_t*>(l1b + ((wb & 0xFFFFu) + (widx << 1)));   // escape (bit 15) becomes the sign
        const uint32_t e_ac = *reinterpret_cast<const uint32_t*>(l1b + ((wb >> 16) + (widx << 2)));
        uint32_t e = IBAL(m_dc) ? e_dc : e_ac;
        uint32_t len = e & 15u, size = (e >> 4) & 15u, run = (e >> 8) & 15u;
        const uint64_t m_esc = WBALLOT((int32_t)e < 0) & m_act;
        if (m_esc) {                                             // a code longer than JS_L1_BITS bits (a few % of symbols), or no code
            if (IBAL(m_esc)) {
                const uint32_t nbx = (e >> 12) & 7u;
                const uint32_t e2 = W.lut2[(e & 0xFFFu) + __builtin_amdgcn_ubfe(win, 32u - JS_L1_BITS - nbx, nbx)];
                const bool nocode = e == 0xC0000000u;            // (AC rows; a DC row says 0, which is "length 0" already)
                len = nocode ? 0u : (e2 >> 8) & 31u; run = nocode ? 0u : (e2 >> 4) & 15u; size = nocode ? 0u : e2 & 15u;
                e = 0;                                           // a single symbol
            }
        }
        uint32_t tot = len + size;
        uint32_t k2 = k + run + 1u;                              // coefficient index behind this symbol (DC: k = 0, run = 0)
        // the AC symbol behind it goes along when the first one does not end the lane's own range, and nothing out of the ordinary can
        // happen on the way (interval end, coefficient overflow); an entry without a visible second symbol has zero bits there
        const uint32_t len2 = (e >> 12) & 15u, size2 = (e >> 16) & 15u, run2 = (e >> 20) & 15u, tot2 = len2 + size2, k3 = k2 + run2 + 1u;
        const uint32_t p1 = cur.p + tot, p2 = p1 + tot2;
        uint64_t m_two = WBALLOT(tot2 != 0u) & WBALLOT(k3 <= 64u) & WBALLOT(p1 < own_end) & WBALLOT(p2 <= seg_end) & m_act;
        // ---- anything out of the ordinary sits behind one vote: no code, the end of the interval inside the code or its value
        //      bits, a run past the 64th coefficient
        uint64_t m_norm = m_act, m_nost = 0ull, m_bad = 0ull;      // m_bad: a code that matches nothing ended the lane's block (walk_slow)
        const uint64_t m_abn = (WBALLOT(len == 0u) | WBALLOT(p1 > seg_end) | WBALLOT(k2 > 64u)) & m_act;
        if (m_abn) {
            const uint64_t m_slow = (WBALLOT(len == 0u) | WBALLOT(cur.p + len > seg_end)) & m_act;
            bool over = false, bad = false;                      // (lane masks change in wave-uniform code only)
            if (IBAL(m_slow)) {                                  // interval / stream end, or a code that matches nothing
                const bool notcap = !IBAL(m_cap);
                const uint32_t seg_was = seg;
                const int ws = walk_slow<true, WL>(im, words, st, nseg, cur, len, seg, seg_end, c, k, blk, notcap, rstf, fl, an);
                rst_blk = seg != seg_was ? blk : rst_blk;
                if (ws == WS_OVER && notcap) { res_p = P_END; res_s = 0; res_n = nblk; }
                over = ws == WS_OVER; bad = ws == WS_BAD_CODE;
                comp = comp_of(T, c); wb = comp == 0 ? wb0 : (comp == 1 ? wb1 : wb2);
                tot = 0; size = 0; k2 = k;                       // the step below does nothing for this lane
            } else if (IBAL(m_act)) {
                if (blk < nblocks) {
                    // value bits past the end of the interval.  With an RSTn behind it the reference's register over-reads -- its decode of the image ENDS
                    // in this block (ANOM_KEY); at the end of the scan data it reads on through the marker bytes (F_SHORT / second attempt).  The walk
                    // itself goes on as the synchronisation walks did: what it writes from here on is replaced (js_parallel_fixup).
                    if (p1 > seg_end) { fl |= F_OVERRUN; an = min(an, ANOM_KEY(blk, seg + 1u < nseg ? AK_DEAD + (IBAL(m_dc) ? 0u : 1u) + (rst_blk == blk ? 2u : 0u) + (IBAL(m_skip) ? 4u : 0u) : AK_MIRROR)); }
                    if (k2 > 64u) fl |= F_COEF_OVERFLOW;
                }
            }
            const uint64_t m_over = WBALLOT(over);
            m_bad = WBALLOT(bad) & m_act;
            m_cap |= m_over; m_act &= ~m_over;
            m_norm &= ~m_slow; m_two &= ~m_slow;
            m_nost = WBALLOT(k2 > 64u);
        }
        if (REC) {                                               // code lengths of what this lane decodes inside its own range (k_write<., true> counts the same symbols)
            const uint32_t hd = tset.dest_id[comp * 2], ha = tset.dest_id[comp * 2 + 1];
            if (IBAL(m_norm & ~m_cap) && blk < nblocks) {
                atomicAdd(&s_histo[(IBAL(m_dc) ? hd : 4u + ha) * 17u + len], 1u);
                if (IBAL(m_two)) atomicAdd(&s_histo[(4u + ha) * 17u + len2], 1u);
            }
            if (IBAL(m_bad & ~m_cap) && blk < nblocks) atomicAdd(&s_histo[(IBAL(m_dc) ? hd : 4u + ha) * 17u + 1u], 1u);     // (one bit "used", :1178-1186)
        }
        // ---- value bits: EXTEND (HuffmanDc2Signed :859), precision divide (:1234-1238), dequantise (:2278), de-zigzag
        int32_t val = extend_bits(win, len, size), val2 = extend_bits(win, tot + len2, size2);
        if (prec_shift) { val /= (int32_t)(1u << prec_shift); val2 /= (int32_t)(1u << prec_shift); }
        const char* qrow = qzb + comp * 256u;
        const uint32_t qz = *reinterpret_cast<const uint32_t*>(qrow + (((k2 - 1u) & 63u) << 2));           // DC: 0, AC: k + run
        const uint32_t qz2 = *reinterpret_cast<const uint32_t*>(qrow + (((k3 - 1u) & 63u) << 2));
        const uint64_t m_st = m_norm & ~m_skip & ~m_nost & (m_dc | acmask);
        const uint32_t dq = (uint32_t)((int32_t)(int16_t)val * (int32_t)(qz & 0xFFFFu));
        if (IBAL(m_st)) *reinterpret_cast<int16_t*>(lbuf + ((qz >> 16) << 1)) = (int16_t)dq;
        dq0 = IBAL(m_dc & m_norm) ? dq : dq0;                    // the block's DC difference
        if (m_bad) {
            if (IBAL(m_bad)) {                                   // the block keeps its DC difference (none decoded yet: 0) and nothing else
                if (IBAL(m_dc)) dq0 = 0u;
                if (!IBAL(m_skip)) { uint32_t* z = reinterpret_cast<uint32_t*>(lbuf); for (int j = 0; j < WR_STRIDE / 2; j++) z[j] = 0u; *reinterpret_cast<int16_t*>(lbuf) = (int16_t)dq0; }
            }
        }
        if (IBAL(m_two & ~m_skip & acmask)) *reinterpret_cast<int16_t*>(lbuf + ((qz2 >> 16) << 1)) = (int16_t)((int32_t)(int16_t)val2 * (int32_t)(qz2 & 0xFFFFu));
        // ---- advance
        const bool two = IBAL(m_two);
        { const uint32_t adv = tot + (two ? tot2 : 0u); cur.sh -= (int32_t)adv; cur.p += adv; }
        if (IBAL(WBALLOT(cur.sh < 0) & m_act)) {                 // (lanes that are not active compute on whatever they hold: they must not load)
            cur.sh += 32; cur.w0 = cur.w1; cur.w1 = bswap32(cur.nxt);
            cur.nxt = cur_fetch<WL>(cur);
        }
        const uint32_t kn = two ? k3 : k2;
        const uint64_t m_eob = (m_two & WBALLOT((run2 | size2) == 0u)) | (~m_two & WBALLOT((run | size) == 0u));
        const uint64_t m_done = (m_norm & ~m_dc & (m_eob | WBALLOT(kn >= 64u))) | m_bad;
        k = IBAL(m_done) ? 0u : kn;
        if (m_done) {
            const uint64_t m_flush = m_done & ~m_skip & WBALLOT(blk < nblocks);
            const uint32_t fblk = blk;
            if (IBAL(m_done)) {
                c = c + 1 == T.nb ? 0u : c + 1; comp = comp_of(T, c); wb = comp == 0 ? wb0 : (comp == 1 ? wb1 : wb2);
                nblk += IBAL(m_cap) ? 0u : 1u;
                if (REC && IBAL(m_flush)) {                       // the next MCU / the image's last block starts here (before any restart handling)
                    if (c == 0) rec_pos[(blk + 1) / T.nb] = cur.p;
                    if (blk + 2u == nblocks) rec_pos[nblocks / T.nb + 1u] = cur.p;
                }
                blk++;
            }
            m_skip &= ~m_done;
            // ---- the whole wave moves every block that completed in this step: 8 lanes x 16 bytes per block, eight blocks per store
            // instruction (round 6; four per instruction before: 3.35 -> 3.23 ms, a step completes 4.5 blocks on average -- one trip instead of two).  The flushing lanes are ranked (mbcnt); two wave permutes hand lane j the id and the block number of the j-th
            // flushing lane, and per trip a group of 16 lanes fetches "its" pair with two more permutes -- no scalar loop over the vote.
            if (m_flush) {
                const bool flush = IBAL(m_flush);
                const uint32_t nfl = (uint32_t)__builtin_popcountll(m_flush);
                const uint32_t rk = __builtin_amdgcn_mbcnt_hi((uint32_t)(m_flush >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m_flush, 0u));
                const uint32_t dst = (flush ? rk : nfl + lane - rk) << 2;                // a full permutation: flushing lanes first, in lane order
                const uint32_t ent_l = (uint32_t)__builtin_amdgcn_ds_permute((int)dst, (int)lane);
                const uint32_t ent_b = (uint32_t)__builtin_amdgcn_ds_permute((int)dst, (int)fblk);
                for (uint32_t t0 = 0; t0 < nfl; t0 += 8u) {
                    const uint32_t idx = t0 + (lane >> 3);
                    const uint32_t src = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(idx << 2), (int)ent_l);
                    const uint32_t b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(idx << 2), (int)ent_b);
                    if (idx < nfl) {
                        uint4* sb = reinterpret_cast<uint4*>(s_blk[wave0 + src]) + (lane & 7u);
                        const uint4 v = *sb;
                        *sb = make_uint4(0u, 0u, 0u, 0u);
                        // (written once, read once by a later kernel, 6.4 GB per 1024 images: kept out of the caches' way -- 3.42 -> 3.31 ms)
                        { typedef uint32_t u32x4_nt __attribute__((ext_vector_type(4))); u32x4_nt t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
                          __builtin_nontemporal_store(t, reinterpret_cast<u32x4_nt*>(reinterpret_cast<char*>(cbase) + ((b << 7) | ((lane & 7u) << 4)))); }   // scalar base + 32-bit vector offset
                    }
                }
                if (flush) {
                    dcq0 = __builtin_amdgcn_alignbit(dcq1, dcq0, 16u); dcq1 = __builtin_amdgcn_alignbit(dcq2, dcq1, 16u); dcq2 = __builtin_amdgcn_alignbit(dcq3, dcq2, 16u);
                    dcq3 = (dcq3 >> 16) | (dq0 << 16); dccnt++; dclast = fblk;
                }
                if (WBALLOT(dccnt >= 8u)) {
                    if (dccnt >= 8u) {
                        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
                        u32x4_t t; t.x = dcq0; t.y = dcq1; t.z = dcq2; t.w = dcq3;
                        *reinterpret_cast<u32x4_t*>(dbase + (dclast - 7u)) = t;
                        dccnt = 0;
                    }
                }
            }
        }
    }
    dc_store_rest();
    if (verify) {
        if (check_n && !IBAL(m_cap) && res_p != P_END) { res_p = cur.p; res_s = cur.p == P_END ? 0u : ST_MAKE(seg, c, k); res_n = nblk; }
        // the chain must be at its fixed point, and the block count that fed the prefix sum must be the real one
        const uint32_t want_p = HALF && !hi ? half_p[g] : A.out_p[g], want_s = HALF && !hi ? half_s[g] : A.out_s[g];
        const uint32_t want_n = HALF ? (hi ? A.nblk[g] - half_n[g] : half_n[g]) : A.nblk[g];
        if (res_p != want_p || res_s != want_s || (check_n && res_n != want_n)) fl |= F_NOSYNC;
    }
    if (fl) { FLAG_OR(flags, img, fl); if (an != 0xFFFFFFFFu) ANOM_MIN(flags, img, an); }
    if (REC) {
        __syncthreads();
        uint32_t* ho = side + im.side_off + JS_SIDE_HISTO;
        for (uint32_t q = threadIdx.x; q < 2 * 4 * 17; q += SY_THREADS) { const uint32_t v = s_histo[q]; if (v) atomicAdd(&ho[q], v); }
    }
}

// One workgroup (1024 lanes) per image: DC differences (in dccum, decode order) -> cumulative DC per block.
// int16 wrapping sums per component (m_nDcLum += ..., :3280/:3355/:3386), reset at every MCU the write pass marked as
// the first of a restart interval (DecodeRestartDcState :2693).  One lane per MCU, 1024 MCUs per step: the lanes of a
// wave read one contiguous stretch of the DC array; a segmented inclusive scan (wave shuffles, then the 16 wave totals)
// gives every MCU the sums that enter it, and a running carry links the steps.
#define DC_THREADS 1024
struct DcSeg { int s0, s1, s2; int r; };                            // sums since the last reset inside the span, reset seen
__device__ __forceinline__ DcSeg dc_combine(const DcSeg& a, const DcSeg& b) { DcSeg o; if (b.r) o = b; else { o.s0 = a.s0 + b.s0; o.s1 = a.s1 + b.s1; o.s2 = a.s2 + b.s2; o.r = a.r; } return o; }
// MCUs [m_begin, m_end) of the image; `carry` = the sums that enter m_begin.  WRITE: store the cumulative values; else only
// summarise the range (sums since its last reset, reset seen) into *summary -- the first level of the two-level scan of small jobs.
template <int NBMAX, bool WRITE>                                     // blocks per MCU this instance is unrolled for
__device__ __forceinline__ void dc_scan_range(const JsImage& im, int16_t* __restrict__ d, const uint8_t* __restrict__ rf, DcSeg* s_w,
                                              uint32_t m_begin, uint32_t m_end, DcSeg carry, DcSeg* summary)
{
    const uint32_t nb = im.blk_per_mcu;
    const uint32_t n1 = im.samp_h[1] * im.samp_v[1], n2 = im.ncomp == 3 ? n1 + im.samp_h[2] * im.samp_v[2] : nb;
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int any_reset = 0;
    for (uint32_t base = m_begin; base < m_end; base += DC_THREADS) {
        const uint32_t m = base + t; const bool valid = m < m_end;
        // the MCU's restart mark: 0 = none, j + 1 = the predictors are cleared in front of its block j (a marker on the MCU boundary: 1;
        // a marker the reference met inside the MCU -- a damaged interval -- : the block that was in progress, walk_slow)
        const uint32_t rw = valid ? rf[m] : 0u, rj = rw & 63u;   // (bit 7: met inside the block; bit 6: a second reset in front of block 0 -- see mark_reset)
        const bool r0 = (rw & 64u) != 0u;
        int v[NBMAX];
        DcSeg own = { 0, 0, 0, rj ? 1 : 0 };                     // the MCU as a scan element: sums since its reset if it has one, else of all its blocks
        #pragma unroll
        for (uint32_t c = 0; c < NBMAX; c++) {
            v[c] = (valid && c < nb) ? (int)d[(size_t)m * nb + c] : 0;
            const int a = (rj && c + 1u < rj) ? 0 : v[c];        // in front of the reset: not part of what the MCU hands on
            if (c < n1) own.s0 += a; else if (c < n2) own.s1 += a; else own.s2 += a;
        }
        DcSeg inc = own;                                         // inclusive scan over the lanes of the wave
        #pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            DcSeg a; a.s0 = __shfl_up(inc.s0, off); a.s1 = __shfl_up(inc.s1, off); a.s2 = __shfl_up(inc.s2, off); a.r = __shfl_up(inc.r, off);
            if (lane >= (uint32_t)off) inc = dc_combine(a, inc);
        }
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        DcSeg pre = carry, tot = carry;                          // what enters this wave / leaves the step
        for (uint32_t w = 0; w < DC_THREADS / 64; w++) { const DcSeg x = s_w[w]; if (w < wave) pre = dc_combine(pre, x); tot = dc_combine(tot, x); }
        // sums entering this MCU: everything in front of it -- the wave's exclusive prefix behind what enters the wave
        DcSeg exc; exc.s0 = __shfl_up(inc.s0, 1); exc.s1 = __shfl_up(inc.s1, 1); exc.s2 = __shfl_up(inc.s2, 1); exc.r = __shfl_up(inc.r, 1);
        if (lane == 0) { exc.s0 = exc.s1 = exc.s2 = 0; exc.r = 0; }
        const DcSeg in = dc_combine(pre, exc);
        int c0 = in.s0, c1 = in.s1, c2 = in.s2;
        if (WRITE && valid) {
            #pragma unroll
            for (uint32_t c = 0; c < NBMAX; c++) if (c < nb) {
                if (c + 1u == rj || (c == 0 && r0)) { c0 = 0; c1 = 0; c2 = 0; }
                int16_t o;
                if (c < n1) { c0 += v[c]; o = (int16_t)c0; } else if (c < n2) { c1 += v[c]; o = (int16_t)c1; } else { c2 += v[c]; o = (int16_t)c2; }
                d[(size_t)m * nb + c] = o;
            }
        }
        any_reset |= tot.r;
        carry = tot; carry.r = 0;
        __syncthreads();
    }
    if (!WRITE && t == 0) { carry.r = any_reset; *summary = carry; }
}
__global__ void __launch_bounds__(DC_THREADS) k_dc_scan(const JsImage* __restrict__ imgs, const JsTableSet* __restrict__ tables,
                                                        int16_t* __restrict__ dccum, const uint8_t* __restrict__ mcu_rst)
{
    const uint32_t img = blockIdx.x; const JsImage& im = imgs[img];
    if (!tables[im.tableset].lut_ok) return;
    __shared__ DcSeg s_w[DC_THREADS / 64];
    int16_t* d = dccum + im.coef_off; const uint8_t* rf = mcu_rst + im.mcu_off;
    const DcSeg zero = { 0, 0, 0, 0 };
    if (im.blk_per_mcu <= 6) dc_scan_range<6, true>(im, d, rf, s_w, 0, im.mcu_xmax * im.mcu_ymax, zero, nullptr);   // 4:4:4, 4:2:2, 4:2:0, grayscale: a short unrolled body
    else dc_scan_range<JS_MAX_BLK_PER_MCU, true>(im, d, rf, s_w, 0, im.mcu_xmax * im.mcu_ymax, zero, nullptr);
}
// Small jobs (a single large image): the MCUs of an image are cut into DC_PARTS_MAX ranges, one workgroup each.  Level 1
// (apply = 0) summarises every range; level 2 (apply = 1) folds the summaries of the ranges before its own into the sums that
// enter it and writes.  Two launches instead of one serial pass of nmcu / 1024 steps.
#define DC_PARTS_MAX 64
__global__ void __launch_bounds__(DC_THREADS) k_dc_scan_parts(const JsImage* __restrict__ imgs, const JsTableSet* __restrict__ tables,
                                                              int16_t* __restrict__ dccum, const uint8_t* __restrict__ mcu_rst, DcSeg* __restrict__ summaries, int apply)
{
    const uint32_t img = blockIdx.y, part = blockIdx.x; const JsImage& im = imgs[img];
    if (!tables[im.tableset].lut_ok) return;
    const uint32_t nmcu = im.mcu_xmax * im.mcu_ymax;
    const uint32_t per = ((nmcu + DC_PARTS_MAX - 1) / DC_PARTS_MAX + DC_THREADS - 1) / DC_THREADS * DC_THREADS;   // whole steps per range
    const uint32_t m0 = part * per, m1 = min(nmcu, m0 + per);
    DcSeg* sm = summaries + (size_t)img * DC_PARTS_MAX;
    if (m0 >= nmcu) { if (!apply && threadIdx.x == 0) { const DcSeg z = { 0, 0, 0, 0 }; sm[part] = z; } return; }
    __shared__ DcSeg s_w[DC_THREADS / 64];
    int16_t* d = dccum + im.coef_off; const uint8_t* rf = mcu_rst + im.mcu_off;
    DcSeg carry = { 0, 0, 0, 0 };
    if (apply) { for (uint32_t q = 0; q < part; q++) carry = dc_combine(carry, sm[q]); carry.r = 0; }
    const bool small = im.blk_per_mcu <= 6;
    if (!apply) { if (small) dc_scan_range<6, false>(im, d, rf, s_w, m0, m1, carry, sm + part); else dc_scan_range<JS_MAX_BLK_PER_MCU, false>(im, d, rf, s_w, m0, m1, carry, sm + part); }
    else        { if (small) dc_scan_range<6, true>(im, d, rf, s_w, m0, m1, carry, nullptr);  else dc_scan_range<JS_MAX_BLK_PER_MCU, true>(im, d, rf, s_w, m0, m1, carry, nullptr); }
}

void js_launch_unstuff(hipStream_t st, int wl, const JsImage* imgs, const uint32_t* us_base, uint32_t nimg, uint32_t total_chunks, const uint8_t* raw,
                       uint32_t* chunk_keep, uint32_t* chunk_rst, uint8_t* ustr_lin, uint8_t* ustr, uint32_t* seg_tab, uint32_t* side, uint32_t* flags,
                       const uint32_t* sy_base, uint32_t sy_wgs, unsigned long long* us_state, uint32_t epoch, const uint32_t* us4_base, uint32_t total_super,
                       uint32_t* ticket, uint32_t* ticket_base /*host: the counter's value, advanced by what this launch takes*/)
{
    if (!total_chunks) return;
    if (us_state && wl != 4 && us4_base) {   // large jobs: one pass from the file bytes to the interleaved layout (k_unstuff_fused), no transposition pass
#define JS_USF(W) hipLaunchKernelGGL(k_unstuff_fused<W>, dim3(total_super), dim3(US_THREADS), 0, st, imgs, us_base, us4_base, nimg, raw, chunk_keep, chunk_rst, ustr, seg_tab, us_state, epoch, side, flags, ticket, *ticket_base)
        if (wl == 5) JS_USF(5); else if (wl == 6) JS_USF(6); else if (wl == 7) JS_USF(7); else JS_USF(8);
#undef JS_USF
        *ticket_base += total_super;
        return;
    }
    if (us_state) {          // small jobs (64-byte pieces read the linear stream): one pass, the chunks' places come from a decoupled look-back (k_unstuff_write<true>)
        hipLaunchKernelGGL(k_unstuff_write<true>, dim3(total_chunks), dim3(US_THREADS), 0, st, imgs, us_base, nimg, raw, chunk_keep, chunk_rst, wl == 4 ? ustr : ustr_lin, seg_tab, 0u,
                           (uint32_t*)nullptr, us_state, epoch, side, flags, ticket, *ticket_base);
        *ticket_base += total_chunks;
    } else {                 // the three-pass form (cross-check): count, scan per image, write
        hipLaunchKernelGGL(k_unstuff_count, dim3(total_chunks), dim3(US_THREADS), 0, st, imgs, us_base, nimg, raw, chunk_keep, chunk_rst);
        hipLaunchKernelGGL(k_unstuff_scan, dim3(nimg), dim3(256), 0, st, imgs, us_base, chunk_keep, chunk_rst, seg_tab, side, flags);
        hipLaunchKernelGGL(k_unstuff_write<false>, dim3(total_chunks), dim3(US_THREADS), 0, st, imgs, us_base, nimg, raw, chunk_keep, chunk_rst, wl == 4 ? ustr : ustr_lin, seg_tab, 0u,
                           (uint32_t*)nullptr, (unsigned long long*)nullptr, 0u, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u);
    }
    if (wl == 4) return;                                         // (phys_word<4> is the identity)
    if (wl == 6) hipLaunchKernelGGL(k_interleave<6>, dim3(sy_wgs * 4), dim3(256), 0, st, imgs, sy_base, nimg, side, ustr_lin, ustr);
    else if (wl == 8) hipLaunchKernelGGL(k_interleave<8>, dim3(sy_wgs * 4), dim3(256), 0, st, imgs, sy_base, nimg, side, ustr_lin, ustr);
    else if (wl == 7) hipLaunchKernelGGL(k_interleave<7>, dim3(sy_wgs * 4), dim3(256), 0, st, imgs, sy_base, nimg, side, ustr_lin, ustr);
    else hipLaunchKernelGGL(k_interleave<5>, dim3(sy_wgs * 4), dim3(256), 0, st, imgs, sy_base, nimg, side, ustr_lin, ustr);
}
static SubArrays sub_arrays(uint32_t* sub, uint64_t n) { SubArrays a; a.out_p = sub; a.out_s = sub + n; a.in_p = sub + 2 * n; a.in_s = sub + 3 * n; a.nblk = sub + 4 * n; a.base = sub + 5 * n; return a; }
static size_t subtabs_bytes_host(uint32_t tab_rows, uint32_t tab_lut2, bool pairs = false)
{ return pairs ? (size_t)tab_rows * (4u << JS_L1_BITS) + (size_t)tab_lut2 * 4 : (size_t)tab_rows * (2u << JS_L1_BITS) + (((size_t)tab_lut2 * 2 + 15) & ~15ull) + 3 * 64 * 4; }   // (load_subtabs)
void js_launch_sync(hipStream_t st, int wl, uint32_t tab_rows, uint32_t tab_lut2, const JsImage* imgs, const uint32_t* sy_base, uint32_t nimg, uint32_t total_wgs, const JsTableSet* tables,
                    const uint8_t* ustr, const uint32_t* seg_tab, const uint32_t* side, uint32_t* sub, uint64_t nsub, int first_pass, uint32_t it_max)
{
    if (!total_wgs) return;
    if (wl == 4) hipLaunchKernelGGL(k_sync<4>, dim3(total_wgs), dim3(SY_THREADS), subtabs_bytes_host(tab_rows, tab_lut2, true), st, imgs, sy_base, nimg, tables, ustr, seg_tab, side,
                       sub_arrays(sub, nsub), first_pass, tab_rows, tab_lut2, it_max);
    else if (wl == 6) hipLaunchKernelGGL(k_sync<6>, dim3(total_wgs), dim3(SY_THREADS), subtabs_bytes_host(tab_rows, tab_lut2, true), st, imgs, sy_base, nimg, tables, ustr, seg_tab, side,
                       sub_arrays(sub, nsub), first_pass, tab_rows, tab_lut2, it_max);
    else if (wl == 8) hipLaunchKernelGGL(k_sync<8>, dim3(total_wgs), dim3(SY_THREADS), subtabs_bytes_host(tab_rows, tab_lut2, true), st, imgs, sy_base, nimg, tables, ustr, seg_tab, side,
                       sub_arrays(sub, nsub), first_pass, tab_rows, tab_lut2, it_max);
    else if (wl == 7) hipLaunchKernelGGL(k_sync<7>, dim3(total_wgs), dim3(SY_THREADS), subtabs_bytes_host(tab_rows, tab_lut2, true), st, imgs, sy_base, nimg, tables, ustr, seg_tab, side,
                       sub_arrays(sub, nsub), first_pass, tab_rows, tab_lut2, it_max);
    else hipLaunchKernelGGL(k_sync<5>, dim3(total_wgs), dim3(SY_THREADS), subtabs_bytes_host(tab_rows, tab_lut2, true), st, imgs, sy_base, nimg, tables, ustr, seg_tab, side,
                       sub_arrays(sub, nsub), first_pass, tab_rows, tab_lut2, it_max);
}
size_t js_sync_list_words(uint64_t nsub, uint32_t nimg) { return (size_t)2 * nsub + (size_t)nimg * SYR_SLOTS; }
// The large-job form: first launch of k_sync cut after its second round, then `rounds` list rounds (<= SYR_SLOTS - 1) and the verification mode.
// sn_base / sn_wgs: k_sync's workgroup prefix (254 owned sub-sequences each); sy_base / sy_wgs: 256 per workgroup.  lists: js_sync_list_words() words, the
// counters behind the two lists (this part's images: lists + 2 * nsub + first image * SYR_SLOTS) are zeroed here.
void js_launch_sync_rounds(hipStream_t st, int wl, uint32_t tab_rows, uint32_t tab_lut2, const JsImage* imgs, const uint32_t* sn_base, uint32_t sn_wgs, const uint32_t* sy_base, uint32_t sy_wgs,
                           uint32_t nimg, const JsTableSet* tables, const uint8_t* ustr, const uint32_t* seg_tab, const uint32_t* side, uint32_t* sub, uint64_t nsub,
                           uint32_t* lists, uint32_t* rcnt, int rounds)
{
    if (!sn_wgs || !sy_wgs) return;
    if (rounds > SYR_SLOTS - 1) rounds = SYR_SLOTS - 1;
    (void)hipMemsetAsync(rcnt, 0, (size_t)nimg * SYR_SLOTS * 4, st);
    js_launch_sync(st, wl, tab_rows, tab_lut2, imgs, sn_base, nimg, sn_wgs, tables, ustr, seg_tab, side, sub, nsub, 1, 2u);
    const SubArrays A = sub_arrays(sub, nsub);
    const size_t lds = subtabs_bytes_host(tab_rows, tab_lut2, true);
    uint32_t* l0 = lists; uint32_t* l1 = lists + nsub;
#define SYR_WL(K, GRID, BLOCK, LDS, ...) \
    do { if (wl == 4) hipLaunchKernelGGL(K<4>, GRID, BLOCK, LDS, st, __VA_ARGS__); else if (wl == 5) hipLaunchKernelGGL(K<5>, GRID, BLOCK, LDS, st, __VA_ARGS__); \
         else if (wl == 6) hipLaunchKernelGGL(K<6>, GRID, BLOCK, LDS, st, __VA_ARGS__); else if (wl == 7) hipLaunchKernelGGL(K<7>, GRID, BLOCK, LDS, st, __VA_ARGS__); \
         else hipLaunchKernelGGL(K<8>, GRID, BLOCK, LDS, st, __VA_ARGS__); } while (0)
    SYR_WL(k_sync_links, dim3(sy_wgs), dim3(256), 0, imgs, sy_base, nimg, tables, side, A, l0, rcnt);
    for (int r = 0; r < rounds; r++)
        SYR_WL(k_sync_round, dim3(sy_wgs), dim3(SY_THREADS), lds, imgs, sy_base, nimg, tables, ustr, seg_tab, side, A, (const uint32_t*)((r & 1) ? l1 : l0), (r & 1) ? l0 : l1, rcnt, (uint32_t)r, tab_rows, tab_lut2);
#undef SYR_WL
    js_launch_sync(st, wl, tab_rows, tab_lut2, imgs, sn_base, nimg, sn_wgs, tables, ustr, seg_tab, side, sub, nsub, 2, 0u);
}
// ---- candidate synchronisation (small jobs).  cand: js_cand_bytes(nsub) bytes; req: nimg * JS_CAND_REQ_WORDS words
static CandArrays cand_arrays(uint32_t* c, uint64_t n)
{
    CandArrays C; C.n = n; C.xp = c; C.xs = c + CD_H * n; uint32_t* m = c + 2 * CD_H * n;
    C.mep = m; C.mes = m + CD_SLOTS * n; C.mxp = m + 2 * CD_SLOTS * n; C.mxs = m + 3 * CD_SLOTS * n; C.mnb = m + 4 * CD_SLOTS * n;
    C.mmp = m + 5 * CD_SLOTS * n; C.mms = m + 6 * CD_SLOTS * n; C.mmn = m + 7 * CD_SLOTS * n;
    uint32_t* r = m + 8 * CD_SLOTS * n;
    C.map = reinterpret_cast<uint2*>(r); C.hp = r + 2 * n; C.hs = r + 3 * n; C.hn = r + 4 * n; C.sel = reinterpret_cast<uint8_t*>(r + 5 * n);
    return C;
}
size_t js_cand_bytes(uint64_t nsub) { return (size_t)nsub * ((2 * CD_H + 8 * CD_SLOTS + 5) * 4 + 2) + 8192; }   // (slack: the chain reads whole 512-map tiles)
static_assert(CD_DIAG_WORDS == JS_CAND_REQ_WORDS, "diagnostics area size");
static_assert(CD_H == JS_CAND_MAX_BLK, "hypotheses");
#define CAND_WL(K, GRID, BLOCK, LDS, ...) \
    do { if (wl == 4) hipLaunchKernelGGL(K<4>, GRID, BLOCK, LDS, st, __VA_ARGS__); else if (wl == 5) hipLaunchKernelGGL(K<5>, GRID, BLOCK, LDS, st, __VA_ARGS__); \
         else if (wl == 6) hipLaunchKernelGGL(K<6>, GRID, BLOCK, LDS, st, __VA_ARGS__); else if (wl == 7) hipLaunchKernelGGL(K<7>, GRID, BLOCK, LDS, st, __VA_ARGS__); \
         else hipLaunchKernelGGL(K<8>, GRID, BLOCK, LDS, st, __VA_ARGS__); } while (0)
#define CAND_WL2(K, GRID, BLOCK, LDS, ...) \
    do { if (wl == 4) hipLaunchKernelGGL((K<4, false>), GRID, BLOCK, LDS, st, __VA_ARGS__); else if (wl == 5) hipLaunchKernelGGL((K<5, false>), GRID, BLOCK, LDS, st, __VA_ARGS__); \
         else if (wl == 6) hipLaunchKernelGGL((K<6, false>), GRID, BLOCK, LDS, st, __VA_ARGS__); else if (wl == 7) hipLaunchKernelGGL((K<7, false>), GRID, BLOCK, LDS, st, __VA_ARGS__); \
         else hipLaunchKernelGGL((K<8, false>), GRID, BLOCK, LDS, st, __VA_ARGS__); } while (0)
void js_launch_cand_sync(hipStream_t st, int wl, uint32_t tab_rows, uint32_t tab_lut2, const JsImage* imgs, const uint32_t* sy_base, uint32_t nimg, uint32_t sy_wgs, uint32_t max_blk,
                         const JsTableSet* tables, const uint8_t* ustr, const uint32_t* seg_tab, const uint32_t* side, uint32_t* sub, uint64_t nsub, uint32_t* cand, uint32_t* req, int fill_rounds,
                         int mid /* the memo walks report the state at the middle of every sub-sequence too (64-byte pieces only): the write pass will run two lanes per sub-sequence */)
{
    if (!sy_wgs || !nimg) return;
    const size_t lds = subtabs_bytes_host(tab_rows, tab_lut2, true);
    const CandArrays C = cand_arrays(cand, nsub);
    CAND_WL(k_cand_spec, dim3(sy_wgs, max_blk), dim3(SY_THREADS), lds, imgs, sy_base, nimg, tables, ustr, seg_tab, side, C, tab_rows, tab_lut2);
    if (mid && wl == 4) {
        hipLaunchKernelGGL((k_cand_walk<4, true>), dim3(sy_wgs, CD_H), dim3(SY_THREADS), lds, st, imgs, sy_base, nimg, tables, ustr, seg_tab, side, C, tab_rows, tab_lut2);
        hipLaunchKernelGGL((k_cand_chain<4, true>), dim3(nimg), dim3(CC_THREADS), lds, st, imgs, tables, ustr, seg_tab, side, C, req, fill_rounds, tab_rows, tab_lut2);
    } else {
        mid = 0;
        CAND_WL2(k_cand_walk, dim3(sy_wgs, CD_H), dim3(SY_THREADS), lds, imgs, sy_base, nimg, tables, ustr, seg_tab, side, C, tab_rows, tab_lut2);
        CAND_WL2(k_cand_chain, dim3(nimg), dim3(CC_THREADS), lds, imgs, tables, ustr, seg_tab, side, C, req, fill_rounds, tab_rows, tab_lut2);
    }
    CAND_WL(k_cand_apply, dim3(sy_wgs), dim3(SY_THREADS), 0, imgs, sy_base, nimg, tables, side, C, sub_arrays(sub, nsub), mid);
}
void js_launch_block_scan(hipStream_t st, int wl, const JsImage* imgs, uint32_t nimg, const JsTableSet* tables, uint32_t* sub, uint64_t nsub, uint32_t* side, uint32_t* flags)
{
    if (!nimg) return;
    if (nimg <= 8) {                                             // a few (large) images: wide workgroups, fewer serial steps
        if (wl == 4) hipLaunchKernelGGL((k_block_scan<4, 1024>), dim3(nimg), dim3(1024), 0, st, imgs, tables, sub_arrays(sub, nsub), side, flags);
    else if (wl == 6) hipLaunchKernelGGL((k_block_scan<6, 1024>), dim3(nimg), dim3(1024), 0, st, imgs, tables, sub_arrays(sub, nsub), side, flags);
    else if (wl == 8) hipLaunchKernelGGL((k_block_scan<8, 1024>), dim3(nimg), dim3(1024), 0, st, imgs, tables, sub_arrays(sub, nsub), side, flags);
    else if (wl == 7) hipLaunchKernelGGL((k_block_scan<7, 1024>), dim3(nimg), dim3(1024), 0, st, imgs, tables, sub_arrays(sub, nsub), side, flags);
        else hipLaunchKernelGGL((k_block_scan<5, 1024>), dim3(nimg), dim3(1024), 0, st, imgs, tables, sub_arrays(sub, nsub), side, flags);
    } else if (wl == 4) hipLaunchKernelGGL((k_block_scan<4, 256>), dim3(nimg), dim3(256), 0, st, imgs, tables, sub_arrays(sub, nsub), side, flags);
    else if (wl == 6) hipLaunchKernelGGL((k_block_scan<6, 256>), dim3(nimg), dim3(256), 0, st, imgs, tables, sub_arrays(sub, nsub), side, flags);
    else if (wl == 8) hipLaunchKernelGGL((k_block_scan<8, 256>), dim3(nimg), dim3(256), 0, st, imgs, tables, sub_arrays(sub, nsub), side, flags);
    else if (wl == 7) hipLaunchKernelGGL((k_block_scan<7, 256>), dim3(nimg), dim3(256), 0, st, imgs, tables, sub_arrays(sub, nsub), side, flags);
    else hipLaunchKernelGGL((k_block_scan<5, 256>), dim3(nimg), dim3(256), 0, st, imgs, tables, sub_arrays(sub, nsub), side, flags);
}
void js_launch_write(hipStream_t st, int wl, uint32_t tab_rows, uint32_t tab_lut2, const JsImage* imgs, const uint32_t* sy_base, uint32_t nimg, uint32_t total_wgs, const JsTableSet* tables,
                     const uint8_t* ustr, const uint32_t* seg_tab, uint32_t* side, uint32_t* sub, uint64_t nsub,
                     int16_t* coef, int16_t* dccum, uint8_t* mcu_rst, uint32_t* flags, uint32_t* cand_half, bool v1 /* the first form of the kernel, kept as a cross-check */,
                     uint32_t* rec_pos /* null, or (64-byte pieces, second form only -- js_write_can_record) the MCU-top positions the pass is to record along with the histogram */)
{
    if (!total_wgs) return;
    if (!v1 && cand_half && wl == 4) {                               // two lanes per sub-sequence, the second from the middle state of the selected memo walk
        const CandArrays C = cand_arrays(cand_half, nsub);
        if (rec_pos) hipLaunchKernelGGL((k_write2<4, true, true>), dim3(2 * total_wgs), dim3(SY_THREADS), wtabs_bytes(tab_rows, tab_lut2), st, imgs, sy_base, nimg, tables, ustr, seg_tab, side,
                           sub_arrays(sub, nsub), coef, dccum, mcu_rst, flags, tab_rows, tab_lut2, (const uint32_t*)C.hp, (const uint32_t*)C.hs, (const uint32_t*)C.hn, rec_pos);
        else hipLaunchKernelGGL((k_write2<4, true>), dim3(2 * total_wgs), dim3(SY_THREADS), wtabs_bytes(tab_rows, tab_lut2), st, imgs, sy_base, nimg, tables, ustr, seg_tab, side,
                           sub_arrays(sub, nsub), coef, dccum, mcu_rst, flags, tab_rows, tab_lut2, (const uint32_t*)C.hp, (const uint32_t*)C.hs, (const uint32_t*)C.hn);
        return;
    }
    if (!v1) {
        if (wl == 4 && rec_pos) hipLaunchKernelGGL((k_write2<4, false, true>), dim3(total_wgs), dim3(SY_THREADS), wtabs_bytes(tab_rows, tab_lut2), st, imgs, sy_base, nimg, tables, ustr, seg_tab, side,
                           sub_arrays(sub, nsub), coef, dccum, mcu_rst, flags, tab_rows, tab_lut2, (const uint32_t*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr, rec_pos);
        else if (wl == 4) hipLaunchKernelGGL((k_write2<4>), dim3(total_wgs), dim3(SY_THREADS), wtabs_bytes(tab_rows, tab_lut2), st, imgs, sy_base, nimg, tables, ustr, seg_tab, side,
                           sub_arrays(sub, nsub), coef, dccum, mcu_rst, flags, tab_rows, tab_lut2);
    else if (wl == 6) hipLaunchKernelGGL((k_write2<6>), dim3(total_wgs), dim3(SY_THREADS), wtabs_bytes(tab_rows, tab_lut2), st, imgs, sy_base, nimg, tables, ustr, seg_tab, side,
                           sub_arrays(sub, nsub), coef, dccum, mcu_rst, flags, tab_rows, tab_lut2);
    else if (wl == 8) hipLaunchKernelGGL((k_write2<8>), dim3(total_wgs), dim3(SY_THREADS), wtabs_bytes(tab_rows, tab_lut2), st, imgs, sy_base, nimg, tables, ustr, seg_tab, side,
                           sub_arrays(sub, nsub), coef, dccum, mcu_rst, flags, tab_rows, tab_lut2);
    else if (wl == 7) hipLaunchKernelGGL((k_write2<7>), dim3(total_wgs), dim3(SY_THREADS), wtabs_bytes(tab_rows, tab_lut2), st, imgs, sy_base, nimg, tables, ustr, seg_tab, side,
                           sub_arrays(sub, nsub), coef, dccum, mcu_rst, flags, tab_rows, tab_lut2);
        else hipLaunchKernelGGL((k_write2<5>), dim3(total_wgs), dim3(SY_THREADS), wtabs_bytes(tab_rows, tab_lut2), st, imgs, sy_base, nimg, tables, ustr, seg_tab, side,
                           sub_arrays(sub, nsub), coef, dccum, mcu_rst, flags, tab_rows, tab_lut2);
        return;
    }
    if (wl == 4) hipLaunchKernelGGL((k_write<4, false>), dim3(total_wgs), dim3(SY_THREADS), wtabs_bytes(tab_rows, tab_lut2), st, imgs, sy_base, nimg, tables, ustr, seg_tab, side,
                       sub_arrays(sub, nsub), coef, dccum, mcu_rst, flags, tab_rows, tab_lut2, 0u, (uint32_t*)nullptr);
    else if (wl == 6) hipLaunchKernelGGL((k_write<6, false>), dim3(total_wgs), dim3(SY_THREADS), wtabs_bytes(tab_rows, tab_lut2), st, imgs, sy_base, nimg, tables, ustr, seg_tab, side,
                       sub_arrays(sub, nsub), coef, dccum, mcu_rst, flags, tab_rows, tab_lut2, 0u, (uint32_t*)nullptr);
    else if (wl == 8) hipLaunchKernelGGL((k_write<8, false>), dim3(total_wgs), dim3(SY_THREADS), wtabs_bytes(tab_rows, tab_lut2), st, imgs, sy_base, nimg, tables, ustr, seg_tab, side,
                       sub_arrays(sub, nsub), coef, dccum, mcu_rst, flags, tab_rows, tab_lut2, 0u, (uint32_t*)nullptr);
    else if (wl == 7) hipLaunchKernelGGL((k_write<7, false>), dim3(total_wgs), dim3(SY_THREADS), wtabs_bytes(tab_rows, tab_lut2), st, imgs, sy_base, nimg, tables, ustr, seg_tab, side,
                       sub_arrays(sub, nsub), coef, dccum, mcu_rst, flags, tab_rows, tab_lut2, 0u, (uint32_t*)nullptr);
    else hipLaunchKernelGGL((k_write<5, false>), dim3(total_wgs), dim3(SY_THREADS), wtabs_bytes(tab_rows, tab_lut2), st, imgs, sy_base, nimg, tables, ustr, seg_tab, side,
                       sub_arrays(sub, nsub), coef, dccum, mcu_rst, flags, tab_rows, tab_lut2, 0u, (uint32_t*)nullptr);
}
// =====================================================================================
//  Side outputs of an image the parallel path decoded, without the sequential kernel:
//   * k_unstuff_write (side mode)  -> compacted-byte index of every 16-byte group of the scan  (inverse map)
//   * k_write<WL, true>            -> bit position at which every MCU starts, Huffman code-length histogram
//   * k_side_maps                  -> m_pMcuFileMap (:3229, PackFileOffset :5104), the three block-DC maps (:3524-3608) and the
//                                     status words the reference is left with after the last MCU.  Where the answer depends on
//                                     the register's bookkeeping rather than on the bit position alone (register run empty in
//                                     front of an RSTn; end of the scan) the exact-mirror reader is run over the few MCUs before.
// =====================================================================================
// File offset (relative to the file start) of byte `u` of the compacted stream of image `im`.
__device__ uint32_t raw_of_compacted(const JsImage& im, const uint8_t* __restrict__ raw, const uint32_t* __restrict__ us_out, uint32_t nthreads, uint32_t u)
{
    uint32_t lo = 0, hi = nthreads;                                // last group whose first kept byte has index <= u
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (us_out[mid] <= u) lo = mid; else hi = mid; }
    const uint64_t s = im.file_off + im.scan_start, e = s + im.scan_len;
    const uint64_t o16 = (s & ~15ull) + (uint64_t)lo * 16;
    const UsBytes c = us_classify(raw, o16, s, e);
    uint32_t m = c.keep_mask;
    for (uint32_t k = u - us_out[lo]; k && m; k--) m &= m - 1;       // drop the k lowest kept bytes
    if (!m) return im.scan_start + im.scan_len;                     // past the end of the scan data
    return (uint32_t)(o16 + (uint32_t)__builtin_ctz(m) - im.file_off);
}

// Brings an exact-mirror reader to the state the reference's reader has when its MCU loop reaches MCU `m_top` (m_top ==
// number of MCUs: after the last one).  The register state at an MCU top is a function of the bit position alone --
// four bytes buffered from byte(p), p & 7 bits of the first consumed -- PROVIDED the refill that follows was not cut
// short by a restart marker; otherwise slots of the position array keep older values that can surface later (the
// register's bookkeeping shifts them down when it runs empty, ScanBuffConsume :921-955).  The walk therefore starts at
// the closest earlier MCU top that either begins a restart interval (freshly restarted buffer, DecodeRestartScanBuf
// :4038) or lies at least 5 bytes before the next RSTn, and decodes forward from there with the mirror itself.
// Returns the number of RSTn markers the reference has seen before the point the walk started from.
__device__ __forceinline__ uint32_t mirror_to_mcu_top(const JsImage& im, const JsTableSet* __restrict__ tables, const uint8_t* __restrict__ raw, const uint32_t* __restrict__ st,
                                      uint32_t nseg, uint32_t total_bytes, const uint8_t* __restrict__ mcu_rst, const uint32_t* __restrict__ mcu_pos,
                                      const uint32_t* __restrict__ us_out, uint32_t us_threads, uint32_t m_top, ExactReader& r, uint32_t* dummy_histo, int16_t* scratch,
                                      uint32_t* events, const ExactReader* lds_tabs = nullptr, uint32_t p_last_blk = 0 /* m_top == number of MCUs: bit position of the last block's top, or 0 */)
{
    uint32_t* meta12 = dummy_histo + 2 * 4 * 17;                  // the caller's scratch has room for the 12 slot words behind the histogram
    r.file = raw + im.file_off; r.flen = im.file_len; r.ts = tables + im.tableset; r.histo = dummy_histo; r.fast = &r.ts->fast[0][0]; r.meta = meta12; r.q = &r.ts->qzz[0][0]; r.zz = c_zigzag; r.win_at = 0xFFFFFFFFu; r.win = 0;
    for (int i = 0; i < 6; i++) { meta12[i] = r.ts->size[i]; meta12[6 + i] = r.ts->dest_id[i]; }
    if (lds_tabs) { r.fast = lds_tabs->fast; r.q = lds_tabs->q; r.zz = lds_tabs->zz; }                  // (copies of the same tables in LDS: k_side_chunks)
    // the markers the look-ahead runs into at the end of the scan (":  Scan Data encountered marker", :1536) are logged from here
    r.ev = (events && im.ev_cap) ? events + im.ev_off : nullptr; r.ev_cap = im.ev_cap; r.ev_only = JS_EV_MARKER; r.ev_end = m_top == im.mcu_xmax * im.mcu_ymax ? 1u : 0u;
    r.rst_interval = im.rst_interval; r.precision = im.precision; r.err_max = im.err_max; r.warn_bad = 0; r.warn_marker = 0;
    r.rst_count = 0; r.rst_last = 0; r.rst_expect = 0; r.rst_handled = 0;
    uint32_t m0 = m_top ? m_top - 1 : 0, rst_before = 0, c0 = 0;
    // The end of the scan needs the reader only for what the look-ahead meets there: it may start at the top of the last BLOCK (the state at any symbol
    // boundary is a function of the bit position, as at an MCU top) -- a sixth of the symbols of a 4:2:0 MCU, at 1.3 us each on one lane.
    bool from_blk = false;
    if (p_last_blk && im.blk_per_mcu > 1u && m_top) {
        const uint32_t sg = find_interval(st, nseg, p_last_blk >> 3);
        if ((sg + 1 >= nseg || st[sg + 1] * 8 >= p_last_blk + 40) && p_last_blk > mcu_pos[m_top - 1]) {
            rst_before = sg; c0 = im.blk_per_mcu - 1u; from_blk = true;
            ex_restart_scan_buf(r, raw_of_compacted(im, raw, us_out, us_threads, p_last_blk >> 3), true); ex_topup(r); ex_consume(r, p_last_blk & 7u);
        }
    }
    if (!from_blk)
    for (;; m0--) {
        if (m0 == 0) { ex_restart_scan_buf(r, im.scan_start, false); ex_topup(r); break; }
        const uint32_t p = mcu_pos[m0];
        if (mcu_rst[im.mcu_off + m0] == 1u || (mcu_rst[im.mcu_off + m0] & 64u)) {   // first MCU of an interval: the RSTn in front of it has been handled (:1644-1680); (a mark > 1: the marker lies INSIDE the MCU, the reader meets it itself)
            const uint32_t u1 = (p + 7) >> 3;                       //   fewer than 8 pad bits, else the parallel path had flagged the image
            rst_before = find_interval(st, nseg, min(u1, total_bytes ? total_bytes - 1 : 0));
            ex_restart_scan_buf(r, raw_of_compacted(im, raw, us_out, us_threads, u1), true); ex_topup(r);
            break;
        }
        const uint32_t sg = find_interval(st, nseg, p >> 3);
        if (sg + 1 >= nseg || st[sg + 1] * 8 >= p + 40) {           // the refill at this MCU top is not cut short by an RSTn
            rst_before = sg;
            ex_restart_scan_buf(r, raw_of_compacted(im, raw, us_out, us_threads, p >> 3), true); ex_topup(r); ex_consume(r, p & 7u);
            break;
        }
    }
    r.ptr_first = im.scan_start;
    int16_t dc_y = 0, dc_cb = 0, dc_cr = 0;
    for (uint32_t mi = m0; mi < m_top; mi++) {
        for (uint32_t c = mi == m0 ? c0 : 0u; c < im.blk_per_mcu; c++) {
            ex_decode_block(r, im.blk_comp[c], im.decode_ac, scratch, dc_y, dc_cb, dc_cr);
            if (r.cur_err) { if (r.warn_bad < r.err_max) r.warn_bad++; r.cur_err = 0; }   // CheckScanErrors :2605
        }
        if (im.rst_en) r.mcus_left--;
    }
    return rst_before;
}

__global__ void __launch_bounds__(256) k_side_maps(const JsImage* __restrict__ imgs, uint32_t img, const JsTableSet* __restrict__ tables, const uint8_t* __restrict__ raw,
                                                   const uint32_t* __restrict__ seg_tab, const int16_t* __restrict__ dccum, const uint8_t* __restrict__ mcu_rst,
                                                   const uint32_t* mcu_pos, const uint32_t* us_out, uint32_t us_threads,
                                                   uint32_t* __restrict__ side, uint32_t* __restrict__ events, uint32_t* __restrict__ anoms, uint32_t dead_blk, uint32_t cut_mcu,
                                                   const uint8_t* __restrict__ img_mask = nullptr, const uint32_t* __restrict__ us_base = nullptr /* a whole batch: image blockIdx.y if
                                                   its mask says so; positions at mcu_pos + rec_off, the inverse map of ALL chunks in us_out (256 words per chunk of us_base) */)
{
    if (img_mask) { img = blockIdx.y; if (!img_mask[img]) return; }
    const JsImage& im = imgs[img];
    if (img_mask) { mcu_pos += im.rec_off; us_out += (size_t)us_base[img] * 256u; us_threads = (us_base[img + 1] - us_base[img]) * 256u; }
    uint32_t* sd = side + im.side_off;
    const uint32_t nmcu = im.mcu_xmax * im.mcu_ymax, nblk = im.blk_xmax * im.blk_ymax, nseg = min(sd[11], im.seg_cap - 1);
    // dead_blk: the block in which the reference's own decode ends (none: ~0) -- behind its MCU the row is not visited, and of every later row only the first MCU
    const uint32_t mstar = dead_blk == 0xFFFFFFFFu ? 0xFFFFFFFFu : dead_blk / im.blk_per_mcu;
    const uint32_t* st = seg_tab + im.seg_off;
    uint32_t* mcu_map = sd + JS_SIDE_MCUMAP;
    int16_t* bdc0 = reinterpret_cast<int16_t*>(mcu_map + nmcu);
    const uint32_t stride = 2 * ((nblk + 1) / 2);
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x, gsz = gridDim.x * 256;
    // ---- MCU file map: position of the first buffered byte and the bit alignment when the MCU loop reaches MCU m (:3229)
    // (the mirror reader's run-time-indexed arrays: one set per WAVE in LDS, its lanes take turns -- as private arrays they, and with them the reader's
    //  whole state, lived in scratch memory: a round trip per access for the one lane in thousands that needs the reader)
    __shared__ uint32_t s_mh[4][2 * 4 * 17 + 12]; __shared__ int16_t s_ms[4][64];
    const uint32_t wv = threadIdx.x >> 6, ln = threadIdx.x & 63u;
    for (uint32_t m0 = blockIdx.x * 256; m0 <= nmcu; m0 += gsz) {
        const uint32_t m = m0 + threadIdx.x; const bool in = m <= nmcu;
        const uint32_t p = (in && m) ? mcu_pos[m] : 0u, ub = p >> 3, a = p & 7u;
        bool empty = false;                                         // the previous interval was consumed to its last bit: the register is empty
        if (in && m && m < nmcu && a == 0) { const uint32_t sg = find_interval(st, nseg, ub); empty = sg >= 1 && st[sg] == ub; }
        const bool mirror = in && (m == nmcu || empty);
        // (a scan that BEGINS with an RSTn: the very first refill meets the marker and loads nothing, :3007-3019 -- at the top of MCU 0 the register is empty and the
        //  position array still holds the zeros DecodeRestartScanBuf left, :4038-4075; found by tools/fuzz_gpu.py seed 202 in round 6, wrong since round 1)
        bool lead_rst = false;
        if (in && m == 0 && im.scan_start + 1u < im.file_len) { const uint8_t* f = raw + im.file_off + im.scan_start; lead_rst = f[0] == 0xFF && f[1] >= 0xD0 && f[1] <= 0xD7; }
        if (in && !mirror) mcu_map[m] = lead_rst ? 0u : (raw_of_compacted(im, raw, us_out, us_threads, ub) << 4) + a;
        for (uint64_t todo = WBALLOT(mirror); todo; todo &= todo - 1) {
            if (ln != (uint32_t)__builtin_ctzll(todo)) continue;
            // what an empty register still shows depends on how its last bytes were loaded, and the end of the scan is where the
            // look-ahead meets EOI / trailing bytes: take both from the mirror reader itself
            ExactReader r;
            const uint32_t before = mirror_to_mcu_top(im, tables, raw, st, nseg, sd[10], mcu_rst, mcu_pos, us_out, us_threads, m, r, s_mh[wv], s_ms[wv],
                                                      m == nmcu ? events : nullptr, nullptr, m == nmcu ? mcu_pos[nmcu + 1u] : 0u);
            if (m < nmcu) mcu_map[m] = (r.pos0 << 4) + r.align;
            else {                                                  // status words after the last MCU
                sd[0] = r.scan_bad; sd[1] = r.scan_end; sd[2] = before + r.rst_count; sd[3] = nmcu * im.samp_h[1] * im.samp_v[1] * 64u;
                sd[4] = r.pos0; sd[5] = r.align; sd[6] = r.warn_bad; sd[7] = im.scan_start;
                if (anoms) anoms[1] = r.warn_marker;              // (an image with overflow records: its counter is put together on the host)
            }
        }
    }
    // ---- block-DC maps: the cumulative DC of the block that wrote the cell last (MCU raster order, :3524-3608)
    const int16_t* dc = dccum + im.coef_off;
    for (uint32_t q = gid; q < im.ncomp * nblk; q += gsz) {
        const uint32_t comp = q / nblk + 1, cell = q % nblk, bx = cell % im.blk_xmax, by = cell / im.blk_xmax;
        const uint32_t eh = im.expand_h[comp], ev = im.expand_v[comp];
        int best = -1; uint32_t best_blk = 0;
        for (uint32_t c = 0; c < im.blk_per_mcu; c++) {
            if (im.blk_comp[c] != comp || !eh || !ev) continue;
            const uint32_t ch = im.blk_ch[c], cv = im.blk_cv[c];
            if (bx < ch || by < cv || (bx - ch) % eh || (by - cv) % ev) continue;
            const uint32_t mx = (bx - ch) / eh, my = (by - cv) / ev;
            if (mx >= im.mcu_xmax || my >= im.mcu_ymax) continue;
            const int mi = (int)(my * im.mcu_xmax + mx);
            if (mstar != 0xFFFFFFFFu && (uint32_t)mi > mstar && (mx != 0 || my == mstar / im.mcu_xmax)) continue;     // an MCU the reference never reaches writes nothing
            if ((uint32_t)mi >= cut_mcu) continue;                                                                     // (MCUs the run-on lane of k_side_chunks writes itself, in the reference's order)
            if (mi > best) { best = mi; best_blk = (uint32_t)mi * im.blk_per_mcu + c; }
        }
        // (a restart handled while block j of the MCU was in progress -- its mark, j + 1 > 1 -- clears the reference's per-MCU array of sums, :3524-3608 /
        //  DecodeRestartDcState: the blocks of that MCU in front of j show 0 in the maps, whatever their sums were)
        if (best >= 0) { const uint32_t mark = mcu_rst[im.mcu_off + (uint32_t)best] & 63u; bdc0[(comp - 1) * stride + cell] = (mark > 1u && best_blk % im.blk_per_mcu + 1u < mark) ? (int16_t)0 : dc[best_blk]; }
    }
}

// =====================================================================================
//  Side outputs and messages of a DAMAGED image, in parallel (round 6).  The reference's log of such a file -- "Bad huffman code",
//  "Can't find huffman bitstring", "Bad marker", "nNumCoeffs>64", "Bad scan data in MCU(..)", restart bookkeeping (:1167-1187, :1644-1757,
//  :2605-2660) -- quotes the reader's position array and its register (ScanBuffConsume :921-955), which only the exact reader reproduces; run
//  over a whole 1080p image on one lane that reader takes 1.2 s.  But the walks of the parallel path follow a damaged stream the reference's
//  way, so the bit position of EVERY MCU top is known (side walk, mcu_pos) -- and at an MCU top the reader's state is a function of the
//  bit position and of a few bytes of history (mirror_to_mcu_top).  The image is therefore cut into chunks of a few MCUs; one lane per
//  chunk (a wave of its own: the reader is one long chain of branches) brings an exact reader to its chunk's first MCU top and runs the
//  reference's MCU loop over the chunk in side-only mode: MCU file map, code-length histogram, the events with the exact positions.
//  What a lane cannot know locally is put together on the host (js_side_only): the warning counter the messages share (every lane counts
//  and gates from zero: the first nErrMaxDecodeScan counted events of the image are among the first nErrMaxDecodeScan of their lanes),
//  scan_bad (cleared by every restart, set by the errors), the number of RSTn read, the restart countdown at the chunk's start (handed in).
//  A decode that ENDS (value bits past an interval end: scan_end + scan_bad, :3623-3625) is followed by the lane that meets the end, alone,
//  through the rest of the image -- one MCU per row; what later chunks' lanes produced is dropped by the host.
//  Record of a chunk (u32): 0 died at MCU (~0: no), 1 bits (1: a restart was handled, 2: scan_bad, 4: scan_end), 2 pos0, 3 align, 4 RSTn read,
//  5 num_pixels, 6 counted events, 8..143 histogram, 144 events logged, 145.. the events (JS_EV_WORDS each).
// =====================================================================================
#define SC_WAVES 4
#define SC_HDR 145
__global__ void __launch_bounds__(64 * SC_WAVES) k_side_chunks(const JsImage* __restrict__ imgs, uint32_t img, const JsTableSet* __restrict__ tables, const uint8_t* __restrict__ raw,
                                                             const uint32_t* __restrict__ seg_tab, const uint8_t* __restrict__ mcu_rst, const uint32_t* __restrict__ mcu_pos,
                                                             const uint32_t* __restrict__ us_out, uint32_t us_threads, uint32_t* __restrict__ side, const int16_t* __restrict__ dccum,
                                                             uint32_t ch_mcus, uint32_t nchunks, uint32_t ev_cap, const uint32_t* __restrict__ mcus_left0,
                                                             uint32_t* __restrict__ recs, uint32_t* __restrict__ map_own, unsigned long long* __restrict__ map_beyond,
                                                             uint32_t run_on_mcu, uint32_t* __restrict__ fill_desc)
{
    __shared__ uint32_t s_fast[6 * (1 << JS_FAST_BITS)]; __shared__ uint16_t s_q[3 * 64]; __shared__ uint8_t s_zz[64];
    __shared__ uint32_t s_h[SC_WAVES][2 * 4 * 17 + 12]; __shared__ int16_t s_scr[SC_WAVES][64];
    __shared__ uint32_t s_h2[SC_WAVES][2 * 4 * 17];               // (the histogram at the top of the template MCU of a run of zero bytes, below)
    const JsImage& im = imgs[img];
    const JsTableSet& tset = tables[im.tableset];
    {
        const uint32_t* src = &tset.fast[0][0];
        for (uint32_t i = threadIdx.x; i < 6 * (1u << JS_FAST_BITS); i += blockDim.x) s_fast[i] = src[i];
        for (uint32_t i = threadIdx.x; i < 3 * 64; i += blockDim.x) s_q[i] = (&tset.qzz[0][0])[i];
        if (threadIdx.x < 64) s_zz[threadIdx.x] = c_zigzag[threadIdx.x];
    }
    __syncthreads();
    const uint32_t wv = threadIdx.x >> 6, chunk = blockIdx.x * SC_WAVES + wv;
    if ((threadIdx.x & 63u) != 0u || chunk >= nchunks) return;
    const uint32_t* sd = side + im.side_off;
    const uint32_t nmcu = im.mcu_xmax * im.mcu_ymax, nseg = min(sd[11], im.seg_cap - 1), xmax = im.mcu_xmax;
    const uint32_t* st = seg_tab + im.seg_off;
    const uint32_t m_a = chunk * ch_mcus, m_b = min(m_a + ch_mcus, nmcu);
    uint32_t* rec = recs + (size_t)chunk * (SC_HDR + (size_t)ev_cap * JS_EV_WORDS);
    // run_on_mcu (none: ~0): the MCU from which the walks do NOT vouch for the stream any more (the image's pixels behind it came from the mirror's
    // tail take-over).  The lane whose chunk holds it goes on alone to the end of the image; the lanes behind it have nothing to start from.
    if (m_a > run_on_mcu) { rec[0] = 0xFFFFFFFFu; rec[SC_HDR - 1] = 0; return; }
    const bool run_on = run_on_mcu != 0xFFFFFFFFu && run_on_mcu < m_b;
    uint32_t* histo = s_h[wv];
    ExactReader tabs; tabs.fast = s_fast; tabs.q = s_q; tabs.zz = s_zz;
    ExactReader r;
    rec[SC_HDR - 1] = 0;
    if (m_a == 0) {
        // the first chunk starts like the reference itself (its first refill may already meet markers: logged)
        uint32_t* meta12 = histo + 2 * 4 * 17;
        r.file = raw + im.file_off; r.flen = im.file_len; r.ts = &tset; r.fast = s_fast; r.meta = meta12; r.q = s_q; r.zz = s_zz; r.win_at = 0xFFFFFFFFu; r.win = 0;
        for (int i = 0; i < 6; i++) { meta12[i] = tset.size[i]; meta12[6 + i] = tset.dest_id[i]; }
        r.rst_interval = im.rst_interval; r.precision = im.precision;
        r.rst_count = 0; r.rst_last = 0; r.rst_expect = 0; r.rst_handled = 0;
        for (uint32_t i = 0; i < 2 * 4 * 17; i++) histo[i] = 0;
        r.histo = histo; r.err_max = im.err_max; r.warn_bad = 0; r.warn_marker = 0; r.ev = rec + SC_HDR - 1; r.ev_cap = ev_cap; r.ev_only = 0; r.ev_end = 0;
        ex_restart_scan_buf(r, im.scan_start, false);
        ex_topup(r);
    } else {
        const uint32_t before = mirror_to_mcu_top(im, tables, raw, st, nseg, sd[10], mcu_rst, mcu_pos, us_out, us_threads, m_a, r, histo, s_scr[wv], nullptr, &tabs);
        // what the walk to the chunk's first MCU top cannot have established: the number the next RSTn is expected to carry, when it met none itself --
        // one more than the last marker in front of where it started
        if (r.rst_count == 0 && before > 0 && before < nseg) {
            const uint32_t rp = raw_of_compacted(im, raw, us_out, us_threads, st[before]);
            const uint32_t b1 = rp >= 1u ? ex_byte(r, rp - 1u) : 0u;
            if (b1 >= 0xD0u && b1 <= 0xD7u) { r.rst_last = b1 - 0xD0u; r.rst_expect = (r.rst_last + 1u) & 7u; }
        }
        for (uint32_t i = 0; i < 2 * 4 * 17; i++) histo[i] = 0;
        r.err_max = im.err_max; r.warn_bad = 0; r.warn_marker = 0; r.ev = rec + SC_HDR - 1; r.ev_cap = ev_cap; r.ev_only = 0; r.ev_end = 0;
        r.scan_bad = 0; r.cur_err = 0;
    }
    r.mcus_left = mcus_left0[chunk];
    const uint32_t rst_count0 = m_a == 0 ? 0u : r.rst_count, rst_handled0 = r.rst_handled;      // (the first chunk's first refill is the reference's own, :3019)
    uint32_t num_pixels = 0, died_at = 0xFFFFFFFFu;
    int16_t dc_y = 0, dc_cb = 0, dc_cr = 0;
    // The run-on lane also keeps the block-DC maps (:3524-3608) from its chunk on: behind run_on_mcu the restarts were handled by the mirror's tail take-over,
    // the marks of the walks say nothing about them, and where the reference's decode ends there only this lane finds out -- k_side_maps has left those MCUs
    // out (its cut), this lane writes its MCUs in the reference's order with the reference's per-MCU array of sums.  Predictors at its first MCU top as in
    // the tail take-over: the sums the parallel path left for the MCU before, cleared where a restart was followed behind a component's last block.
    uint32_t* mcu_map_sd = side + im.side_off + JS_SIDE_MCUMAP;
    const uint32_t nblk = im.blk_xmax * im.blk_ymax, bstride = 2 * ((nblk + 1) / 2);
    int16_t* bdc0 = reinterpret_cast<int16_t*>(mcu_map_sd + nmcu);
    __shared__ int16_t s_css[SC_WAVES][3][16];
    int16_t (*css)[16] = s_css[wv];
    if (run_on) {
        for (int cc = 0; cc < 3; cc++) for (int q = 0; q < 16; q++) css[cc][q] = 0;
        if (m_a) {
            const uint32_t nb = im.blk_per_mcu, n1 = im.samp_h[1] * im.samp_v[1], n2 = im.ncomp == 3 ? n1 + im.samp_h[2] * im.samp_v[2] : nb;
            const int16_t* dprev = dccum + im.coef_off + (size_t)(m_a - 1) * nb;
            dc_y = dprev[n1 - 1]; if (im.ncomp == 3) { dc_cb = dprev[n2 - 1]; dc_cr = dprev[nb - 1]; }
            const uint32_t rj = mcu_rst[im.mcu_off + m_a - 1u] & 63u;
            if (rj) { if (n1 < rj) dc_y = 0; if (im.ncomp == 3 && n2 < rj) dc_cb = 0; }
        }
    }
    auto one_mcu = [&](uint32_t mi) {                             // the body of DecodeScanImg's MCU loop (:3164-3625), side outputs only
        if (im.rst_en && r.mcus_left == 0 && !r.restart_read) ex_event(r, JS_EV_RST_NOT_DETECTED, r.pos0, r.align);   // :3180-3200
        // PackFileOffset :5104.  Behind its own chunk (a lane that met the end of the decode) the entry is kept per MCU for the EARLIEST chunk that got
        // there: every later lane meets "its" end too -- in a state the reference never had -- and only the first one's is the reference's
        const uint32_t pk = (r.pos0 << 4) + r.align;
        if (mi < m_b) map_own[mi] = pk; else atomicMin(&map_beyond[mi], ((unsigned long long)chunk << 32) | pk);
        const uint32_t mx = mi % xmax, my = mi / xmax;
        for (uint32_t c = 0; c < im.blk_per_mcu; c++) {
            const uint32_t comp = im.blk_comp[c];
            const uint32_t rst_before = r.rst_handled;
            const int16_t d0 = ex_decode_block(r, comp, im.decode_ac, s_scr[wv], dc_y, dc_cb, dc_cr);
            if (run_on && r.rst_handled != rst_before) for (int cc = 0; cc < 3; cc++) for (int q = 0; q < 16; q++) css[cc][q] = 0;
            if (r.cur_err) {                                      // CheckScanErrors :2605
                if (r.warn_bad < r.err_max) { ex_event(r, JS_EV_BAD_SCAN_MCU, mx | (my << 16), comp | (im.blk_ch[c] << 8) | (im.blk_cv[c] << 16), r.pos0, r.align); r.warn_bad++; }
                r.cur_err = 0; }
            if (run_on) { int16_t* acc = comp == 1 ? &dc_y : comp == 2 ? &dc_cb : &dc_cr; *acc = (int16_t)(*acc + d0); css[comp - 1][im.blk_cv[c] * 4 + im.blk_ch[c]] = *acc; }
            if (comp == 1) num_pixels += 64;
        }
        if (run_on) {                                             // per-block cumulative DC maps :3524-3608 (sequential overwrite order preserved)
            const uint32_t lin = (my * im.expand_v[1]) * im.blk_xmax + mx * im.expand_h[1];
            for (uint32_t cv = 0; cv < im.samp_v[1]; cv++) for (uint32_t ch = 0; ch < im.samp_h[1]; ch++) { const uint32_t bi = lin + cv * im.blk_xmax + ch; if (bi < nblk) bdc0[bi] = css[0][cv * 4 + ch]; }
            if (im.ncomp == 3) for (uint32_t comp = 2; comp <= 3; comp++)
                for (uint32_t cv = 0; cv < im.samp_v[comp]; cv++) for (uint32_t ch = 0; ch < im.samp_h[comp]; ch++) {
                    const uint32_t bi = (my * im.expand_v[comp] + cv) * im.blk_xmax + (mx * im.expand_h[comp] + ch);
                    if (bi < nblk) bdc0[(comp - 1) * bstride + bi] = css[comp - 1][cv * 4 + ch]; }
        }
        if (im.rst_en) r.mcus_left--;
    };
    for (uint32_t mi = m_a; mi < m_b; mi++) {
        one_mcu(mi);
        if (r.scan_end && r.scan_bad) { died_at = mi; break; }     // :3623-3625: the row stops here ...
    }
    if (died_at != 0xFFFFFFFFu)                                   // ... and of every later row the first MCU is all that is reached
        for (uint32_t my = died_at / xmax + 1; my < im.mcu_ymax; my++) one_mcu(my * xmax);
    else if (run_on) {
        for (uint32_t mi = m_b; mi < nmcu; mi++) {
            // Out of file (a truncated picture): every byte the reader will ever load is the zero CwindowBuf::Buf returns past the end (WindowBuf.cpp:639),
            // its register holds zero bits, nothing is pending -- every further MCU consumes the same bits, logs the same nothing and counts the same code
            // lengths.  ONE such MCU is decoded; when it left no message its effect is applied to all the MCUs behind it in closed form (positions are
            // file offsets that go on past the end, one per byte) instead of 0.15 ms of sequential decode each.
            const bool zero_state = mi + 1 < nmcu && r.buff == 0u && r.ptr >= r.flen && !r.restart_read && !r.scan_end && r.latch == SB_OK &&
                                    r.err0 == SB_OK && r.err1 == SB_OK && r.err2 == SB_OK && r.err3 == SB_OK && r.num >= 1u;   // (num >= 1: pos0 is the offset of a byte that was loaded)
            uint32_t ev0 = 0, px0 = 0, p0 = 0; int zs_dc[3] = { 0, 0, 0 };
            if (zero_state) { ev0 = r.ev[0]; px0 = num_pixels; p0 = r.pos0 * 8u + r.align; zs_dc[0] = dc_y; zs_dc[1] = dc_cb; zs_dc[2] = dc_cr; for (uint32_t i = 0; i < 2 * 4 * 17; i++) s_h2[wv][i] = histo[i]; }
            one_mcu(mi);
            if (r.scan_end && r.scan_bad) { died_at = mi; for (uint32_t my = mi / xmax + 1; my < im.mcu_ymax; my++) one_mcu(my * xmax); break; }
            const uint32_t p1 = r.pos0 * 8u + r.align, rest = nmcu - 1u - mi;
            if (zero_state && r.ev[0] == ev0 && r.buff == 0u && !r.restart_read && r.latch == SB_OK && r.num >= 1u && p1 > p0) {
                const uint32_t bits = p1 - p0;
                // (the one message that can fall into the run: the restart countdown reaches zero at the top of one of these MCUs, :3180-3200 -- once, it wraps)
                if (im.rst_en && r.mcus_left < rest) { const uint32_t pz = p1 + r.mcus_left * bits; ex_event(r, JS_EV_RST_NOT_DETECTED, pz >> 3, pz & 7u); }
                // (the per-MCU part of it -- map entries and block-DC cells of `rest` MCUs -- is k_side_fill's, on the whole chip: one lane took 4-8 ms for
                //  the 4000 MCUs behind a picture truncated at half its scan)
                fill_desc[1] = mi; fill_desc[2] = p1; fill_desc[3] = bits; fill_desc[4] = chunk;
                fill_desc[5] = (uint32_t)((int)dc_y - zs_dc[0]); fill_desc[6] = (uint32_t)((int)dc_cb - zs_dc[1]); fill_desc[7] = (uint32_t)((int)dc_cr - zs_dc[2]);
                for (int cc = 0; cc < 3; cc++) for (int q = 0; q < 16; q++) fill_desc[8 + cc * 16 + q] = (uint32_t)(int)css[cc][q];
                __threadfence();
                fill_desc[0] = 1u;
                for (uint32_t i = 0; i < 2 * 4 * 17; i++) histo[i] += (histo[i] - s_h2[wv][i]) * rest;
                num_pixels += (num_pixels - px0) * rest;
                const uint32_t pe = p1 + rest * bits;
                r.pos0 = pe >> 3; r.align = pe & 7u;
                if (im.rst_en) r.mcus_left -= rest;
                break;
            }
        }
    }
    rec[0] = died_at;
    rec[1] = (r.rst_handled != rst_handled0 ? 1u : 0u) | (r.scan_bad ? 2u : 0u) | (r.scan_end ? 4u : 0u);
    rec[2] = r.pos0; rec[3] = r.align; rec[4] = r.rst_count - rst_count0; rec[5] = num_pixels; rec[6] = r.warn_bad; rec[7] = r.warn_marker;
    for (uint32_t i = 0; i < 2 * 4 * 17; i++) rec[8 + i] = histo[i];
}
// The closed form of a run of zero bytes (k_side_chunks, run-on lane), applied: MCU file map entries and block-DC cells of every MCU behind the template MCU.
// desc: 0 valid, 1 template MCU, 2 bit position behind it, 3 bits per MCU, 4 the lane's chunk, 5..7 step of the three predictors per MCU, 8.. the per-MCU array
// of sums behind the template.  A cell of the block-DC maps takes the value of the LAST MCU (raster order) that writes it (:3524-3608; neighbouring MCUs' cells overlap).
__global__ void __launch_bounds__(256) k_side_fill(const JsImage* __restrict__ imgs, uint32_t img, uint32_t* __restrict__ side, unsigned long long* __restrict__ map_beyond,
                                                   const uint32_t* __restrict__ desc)
{
    if (!desc[0]) return;
    const JsImage& im = imgs[img];
    const uint32_t nmcu = im.mcu_xmax * im.mcu_ymax, nblk = im.blk_xmax * im.blk_ymax, xmax = im.mcu_xmax, mt = desc[1], p1 = desc[2], bits = desc[3], chunk = desc[4];
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x, gsz = gridDim.x * 256;
    for (uint32_t m = mt + 1 + gid; m < nmcu; m += gsz) {
        const uint32_t pj = p1 + (m - mt - 1u) * bits;
        atomicMin(&map_beyond[m], ((unsigned long long)chunk << 32) | (((pj >> 3) << 4) + (pj & 7u)));
    }
    uint32_t* mcu_map = side + im.side_off + JS_SIDE_MCUMAP;
    int16_t* bdc0 = reinterpret_cast<int16_t*>(mcu_map + nmcu);
    const uint32_t stride = 2 * ((nblk + 1) / 2);
    for (uint32_t q = gid; q < im.ncomp * nblk; q += gsz) {
        const uint32_t comp = q / nblk + 1, cell = q % nblk, bx = cell % im.blk_xmax, by = cell / im.blk_xmax;
        const uint32_t eh = im.expand_h[comp], ev = im.expand_v[comp];
        int best = -1; uint32_t slot = 0;
        for (uint32_t cv = 0; cv < im.samp_v[comp]; cv++) for (uint32_t ch = 0; ch < im.samp_h[comp]; ch++) {
            // MCU (mx, my) writes cell (my * ev' + cv, mx * eh' + ch) of its component: luma through its "corner" (expand bits), chroma through (mcu * expand + index)
            if (!eh || !ev || bx < ch || by < cv || (bx - ch) % eh || (by - cv) % ev) continue;
            const uint32_t mx = (bx - ch) / eh, my = (by - cv) / ev;
            if (mx >= xmax || my >= im.mcu_ymax) continue;
            const int mi = (int)(my * xmax + mx);
            if ((uint32_t)mi > mt && mi > best) { best = mi; slot = cv * 4 + ch; }
        }
        if (best >= 0) bdc0[(comp - 1) * stride + cell] = (int16_t)((int)desc[8 + (comp - 1) * 16 + slot] + (best - (int)mt) * (int)desc[4 + comp]);
    }
}
void js_launch_side_chunks(hipStream_t st, const JsImage* imgs, uint32_t img, const JsTableSet* tables, const uint8_t* raw, const uint32_t* seg_tab, const uint8_t* mcu_rst,
                           const uint32_t* mcu_pos, const uint32_t* us_out, uint32_t us_threads, uint32_t* side, const int16_t* dccum, uint32_t ch_mcus, uint32_t nchunks, uint32_t ev_cap,
                           const uint32_t* mcus_left0, uint32_t* recs, uint32_t* map_own, unsigned long long* map_beyond, uint32_t run_on_mcu, uint32_t* fill_desc)
{
    if (!nchunks) return;
    hipLaunchKernelGGL(k_side_chunks, dim3((nchunks + SC_WAVES - 1) / SC_WAVES), dim3(64 * SC_WAVES), 0, st, imgs, img, tables, raw, seg_tab, mcu_rst, mcu_pos, us_out, us_threads, side, dccum,
                       ch_mcus, nchunks, ev_cap, mcus_left0, recs, map_own, map_beyond, run_on_mcu, fill_desc);
    if (run_on_mcu != 0xFFFFFFFFu) hipLaunchKernelGGL(k_side_fill, dim3(128), dim3(256), 0, st, imgs, img, side, map_beyond, fill_desc);
}
// The side passes of MANY images of a decoded batch in four launches (js_side_all): the outputs cleared, the inverse byte map of every chunk, the side walk over the
// whole batch (images outside the mask leave at once), the maps with one grid row per image.  pos_all: rec_off-indexed MCU-top positions (zeroed here by the
// caller), us_all: 256 words per chunk of the batch.
__global__ void __launch_bounds__(256) k_side_clear_all(const JsImage* __restrict__ imgs, const uint8_t* __restrict__ img_mask, uint32_t* __restrict__ side, uint32_t* __restrict__ events)
{
    const uint32_t img = blockIdx.y;
    if (!img_mask[img]) return;
    const JsImage& im = imgs[img];
    uint32_t* sd = side + im.side_off;
    const uint32_t words = js_side_words(im.mcu_xmax * im.mcu_ymax, im.blk_xmax * im.blk_ymax);
    for (uint32_t q = blockIdx.x * 256 + threadIdx.x; q < words; q += gridDim.x * 256) if (q < 8u || q >= JS_SIDE_HISTO) sd[q] = 0u;
    if (events && im.ev_cap && blockIdx.x == 0 && threadIdx.x == 0) events[im.ev_off] = 0u;
}
void js_launch_side_pass_all(hipStream_t st, int wl, uint32_t tab_rows, uint32_t tab_lut2, const JsImage* imgs, const uint32_t* us_base, const uint32_t* sy_base, uint32_t nimg,
                             uint32_t us_wgs, uint32_t sy_wgs, const JsTableSet* tables, const uint8_t* raw, const uint32_t* chunk_keep, const uint32_t* chunk_rst, const uint8_t* ustr,
                             uint32_t* seg_tab, uint32_t* side, uint32_t* sub, uint64_t nsub, const int16_t* dccum, uint8_t* mcu_rst, uint32_t* pos_all, uint32_t* us_all,
                             uint32_t* events, const uint8_t* img_mask)
{
    if (!us_wgs || !sy_wgs || !nimg) return;
    hipLaunchKernelGGL(k_side_clear_all, dim3(16, nimg), dim3(256), 0, st, imgs, img_mask, side, events);
    hipLaunchKernelGGL(k_unstuff_write<false>, dim3(us_wgs), dim3(US_THREADS), 0, st, imgs, us_base, nimg, raw, const_cast<uint32_t*>(chunk_keep), const_cast<uint32_t*>(chunk_rst), (uint8_t*)nullptr, seg_tab, 0u, us_all,
                       (unsigned long long*)nullptr, 0u, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u);
#define JS_ALL_WALK(W) hipLaunchKernelGGL((k_write<W, true>), dim3(sy_wgs), dim3(SY_THREADS), wtabs_bytes(tab_rows, tab_lut2), st, imgs, sy_base, nimg, tables, ustr, seg_tab, side, \
                       sub_arrays(sub, nsub), (int16_t*)nullptr, (int16_t*)nullptr, mcu_rst, (uint32_t*)nullptr, tab_rows, tab_lut2, 0u, pos_all, img_mask)
    if (wl == 4) JS_ALL_WALK(4); else if (wl == 6) JS_ALL_WALK(6); else if (wl == 8) JS_ALL_WALK(8); else if (wl == 7) JS_ALL_WALK(7); else JS_ALL_WALK(5);
#undef JS_ALL_WALK
    hipLaunchKernelGGL(k_side_maps, dim3(16, nimg), dim3(256), 0, st, imgs, 0u, tables, raw, seg_tab, dccum, mcu_rst, pos_all, us_all, 0u, side, events, (uint32_t*)nullptr, 0xFFFFFFFFu, 0xFFFFFFFFu,
                       img_mask, us_base);
}
void js_launch_side_pass(hipStream_t st, int wl, uint32_t tab_rows, uint32_t tab_lut2, const JsImage* imgs, const uint32_t* us_base, const uint32_t* sy_base, uint32_t nimg,
                         uint32_t img, uint32_t us_wg0, uint32_t us_wgs, uint32_t sy_wg0, uint32_t sy_wgs, const JsTableSet* tables, const uint8_t* raw,
                         const uint32_t* chunk_keep, const uint32_t* chunk_rst, const uint8_t* ustr, uint32_t* seg_tab, uint32_t* side, uint32_t* sub, uint64_t nsub,
                         const int16_t* dccum, uint8_t* mcu_rst, uint32_t* mcu_pos, uint32_t* us_out, uint32_t* events, uint32_t* anoms, uint32_t dead_blk, uint32_t cut_mcu, bool walked)
{
    if (!us_wgs || !sy_wgs) return;
    hipLaunchKernelGGL(k_unstuff_write<false>, dim3(us_wgs), dim3(US_THREADS), 0, st, imgs, us_base, nimg, raw, const_cast<uint32_t*>(chunk_keep), const_cast<uint32_t*>(chunk_rst), (uint8_t*)nullptr, seg_tab, us_wg0, us_out,
                       (unsigned long long*)nullptr, 0u, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u);
    if (walked) {}                                                 // (the write pass of the decode recorded positions and histogram itself: k_write2<., ., true>)
    else if (wl == 4) hipLaunchKernelGGL((k_write<4, true>), dim3(sy_wgs), dim3(SY_THREADS), wtabs_bytes(tab_rows, tab_lut2), st, imgs, sy_base, nimg, tables, ustr, seg_tab, side,
                       sub_arrays(sub, nsub), (int16_t*)nullptr, (int16_t*)nullptr, mcu_rst, anoms, tab_rows, tab_lut2, sy_wg0, mcu_pos);
    else if (wl == 6) hipLaunchKernelGGL((k_write<6, true>), dim3(sy_wgs), dim3(SY_THREADS), wtabs_bytes(tab_rows, tab_lut2), st, imgs, sy_base, nimg, tables, ustr, seg_tab, side,
                       sub_arrays(sub, nsub), (int16_t*)nullptr, (int16_t*)nullptr, mcu_rst, anoms, tab_rows, tab_lut2, sy_wg0, mcu_pos);
    else if (wl == 8) hipLaunchKernelGGL((k_write<8, true>), dim3(sy_wgs), dim3(SY_THREADS), wtabs_bytes(tab_rows, tab_lut2), st, imgs, sy_base, nimg, tables, ustr, seg_tab, side,
                       sub_arrays(sub, nsub), (int16_t*)nullptr, (int16_t*)nullptr, mcu_rst, anoms, tab_rows, tab_lut2, sy_wg0, mcu_pos);
    else if (wl == 7) hipLaunchKernelGGL((k_write<7, true>), dim3(sy_wgs), dim3(SY_THREADS), wtabs_bytes(tab_rows, tab_lut2), st, imgs, sy_base, nimg, tables, ustr, seg_tab, side,
                       sub_arrays(sub, nsub), (int16_t*)nullptr, (int16_t*)nullptr, mcu_rst, anoms, tab_rows, tab_lut2, sy_wg0, mcu_pos);
    else hipLaunchKernelGGL((k_write<5, true>), dim3(sy_wgs), dim3(SY_THREADS), wtabs_bytes(tab_rows, tab_lut2), st, imgs, sy_base, nimg, tables, ustr, seg_tab, side,
                       sub_arrays(sub, nsub), (int16_t*)nullptr, (int16_t*)nullptr, mcu_rst, anoms, tab_rows, tab_lut2, sy_wg0, mcu_pos);
    hipLaunchKernelGGL(k_side_maps, dim3(64), dim3(256), 0, st, imgs, img, tables, raw, seg_tab, dccum, mcu_rst, mcu_pos, us_out, us_wgs * US_THREADS, side, events, anoms, dead_blk, cut_mcu);
}
// Tail take-over for image `img` of a decoded batch (see ExactTail): the inverse byte map and the MCU bit positions through the first two
// kernels of the side pass, then the mirror reader from the MCU that holds the first block the parallel path could not vouch for.
void js_launch_tail_pass(hipStream_t st, int wl, uint32_t tab_rows, uint32_t tab_lut2, const JsImage* imgs, const uint32_t* us_base, const uint32_t* sy_base, uint32_t nimg,
                         uint32_t us_wg0, uint32_t us_wgs, uint32_t sy_wg0, uint32_t sy_wgs, const JsTableSet* tables, const uint8_t* raw,
                         const uint32_t* chunk_keep, const uint32_t* chunk_rst, const uint8_t* ustr, uint32_t* seg_tab, uint32_t* side, uint32_t* sub, uint64_t nsub,
                         int16_t* coef, int16_t* dccum, uint8_t* mcu_rst, uint32_t* mcu_pos, uint32_t* us_out, const uint32_t* flags, const uint32_t* sel1)
{
    if (!us_wgs || !sy_wgs) return;
    hipLaunchKernelGGL(k_unstuff_write<false>, dim3(us_wgs), dim3(US_THREADS), 0, st, imgs, us_base, nimg, raw, const_cast<uint32_t*>(chunk_keep), const_cast<uint32_t*>(chunk_rst), (uint8_t*)nullptr, seg_tab, us_wg0, us_out,
                       (unsigned long long*)nullptr, 0u, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u);
#define JS_TAIL_WALK(W) hipLaunchKernelGGL((k_write<W, true>), dim3(sy_wgs), dim3(SY_THREADS), wtabs_bytes(tab_rows, tab_lut2), st, imgs, sy_base, nimg, tables, ustr, seg_tab, side, \
                       sub_arrays(sub, nsub), (int16_t*)nullptr, (int16_t*)nullptr, mcu_rst, (uint32_t*)nullptr, tab_rows, tab_lut2, sy_wg0, mcu_pos)
    if (wl == 4) JS_TAIL_WALK(4); else if (wl == 6) JS_TAIL_WALK(6); else if (wl == 8) JS_TAIL_WALK(8); else if (wl == 7) JS_TAIL_WALK(7); else JS_TAIL_WALK(5);
#undef JS_TAIL_WALK
    ExactTail t; t.flags = flags; t.seg_tab = seg_tab; t.mcu_rst = mcu_rst; t.mcu_pos = mcu_pos; t.us_out = us_out; t.us_threads = us_wgs * US_THREADS;
    hipLaunchKernelGGL(k_entropy_exact, dim3(1), dim3(64), 0, st, imgs, sel1, 1u, tables, raw, coef, dccum, side, 0, (uint32_t*)nullptr, t);
}
// An image whose decode ENDS in block `bstar` (ANOM_KEY kinds AK_DEAD_*): the value bits of a symbol ran past the end of a restart interval, the
// reference's register over-reads (:1229-1282), scan_end and scan_bad stay set, and from then on every block fails at its first read without
// consuming anything: block bstar keeps its DC difference if that had been decoded (AK_DEAD_AC) and nothing else, the blocks behind it in its MCU
// are empty with the predictors standing, the rest of that MCU row is never decoded (:3623-3625: the row loop stops, the cleared arrays stay), and
// of every later row only the first MCU is "decoded" -- empty blocks, predictors standing.  No restart is handled any more.
//  k_dead_fill (before the DC scan is repeated for the image): rows from bstar on emptied, DC differences of all blocks back in the dccum
//  array (they survive in slot 0 of the rows), restart marks behind bstar removed -- and the one ON bstar when it stands for a marker met INSIDE the
//  block (mark_reset) that the block's owner did not see before the decode ended (ANOM_KEY: the walk went on and met the next marker in the same block);  k_dead_rows (after it): cumulative DC of the MCUs the
//  reference never reaches back to the cleared arrays' zero.
__global__ void __launch_bounds__(256) k_dead_fill(const JsImage* __restrict__ imgs, uint32_t img, uint32_t bstar, uint32_t kind,
                                                   int16_t* __restrict__ coef, int16_t* __restrict__ dccum, uint8_t* __restrict__ mcu_rst)
{
    const JsImage& im = imgs[img];
    const uint32_t nb = im.blk_per_mcu, total = im.total_blocks, nmcu = im.mcu_xmax * im.mcu_ymax, mstar = bstar / nb;
    int16_t* cb = coef + im.coef_off * 64; int16_t* d = dccum + im.coef_off; uint8_t* rf = mcu_rst + im.mcu_off;
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x, gsz = gridDim.x * 256;
    const bool keep_dc = ((kind - AK_DEAD) & 1u) != 0u, own_mark = ((kind - AK_DEAD) & 2u) != 0u;
    for (uint32_t b = gid; b < total; b += gsz) d[b] = (b < bstar || (b == bstar && keep_dc)) ? cb[(size_t)b * 64] : (int16_t)0;
    uint4* rows = reinterpret_cast<uint4*>(cb + (size_t)bstar * 64);
    for (size_t q = gid; q < (size_t)(total - bstar) * 8; q += gsz) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (q == 0 && keep_dc) v.x = rows[0].x & 0xFFFFu;        // (slot 0 of row bstar: its DC difference)
        rows[q] = v;
    }
    for (uint32_t m = mstar + gid; m < nmcu; m += gsz) {
        const uint32_t mark = rf[m], mblk = mstar * nb + (mark & 63u) - 1u;        // the block in front of which the mark clears the predictors
        if (mark && m > mstar) rf[m] = 0;
        else if (mark && (mblk > bstar || (mblk == bstar && (mark & 128u) && !own_mark))) rf[m] = (mark & 64u) ? 1 : 0;      // (a second mark on the MCU's boundary stays: it lies in front of b*)
    }
}
__global__ void __launch_bounds__(256) k_dead_rows(const JsImage* __restrict__ imgs, uint32_t img, uint32_t bstar, int16_t* __restrict__ dccum)
{
    const JsImage& im = imgs[img];
    const uint32_t nb = im.blk_per_mcu, nmcu = im.mcu_xmax * im.mcu_ymax, mstar = bstar / nb, xmax = im.mcu_xmax;
    int16_t* d = dccum + im.coef_off;
    for (uint32_t q = (mstar + 1) * nb + blockIdx.x * 256 + threadIdx.x; q < nmcu * nb; q += gridDim.x * 256) {
        const uint32_t m = q / nb;
        if (m / xmax == mstar / xmax || m % xmax != 0) d[q] = 0;
    }
}
void js_launch_dead_fill(hipStream_t st, const JsImage* imgs, uint32_t img, uint32_t bstar, uint32_t kind, const JsTableSet* tables, int16_t* coef, int16_t* dccum, uint8_t* mcu_rst)
{
    hipLaunchKernelGGL(k_dead_fill, dim3(256), dim3(256), 0, st, imgs, img, bstar, kind, coef, dccum, mcu_rst);
    hipLaunchKernelGGL(k_dc_scan, dim3(1), dim3(DC_THREADS), 0, st, imgs + img, tables, dccum, (const uint8_t*)mcu_rst);
    hipLaunchKernelGGL(k_dead_rows, dim3(64), dim3(256), 0, st, imgs, img, bstar, dccum);
}
void js_launch_dc_scan(hipStream_t st, const JsImage* imgs, uint32_t nimg, const JsTableSet* tables, int16_t* dccum, const uint8_t* mcu_rst, void* parts_scratch)
{
    if (!nimg) return;
    if (parts_scratch && nimg <= JS_DC_PARTS_IMAGES) {           // small job: two-level scan, DC_PARTS_MAX workgroups per image
        hipLaunchKernelGGL(k_dc_scan_parts, dim3(DC_PARTS_MAX, nimg), dim3(DC_THREADS), 0, st, imgs, tables, dccum, mcu_rst, (DcSeg*)parts_scratch, 0);
        hipLaunchKernelGGL(k_dc_scan_parts, dim3(DC_PARTS_MAX, nimg), dim3(DC_THREADS), 0, st, imgs, tables, dccum, mcu_rst, (DcSeg*)parts_scratch, 1);
    } else hipLaunchKernelGGL(k_dc_scan, dim3(nimg), dim3(DC_THREADS), 0, st, imgs, tables, dccum, mcu_rst);
}
