// jsnoop_progressive.h -- descriptors shared by the progressive (SOF2) host code and kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "jsnoop_types.h"

struct JsProgTable {                 // canonical Huffman decode table (T.81 Annex C / F.2.2.3)
    uint16_t look[256];              // 8-bit look-ahead: len << 8 | symbol for codes of <= 8 bits, 0 = longer
    int32_t  maxcode[18];            // largest code of each length, -1 when the length is unused
    int32_t  valoff[17];             // symbol index = code + valoff[len]
    uint8_t  sym[256];
    uint32_t nsym;
};
struct JsProgFrame {                 // SOF2 frame: per component (0-based, frame order)
    uint32_t ncomp, hs[3], vs[3], first_blk[3];
    uint16_t qnat[3][64];            // quantiser, natural order
};
struct JsProgScan {
    uint32_t img;                    // image of the batch this scan belongs to
    uint32_t ncomp, comp[3];         // scan components (frame indices)
    uint32_t ntabs, tab[4];          // tables this scan uses (indices into the batch's table array), staged per workgroup
    uint32_t dc_slot[3], ac_slot[3]; // per scan component: which of tab[] is its DC / AC table
    uint32_t ss, se, ah, al;
    uint32_t seg_first, nseg, rst_interval;   // seg_first: index into the batch's interval array
    uint32_t nbx, nby;               // block grid of a non-interleaved scan (A.2.3: not padded to whole MCUs)
};
struct JsProgSeg { uint32_t start, end; };   // entropy bytes of one restart interval, file-relative [start, end)

// One launch decodes a whole dependency level of a batch: `nsc` scans (indices in lvl_scans) of any of its images, workgroup ->
// scan through the exclusive prefix lvl_wg (nsc + 1 entries).  status: 4 words per image.
void js_launch_prog_level(hipStream_t st, const JsImage* imgs, const JsProgFrame* frames, const JsProgScan* scans, const uint32_t* lvl_scans, const uint32_t* lvl_wg,
                          uint32_t nsc, uint32_t total_wgs, uint32_t pg_lanes /*1, 2, 4, 8: intervals per wave of the sequential scan kinds*/, const JsProgTable* tabs, const JsProgSeg* segs, const uint8_t* raw, int16_t* coef, uint32_t* status);
void js_launch_prog_finalize(hipStream_t st, const JsImage* imgs, const JsProgFrame* frames, uint32_t nimg, const uint32_t* blk_base /*nimg + 1*/, uint32_t total_blocks,
                             int16_t* coef, int16_t* dccum);
uint32_t js_prog_wgs_of(const JsProgScan& sc, uint32_t pg_lanes);       // workgroups a scan needs in js_launch_prog_level
