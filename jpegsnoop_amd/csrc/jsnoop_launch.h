// jsnoop_launch.h -- host-callable launch wrappers of the kernels in jsnoop_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "jsnoop_types.h"

void js_launch_entropy_exact(hipStream_t st, const JsImage* imgs, const uint32_t* sel, uint32_t nsel, const JsTableSet* tables,
                             const uint8_t* raw, int16_t* coef, int16_t* dccum, uint32_t* side, int side_only, uint32_t* events);
int  js_launch_idct_color(hipStream_t st, const JsImage* imgs, const uint32_t* wg_base, uint32_t nimg, uint32_t total_wgs, uint32_t tile_bytes,
                          const float* lut_t, const int16_t* coef, const int16_t* dccum, uint8_t* dib, int16_t* planes, uint32_t* side,
                          int layout /* 0 = mixed; 1..4 = every image has the fast layout with chroma expansion (2,2) / (2,1) / (1,2) / (1,1) */,
                          unsigned long long* wg_part /* null, or 2 words per workgroup (indexed like wg_base): brightest-pixel / luminance records folded by a second kernel */);
void js_launch_unaligned_probe(hipStream_t st, void* buf48);    // one 16-byte store at byte offset 6 of buf48 (halves 1..8), see k_unaligned_probe
void js_launch_idct_probe(hipStream_t st, const float* lut_t, const int16_t* coef64, float* out64);
void js_launch_clear3(hipStream_t st, void* a, size_t a_bytes, void* b, size_t b_bytes, void* c, size_t c_bytes, const JsImage* imgs = nullptr, uint32_t nimg = 0);   // three arenas to zero in one launch (sizes rounded up to 16 bytes)
void js_launch_clear5(hipStream_t st, uint32_t* a, size_t na, uint32_t* b, size_t nb, uint32_t* c, size_t nc, uint32_t* d, size_t nd, uint32_t* e, size_t ne);   // five word ranges to zero in one launch
void js_launch_dib_checksum(hipStream_t st, const JsImage* imgs, uint32_t nimg, const uint8_t* dib, unsigned long long* sums);
void js_launch_color_probe(hipStream_t st, int y, int cb, int cr, uint32_t* out);
void js_launch_bright_probe(hipStream_t st, const JsImage* imgs, uint32_t img, const uint32_t* side, const int16_t* planes /*or null*/, uint32_t* out8);   // chroma + BGRA of the brightest pixel, and the key they belong to
#define JS_STATS_WORDS 2482          /* public layout, include/jsnoop_gpu.h JSNOOP_STATS_WORDS */
#define JS_STATS_DEV_WORDS 2496      /* + the six uncapped YCC range-event totals, padded */
void js_launch_color_stats(hipStream_t st, const JsImage* imgs, uint32_t img, const int16_t* planes, int hist_en, uint32_t* stats);
void js_launch_clip_order(hipStream_t st, const JsImage* imgs, uint32_t img, const int16_t* planes, uint32_t budget, uint32_t* out6);
void js_launch_tiff_pack(hipStream_t st, const JsImage* imgs, uint32_t img, const uint8_t* dib, const int16_t* planes, int mode, uint8_t* out);
void js_launch_color_sweep(hipStream_t st, uint32_t* out /*2^24 words*/);
void js_launch_unstuff(hipStream_t st, int wl, const JsImage* imgs, const uint32_t* us_base, uint32_t nimg, uint32_t total_chunks, const uint8_t* raw,
                       uint32_t* chunk_keep, uint32_t* chunk_rst, uint8_t* ustr_lin, uint8_t* ustr, uint32_t* seg_tab, uint32_t* side, uint32_t* flags,
                       const uint32_t* sy_base, uint32_t sy_wgs, unsigned long long* us_state /*nullptr: the three-pass form*/, uint32_t epoch,
                       const uint32_t* us4_base /*prefix over super-chunks of four chunks, or nullptr: linear output + transposition pass*/, uint32_t total_super,
                       uint32_t* ticket /*device counter the fused passes take their chunk from*/, uint32_t* ticket_base /*host: its value before the launch; advanced here*/);
void js_launch_sync(hipStream_t st, int wl, uint32_t tab_rows, uint32_t tab_lut2, const JsImage* imgs, const uint32_t* sy_base, uint32_t nimg, uint32_t total_wgs, const JsTableSet* tables,
                    const uint8_t* ustr, const uint32_t* seg_tab, const uint32_t* side, uint32_t* sub, uint64_t nsub, int first_pass, uint32_t it_max = 0 /* rounds at most; 0: to the fixed point */);
// the large-job form of the synchronisation stage: k_sync cut after two rounds, list rounds (k_sync_links / k_sync_round), k_sync's verification mode
#define JS_SYR_SLOTS 12                /* list counters per image */
#define JS_SYR_ROUNDS 8                /* list rounds of a large job (k_sync alone needs 4.8 rounds per workgroup on average, 7 at most in 834 workgroups read) */
size_t js_sync_list_words(uint64_t nsub, uint32_t nimg);
void js_launch_sync_rounds(hipStream_t st, int wl, uint32_t tab_rows, uint32_t tab_lut2, const JsImage* imgs, const uint32_t* sn_base, uint32_t sn_wgs, const uint32_t* sy_base, uint32_t sy_wgs,
                           uint32_t nimg, const JsTableSet* tables, const uint8_t* ustr, const uint32_t* seg_tab, const uint32_t* side, uint32_t* sub, uint64_t nsub,
                           uint32_t* lists /* js_sync_list_words(nsub, images of the BATCH) words */, uint32_t* rcnt /* this part's counters inside it */, int rounds);
#define JS_SY_THREADS 256
#define JS_SY_HALO    2            // lanes of a k_sync workgroup that walk in front of its first sub-sequence
// Candidate synchronisation (the small-job form of the synchronisation stage: k_cand_spec / _walk / _chain / _fill / _apply); leaves the
// sub-sequence arrays as js_launch_sync would, open links marked for a js_launch_sync(..., first_pass = 2) behind it.
#define JS_CAND_MAX_BLK 6            /* images with more blocks per MCU than this synchronise the classic way */
#define JS_CAND_REQ_WORDS 12         /* per image: diagnostics of the chain */
size_t js_cand_bytes(uint64_t nsub);
void js_launch_cand_sync(hipStream_t st, int wl, uint32_t tab_rows, uint32_t tab_lut2, const JsImage* imgs, const uint32_t* sy_base, uint32_t nimg, uint32_t sy_wgs, uint32_t max_blk,
                         const JsTableSet* tables, const uint8_t* ustr, const uint32_t* seg_tab, const uint32_t* side, uint32_t* sub, uint64_t nsub, uint32_t* cand, uint32_t* req, int fill_rounds,
                         int mid /* the memo walks report middle states for a two-lanes-per-sub-sequence write pass (js_launch_write with the same arena) */);
void js_launch_block_scan(hipStream_t st, int wl, const JsImage* imgs, uint32_t nimg, const JsTableSet* tables, uint32_t* sub, uint64_t nsub, uint32_t* side, uint32_t* flags);
void js_launch_write(hipStream_t st, int wl, uint32_t tab_rows, uint32_t tab_lut2, const JsImage* imgs, const uint32_t* sy_base, uint32_t nimg, uint32_t total_wgs, const JsTableSet* tables,
                     const uint8_t* ustr, const uint32_t* seg_tab, uint32_t* side, uint32_t* sub, uint64_t nsub,
                     int16_t* coef, int16_t* dccum, uint8_t* mcu_rst, uint32_t* flags,
                     uint32_t* cand_half /* null, or the candidate arena of a job that was synchronised by candidates: two lanes per sub-sequence (64-byte pieces only) */, bool v1,
                     uint32_t* rec_pos = nullptr /* 64-byte pieces, !v1: the pass records MCU-top bit positions here and the code-length histogram in the side block (the side walk's outputs) */);
// side-output pass over one image the parallel path decoded (MCU file map, block-DC maps, code-length histogram, status words)
void js_launch_side_pass(hipStream_t st, int wl, uint32_t tab_rows, uint32_t tab_lut2, const JsImage* imgs, const uint32_t* us_base, const uint32_t* sy_base, uint32_t nimg,
                         uint32_t img, uint32_t us_wg0, uint32_t us_wgs, uint32_t sy_wg0, uint32_t sy_wgs, const JsTableSet* tables, const uint8_t* raw,
                         const uint32_t* chunk_keep, const uint32_t* chunk_rst, const uint8_t* ustr, uint32_t* seg_tab, uint32_t* side, uint32_t* sub, uint64_t nsub,
                         const int16_t* dccum, uint8_t* mcu_rst, uint32_t* mcu_pos, uint32_t* us_out, uint32_t* events,
                         uint32_t* anoms, uint32_t dead_blk = 0xFFFFFFFFu, uint32_t cut_mcu = 0xFFFFFFFFu /* [0] count, 4 words per record from [4]: block, bit position of the symbol, index it ran to, bit position behind the block; or null */, bool walked = false /* the decode's write pass recorded positions + histogram: no side walk */);
// ... of every image whose byte of img_mask is set, in four launches (pos_all: rec_off-indexed positions, zeroed by the caller; us_all: 256 words per chunk of the batch)
void js_launch_side_pass_all(hipStream_t st, int wl, uint32_t tab_rows, uint32_t tab_lut2, const JsImage* imgs, const uint32_t* us_base, const uint32_t* sy_base, uint32_t nimg,
                             uint32_t us_wgs, uint32_t sy_wgs, const JsTableSet* tables, const uint8_t* raw, const uint32_t* chunk_keep, const uint32_t* chunk_rst, const uint8_t* ustr,
                             uint32_t* seg_tab, uint32_t* side, uint32_t* sub, uint64_t nsub, const int16_t* dccum, uint8_t* mcu_rst, uint32_t* pos_all, uint32_t* us_all,
                             uint32_t* events, const uint8_t* img_mask);
#define JS_DC_PARTS_IMAGES 8         /* batches of up to this many images take the two-level DC scan */
#define JS_DC_PARTS_BYTES (JS_DC_PARTS_IMAGES * 64 * 16)
void js_launch_dead_fill(hipStream_t st, const JsImage* imgs, uint32_t img, uint32_t bstar, uint32_t kind /*ANOM_KEY's death kinds 1..8*/,
                         const JsTableSet* tables, int16_t* coef, int16_t* dccum, uint8_t* mcu_rst);
void js_launch_dc_scan(hipStream_t st, const JsImage* imgs, uint32_t nimg, const JsTableSet* tables, int16_t* dccum, const uint8_t* mcu_rst, void* parts_scratch /*JS_DC_PARTS_BYTES or null*/);
#define JS_US_CHUNK 4096
#define JS_SC_HDR 145                 // words of a chunk record in front of its events (k_side_chunks)
void js_launch_side_chunks(hipStream_t st, const JsImage* imgs, uint32_t img, const JsTableSet* tables, const uint8_t* raw, const uint32_t* seg_tab, const uint8_t* mcu_rst,
                           const uint32_t* mcu_pos, const uint32_t* us_out, uint32_t us_threads, uint32_t* side, const int16_t* dccum, uint32_t ch_mcus, uint32_t nchunks, uint32_t ev_cap,
                           const uint32_t* mcus_left0, uint32_t* recs, uint32_t* map_own, unsigned long long* map_beyond, uint32_t run_on_mcu /*~0: none*/, uint32_t* fill_desc /*64 words, zeroed*/);
void js_launch_tail_pass(hipStream_t st, int wl, uint32_t tab_rows, uint32_t tab_lut2, const JsImage* imgs, const uint32_t* us_base, const uint32_t* sy_base, uint32_t nimg,
                         uint32_t us_wg0, uint32_t us_wgs, uint32_t sy_wg0, uint32_t sy_wgs, const JsTableSet* tables, const uint8_t* raw,
                         const uint32_t* chunk_keep, const uint32_t* chunk_rst, const uint8_t* ustr, uint32_t* seg_tab, uint32_t* side, uint32_t* sub, uint64_t nsub,
                         int16_t* coef, int16_t* dccum, uint8_t* mcu_rst, uint32_t* mcu_pos, uint32_t* us_out, const uint32_t* flags /*the batch's flag arena*/, const uint32_t* sel1 /*device: the image index*/);
