// jsnoop_host.h -- host-side objects behind the opaque handles of include/jsnoop_gpu.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <utility>
#include <vector>
#include "../../include/jsnoop_gpu.h"
#include "jsnoop_types.h"

// The table / frame state that CjfifDecode pushes into CimgDecode through its setters
// (member arrays of reference source/ImgDecode.h:531-615, same shapes).
struct JsTables {
    uint16_t dqt_nat[4][64], dqt_zz[4][64]; int dqt_sel[256];
    int      dht_sel[2][5];
    uint32_t dht_setmax[2], dht_size[2][4];
    uint32_t dht_bitlen[2][4][JS_DHT_CODES], dht_bits[2][4][JS_DHT_CODES], dht_mask[2][4][JS_DHT_CODES], dht_code[2][4][JS_DHT_CODES];
    uint32_t dht_fast[2][4][2 << JS_FAST_BITS];
    uint32_t dht_histo_unused[1];
    int      details_set; uint32_t dim_x, dim_y, num_sof, num_sos, precision;
    uint32_t samp_h[256], samp_v[256];
    int      rst_en; uint32_t rst_interval;
};

struct JsnoopBatch;
struct JsProgBatch;          // progressive-path state of a batch (jsnoop_progressive.cpp)

struct JsnoopDecoder {
    JsTables t;
    int opt_decode_ac, opt_histo_en, opt_stat_clip_en; unsigned opt_err_max; int opt_dump_histo_y = 0;
    struct Overlay { std::vector<uint8_t> data; unsigned begin; };      // CwindowBuf overlays (WindowBuf.cpp:516-620)
    std::vector<Overlay> overlays;
    jsnoop_log_fn log_fn; void* log_user;
    JsnoopBatch* batch;                         // private batch of one image (single-image API)
    int img = 0;                                // index of this object's image in `batch` (0 there; a view onto image i of a caller's batch otherwise)
    unsigned preview_mode; int shift_y, shift_cb, shift_cr; unsigned shift_mcu_x, shift_mcu_y;
    unsigned ins_mcu_x = 0, ins_mcu_y = 0, ins_mcu_len = 0;      // m_nPreviewInsMcu* (:682-699), stored only
    bool preview_is_jpeg, have_image; int host_valid; int last_path; uint32_t last_flags;
    unsigned geom[8];
    // Host copies of the DIB / the planes (GetBitmapPtr, GetPixMapPtrs): page-locked.  A read-back into pageable memory is slower (a
    // 1080p DIB: 2-3 ms instead of 0.15) and leaves the runtime in a state in which every later synchronisation of the process
    // costs more (a single-image progressive decode: 2.8 -> 5.1 ms per call after one GetBitmapPtr).
    struct Pinned { void* p = nullptr; size_t cap = 0; int ensure(size_t need); ~Pinned(); };
    Pinned h_dib, h_planes; std::vector<uint32_t> h_side;
    // what the report reads back besides the side block, fetched with it in ONE round trip (fetch_side): the device's event list, the per-MCU restart marks,
    // chroma + BGRA of the brightest pixel (k_bright_probe) -- valid for the decode they were fetched behind
    std::vector<uint32_t> h_events; std::vector<uint8_t> h_rstf; uint32_t h_bright[8] = {0}; bool report_cache = false, want_bright = false;
    uint32_t zero_histo[2 * 4 * 17] = {0};
    // what the reference keeps for a preview that does not come from the scan decoder (m_pDibTemp filled by the PSD decoder, m_bDibTempReady,
    // m_rectImgBase: source/ImgDecode.h:508-510, SetImageDimensions :2706)
    std::vector<uint8_t> dib_temp; bool dib_temp_ready = false; unsigned base_w = 0, base_h = 0;
    void reset_dqt_tables();                                       // ResetDqtTables :343
    void reset_dht_lookup();                                       // ResetDhtLookup :373
    JsnoopDecoder();
    void reset_state();
    void log(int level, const char* fmt, ...);
    void fetch_side();
    void ensure_side();      // side outputs are produced on first request when the parallel path decoded the image
    bool side_ready;
    void rerender();
    // bHistoEn / bStatClipEn statistics (m_sHisto, m_sStatClip, m_anCcHisto_*, m_anHistoYFull): cleared by DecodeScanImg
    // (:3145-3155), accumulated by every CalcChannelPreview; m_nWarnYccClipNum only restarts in Reset() (:130)
    uint32_t stats[2482]; unsigned warn_ycc_clip; bool hist_latched, clip_latched;
    void stats_pass();
    unsigned head_events = 0, head_counted = 0;                   // messages of the reader's very first refill (BuffTopup :3019, in front of the heading :3022): how many, how many of them counted
    std::vector<std::pair<int, std::string>> pending_log;          // CapYccRange warnings of the current CalcChannelPreview
    void flush_pending_log();
};

struct JsDeviceArenas {
    uint8_t* raw; uint8_t* ustr; int16_t* coef; int16_t* dccum; uint8_t* dib; int16_t* planes; uint32_t* side;
    JsImage* imgs; JsTableSet* tables; uint32_t* wg_base; uint32_t* sel; uint64_t* sums; uint8_t* sub; uint8_t* probe;
    uint32_t* seg; uint32_t* chunk_keep; uint32_t* chunk_rst; uint32_t* us_base; uint32_t* sy_base; uint8_t* mcu_rst; uint32_t* flags; uint8_t* ustr_lin;
    uint32_t* events; uint8_t* dc_parts; uint32_t* cand; uint32_t* cand_req; unsigned long long* wg_part;
    unsigned long long* us_state;                                 // chained-scan state of the fused un-stuffing pass, one word per chunk (k_unstuff_write<true>)
};
struct JsArenaCaps { size_t raw, ustr, coef, dccum, dib, planes, side, imgs, tables, wg_base, sel, sums, sub, probe,
                            seg, chunk_keep, chunk_rst, us_base, sy_base, mcu_rst, flags, ustr_lin, events, dc_parts, cand, cand_req, wg_part, us_state; };

struct JsnoopBatch {
    int device; hipStream_t stream; bool own_stream;
    int color_stats_pass(int i, bool hist_en, uint32_t* acc /*2482 words, accumulated into*/, unsigned* warn_used,
                         std::vector<std::pair<int, std::string>>* notes = nullptr, uint32_t pos0 = 0, uint32_t align = 0);
    int opt_decode_ac, opt_want_planes, opt_force_exact, opt_events = 0;   // opt_events: keep the decoder's event log (single-image API)
    uint64_t event_words = 0;
    std::vector<JsImage> imgs; std::vector<JsTableSet> tables;
    struct JsImgHost { uint32_t dht_setmax[2] = { 0, 0 }; unsigned err_max = 20; bool display = true; };   // what the per-image report needs beyond the descriptor
    std::vector<JsImgHost> hinfo;
    // js_side_all: the side passes of all clean images behind one decode (mask | positions of every image | inverse map of every chunk), the requests counted per decode
    uint8_t* d_side_all = nullptr; size_t side_all_cap = 0; int side_requests = 0;
    uint64_t rec_words = 0;                                       // MCU-top positions of all images (JsImage::rec_off)
    uint32_t* rec_pos = nullptr;                                  // js_side_prepare: where the NEXT decode's write pass records MCU-top positions (null: it does not)
    std::vector<uint8_t> side_pre;                                // per image: 1 = the clean-image side pass was enqueued behind the decode (js_side_prelaunch)
    std::vector<uint8_t> side_mode;                               // per image: who produced its side outputs last (1 = parallel side pass, 2 = the exact-mirror reader, 3 = parallel side pass + chunked exact readers)
    std::vector<std::vector<uint32_t>> side_anoms;                // per image: the coefficient-index overflows of the side walk, in block order (4 words each)
    std::vector<uint8_t> side_done;                               // per image: the side-output pass has run since the last decode
    std::vector<uint8_t> side_chunk_ok;                           // per image: a flagged image whose every MCU top the parallel walks vouch for (up to the end of the reference's own decode): its report comes from k_side_chunks
    std::vector<std::vector<uint32_t>> side_events;               // per image: the messages of the chunked side pass in the reference's order (JS_EV_WORDS each, not yet gated by the warning counter)
    uint32_t* d_chunk_tmp = nullptr; size_t chunk_tmp_cap = 0;    // scratch of the chunked side pass
    std::vector<uint32_t> host_anom;                              // per image: first block (decode order) the parallel path could not vouch for (0xFFFFFFFF: none)
    std::vector<uint8_t> host_anom_kind;                          // ... and what it was: 0 = the mirror takes over there, 1..8 = the reference's decode ends in that block (ANOM_KEY, jsnoop_kernels.hip)
    std::vector<uint32_t> host_flags, host_path, h_us_base, h_us4_base, h_sy_base, h_sn_base, h_wg_base;
    // A large batch decodes as two halves on two streams (stream, aux[0]): the kernels of one half fill the tails and the thinly
    // populated phases (second synchronisation launch, DC scan) of the other -- 14.2 -> 13.7 ms per 1024 images.  Both halves live in
    // the same arenas; a launch over a half passes pointers to its first image and its first prefix entry.
    int split_parts = 1;                                          // what tune.split came to for the uploaded batch
    JsnoopTuning tune;                                            // how this batch decodes (jsnoop_batch_set_tuning; process defaults: js_env_tuning)
    hipEvent_t ev2[JSNOOP_NUM_STAGES + 1];                        // stage events of the second half (timed decodes)
    bool last_timed_split = false;
    bool last_used_parallel = false;                              // false: no table set of the batch fits the parallel path, the exact-mirror kernel decoded everything
    uint32_t* d_side_tmp = nullptr; size_t side_tmp_cap = 0;      // scratch of the side-output pass (one image at a time)
    uint8_t* pinned; size_t pinned_cap; uint64_t raw_bytes;
    uint8_t* d2h_land = nullptr;                                  // page-locked landing buffer of the read-back calls (32 MiB, on first use)
    uint8_t* h_desc = nullptr; size_t h_desc_cap = 0; hipEvent_t ev_up = nullptr;   // page-locked staging block of the descriptors (upload), the event behind its copies
    int  d2h_staged(void* dst, const void* src, size_t bytes);
    JsDeviceArenas dev; JsArenaCaps cap;
    bool uploaded;
    uint32_t us_ticket_base[2] = { 0, 0 };                        // value of the two chunk-ticket counters (behind us_state; one per part of a split decode) before the next launch
    uint32_t us_epoch = 0;                                        // decode counter 1..255 the chained-scan state words are tagged with (cleared at upload)
    uint64_t total_blocks, dib_bytes, side_words, total_subseq, ustr_bytes, seg_words, mcu_bytes; uint32_t total_wgs, strips_per_wg, us_chunks, sy_wgs, sn_wgs, max_mcu_h, max_mcu_w;
    // small jobs (64-byte sub-sequences, a few hundred thousand walks at most) synchronise by candidates (k_cand_*) instead of k_sync's rounds: cand_rounds
    // = fill rounds of the chain, -1 = off (JSNOOP_CAND=0, more than JS_CAND_MAX_BLK blocks per MCU, a larger job); cand_blk = most blocks per MCU in the batch
    int cand_rounds = -1; uint32_t cand_blk = 0; bool cand_half = false;   // cand_half: the smallest jobs (one large image, a handful) also run the write pass with two lanes per sub-sequence
    int sync_rounds;                 // > 0: js_launch_sync_rounds with that many list rounds (large jobs), else sync_launches launches of k_sync
    int sync_launches; int sub_wl;   // log2(words per sub-sequence): 4 / 5 / 7 = 64- / 128- / 512-byte sub-sequences (chosen per batch; 6 and 8 through the tuning struct)
    uint32_t tab_rows, tab_lut2, tab_rows_w;     // largest decode-table footprint in the batch (sizes the kernels' LDS); _w: DC rows | AC rows << 8
    hipEvent_t ev[JSNOOP_NUM_STAGES + 1];
    // helper streams for work that forks inside one decode (independent scans of a progressive file), created on first use
    enum { kAux = 4 };
    hipStream_t aux[kAux] = { nullptr, nullptr, nullptr, nullptr }; hipEvent_t aux_ev[kAux + 1] = { nullptr, nullptr, nullptr, nullptr, nullptr };
    int ensure_aux();
    JsProgBatch* prog = nullptr;                                 // set once a progressive (SOF2) file was added: the batch then holds only such files
    int  add_progressive(JsnoopDecoder* d, const uint8_t* file, size_t len);
    int  decode_progressive(bool timed);
    int  sync_progressive();
    float lut[64][64]; float* d_lut;
    explicit JsnoopBatch(void* user_stream);
    ~JsnoopBatch();
    int  init();
    void clear();
    int  reserve_pinned(size_t need);
    int  add(JsnoopDecoder* d, const uint8_t* file, size_t len, unsigned scan_start, int display, int quiet = 1);
    int  add_described(const JsImage& desc, const uint8_t* file, size_t len);
    // Image i of `src` once more in this batch (same bytes, tables, geometry, options); through_markers: its scan runs to the end of the
    // file, markers that are no RSTn and FF FF pairs staying in the stream as the data the reference reads them as (BuffAddByte :1486-1561).
    int  add_clone(const JsnoopBatch* src, uint32_t i, bool through_markers);
    JsnoopBatch* helper = nullptr; bool is_helper = false;       // a private one-image batch for second attempts at flagged images (js_parallel_fixup)
    int  tile(int total);
    int  upload();
    int  decode(bool timed);
    int  sync();
    int  read_dib(int i, uint8_t* dst);
    int  read_planes(int i, int16_t* y, int16_t* cb, int16_t* cr);
    int  run_exact(const std::vector<uint32_t>& which);       // (entropy only: the caller re-runs the back end)
    int  redo_back_end(const std::vector<uint32_t>& which);   // the back end again for the listed images (their reductions cleared first); many images: the whole batch
    int  launch_back_end_part(hipStream_t st, uint32_t i0, uint32_t n);   // ... over images [i0, i0 + n) on stream st
    int  launch_back_end(uint32_t nimg);        // k_idct_color over the first nimg images (tile size from their CURRENT preview state)
};

// flags that leave coefficients, planes and DIB of the parallel path reference-exact (bookkeeping differs: status words, warning counter, log)
#define JS_FLAGS_PIXEL_EXACT (JSNOOP_FLAG_COEF_OVERFLOW | JSNOOP_FLAG_BAD_CODE | JSNOOP_FLAG_RST_MISALIGN)
// ... of which the parallel side pass can also produce the bookkeeping (the others get theirs from the mirror's side-only pass)
#define JS_FLAGS_SIDE_PARALLEL (JSNOOP_FLAG_COEF_OVERFLOW)
int  js_clear_flags(JsnoopBatch* b);                              // flag arena: two words per image (flags, complement of the first anomalous block)
int  js_read_flags(JsnoopBatch* b);                               // -> host_flags, host_anom
void js_set_error(const char* fmt, ...);
const JsnoopTuning& js_env_tuning();                             // the process defaults: environment variables, read once
int  js_check_tuning(const JsnoopTuning& t);                      // 0 / -1 + error text
int  js_import_tuning(const JsnoopTuning* in, JsnoopTuning* out); // a caller's struct (its struct_size bytes) -> this library's, checked
void js_export_tuning(const JsnoopTuning& t, JsnoopTuning* out);  // ... and back, no byte past the caller's struct_size
void js_debug_cand_links(JsnoopBatch* b, hipStream_t st, uint32_t i0, uint32_t n);   // JSNOOP_DBG_CAND_LINKS (jsnoop_parallel.cpp)
// roctx ranges around the host-side stages (rocprofv3 --marker-trace makes a timeline self-describing: upload / clear / entropy
// stages / back end / read-back).  One push and pop per stage and call: nothing per image.
#include <rocprofiler-sdk-roctx/roctx.h>
struct JsRange { explicit JsRange(const char* name) { roctxRangePushA(name); } ~JsRange() { roctxRangePop(); } };
bool js_geometry(JsnoopDecoder* d, JsImage* im);
bool js_describe_image(JsnoopDecoder* d, JsImage* im, JsTableSet* ts, uint32_t file_len, uint32_t scan_start, int display, int quiet, const uint8_t* file = nullptr);
void js_emit_head_events(JsnoopDecoder* d, const uint8_t* file, size_t len, uint32_t scan_start);      // jsnoop_report.cpp: what the very first refill logs, in front of the heading
void js_emit_decode_events(JsnoopDecoder* d);                      // jsnoop_report.cpp
void js_emit_report(JsnoopDecoder* d, bool display, bool quiet);
void js_build_parallel_luts(JsTableSet* ts, uint32_t ncomp);          // jsnoop_parallel.cpp
int js_selftest_tables(unsigned seed, unsigned rounds);                // jsnoop_parallel.cpp (host only)
int  js_parallel_entropy_part(JsnoopBatch* b, hipStream_t st, uint32_t i0, uint32_t n, hipEvent_t* evs /*stage events or null*/, hipEvent_t after_stage = nullptr /* recorded behind stage `which_stage` (1 un-stuffing, 2 synchronisation) */, int which_stage = 0);   // images [i0, i0 + n)
int  js_parallel_entropy(JsnoopBatch* b, bool timed);                 // 1 = launched, 0 = not applicable, <0 error
int  js_parallel_fixup(JsnoopBatch* b);
int  js_side_only(JsnoopBatch* b, uint32_t i);
int  js_side_prepare(JsnoopBatch* b, uint32_t i);                 // ... and before the decode: its write pass records what the side walk would (one image, 64-byte pieces)
int  js_side_prelaunch(JsnoopBatch* b, uint32_t i);               // the clean-image side pass enqueued behind the decode, before the wait (a caller that wants the report anyway)
int  js_jfif_walk(JsnoopDecoder* d, const uint8_t* file, size_t len, unsigned* scan_start);   // jfif_front.cpp
size_t js_prog_count(const JsnoopBatch* b);                            // jsnoop_progressive.cpp: progressive images in the batch
void js_prog_clear(JsnoopBatch* b);
void js_prog_dirty(JsnoopBatch* b);
void js_prog_free(JsnoopBatch* b);
void js_prog_dup(JsnoopBatch* b, uint32_t src, uint32_t dst);
bool js_is_progressive(const uint8_t* file, size_t len);               // first SOF marker of the stream is SOF2
