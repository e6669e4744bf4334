// jsnoop_host.cpp -- host side of libjsnoop_gpu.so: the CimgDecode-shaped decoder state,
// batch arenas in HBM, pinned staging (the CwindowBuf replacement) and the C ABI of
// include/jsnoop_gpu.h.  Reference line numbers refer to source/ImgDecode.cpp unless
// another file is named.  There is no CPU decode path in this library: every pixel
// comes out of the HIP kernels in jsnoop_kernels.hip.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <chrono>
#include <algorithm>
#include <atomic>
#include "../../include/jsnoop_gpu.h"
#include "jsnoop_types.h"
#include "jsnoop_launch.h"
#include "jsnoop_host.h"
#include "jsnoop_bytes.h"

// ------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
void js_set_error(const char* fmt, ...)
{
    char buf[1024]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap); g_err = buf;
}
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    js_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); return -1; } } while (0)

// for the void entry points (the reference's getters return nothing): a failed device call is recorded for jsnoop_last_error()
#define HIP_NOTE(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) js_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

static thread_local int g_device = 0;        // jsnoop_set_device is per host thread (one thread per GPU is the natural use of the C ABI)

static const uint8_t kZigZag[64] = {
     0, 1, 8,16, 9, 2, 3,10, 17,24,32,25,18,11, 4, 5, 12,19,26,33,40,48,41,34, 27,20,13, 6, 7,14,21,28,
    35,42,49,56,57,50,43,36, 29,22,15,23,30,37,44,51, 58,59,52,45,38,31,39,46, 53,60,61,54,47,55,62,63 };

// ------------------------------------------------------------------------------ decoder state
JsnoopDecoder::JsnoopDecoder()
{
    memset(&t, 0, sizeof t);
    opt_decode_ac = 1; opt_histo_en = 0; opt_stat_clip_en = 0; opt_err_max = 20;
    log_fn = nullptr; log_user = nullptr; batch = nullptr;
    preview_mode = 1; shift_y = shift_cb = shift_cr = 0; shift_mcu_x = shift_mcu_y = 0;
    preview_is_jpeg = false; have_image = false; host_valid = 0; last_path = 0; last_flags = 0; side_ready = false;
    memset(geom, 0, sizeof geom);
    memset(stats, 0, sizeof stats); warn_ycc_clip = 0; hist_latched = clip_latched = false;
    reset_state();
}
void JsnoopDecoder::log(int level, const char* fmt, ...)
{
    if (!log_fn) return;
    char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    log_fn(log_user, level, buf);
}
void JsnoopDecoder::reset_dht_lookup()                                         // ResetDhtLookup :373-415
{
    memset(t.dht_histo_unused, 0, sizeof t.dht_histo_unused);
    memset(t.dht_setmax, 0, sizeof t.dht_setmax); memset(t.dht_size, 0, sizeof t.dht_size);
    memset(t.dht_bitlen, 0, sizeof t.dht_bitlen); memset(t.dht_bits, 0, sizeof t.dht_bits);
    memset(t.dht_mask, 0, sizeof t.dht_mask); memset(t.dht_code, 0, sizeof t.dht_code);
    memset(t.dht_fast, 0xFF, sizeof t.dht_fast);                               // DHT_CODE_UNUSED :391
    for (int c = 0; c < 2; c++) for (int i = 0; i < 5; i++) t.dht_sel[c][i] = -1;
}
void JsnoopDecoder::reset_dqt_tables()                                         // ResetDqtTables :343-361
{
    for (int i = 0; i < 256; i++) t.dqt_sel[i] = -1;
    memset(t.dqt_nat, 0, sizeof t.dqt_nat); memset(t.dqt_zz, 0, sizeof t.dqt_zz);
    t.num_sof = 0;
}
void JsnoopDecoder::reset_state()                                              // ResetState :286-306
{
    reset_dht_lookup(); reset_dqt_tables();
    memset(t.samp_h, 0, sizeof t.samp_h); memset(t.samp_v, 0, sizeof t.samp_v);
    t.details_set = 0; t.num_sof = 0; t.num_sos = 0; t.precision = 0;
}

// The geometry and readiness checks DecodeScanImg performs before its MCU loop (:2755-3123),
// restated on the table state.  Returns false (with a log line) exactly where the reference returns early.
// Geometry of the scan as DecodeScanImg derives it (:2753-2872): MCU size, block layout of an MCU, MCU-rounded image size.
bool js_geometry(JsnoopDecoder* d, JsImage* im)
{
    JsTables& t = d->t;
    memset(im, 0, sizeof *im);
    if (!t.details_set) { d->log(2, "*** ERROR: Decoding image before Image components defined ***"); return false; }
    if (t.num_sos != 1 && t.num_sos != 3) { d->log(1, "  NOTE: Number of SOS components not supported [%u]", t.num_sos); return false; }
    uint32_t hmax = 0, vmax = 0;
    for (uint32_t c = 1; c <= t.num_sos; c++) { hmax = std::max(hmax, t.samp_h[c]); vmax = std::max(vmax, t.samp_v[c]); }
    if (t.num_sos == 1) {                                                       // :2805-2817
        if (t.samp_h[1] != 1 || t.samp_v[1] != 1) d->log(1, "    Altering sampling factor for single component scan to 0x11");
        t.samp_h[1] = t.samp_v[1] = 1; hmax = vmax = 1;
    }
    if (hmax == 0 || vmax == 0 || hmax > 4 || vmax > 4) {
        d->log(1, "  NOTE: Degree of subsampling factor not supported [HMax=%u, VMax=%u]", hmax, vmax); return false; }
    im->mcu_w = hmax * 8; im->mcu_h = vmax * 8; im->ncomp = t.num_sos;
    uint32_t nb = 0;
    for (uint32_t c = 1; c <= t.num_sos; c++) {
        if (t.samp_h[c] == 0 || t.samp_v[c] == 0) { d->log(2, "*** ERROR: zero sampling factor for component %u ***", c); return false; }
        im->samp_h[c] = t.samp_h[c]; im->samp_v[c] = t.samp_v[c];
        im->expand_h[c] = hmax / t.samp_h[c]; im->expand_v[c] = vmax / t.samp_v[c];
        for (uint32_t v = 0; v < t.samp_v[c]; v++) for (uint32_t h = 0; h < t.samp_h[c]; h++) {
            im->blk_comp[nb] = (uint8_t)c; im->blk_ch[nb] = (uint8_t)h; im->blk_cv[nb] = (uint8_t)v; nb++; }
    }
    im->blk_per_mcu = nb;
    im->dim_x = t.dim_x; im->dim_y = t.dim_y;
    im->mcu_xmax = t.dim_x / im->mcu_w + (t.dim_x % im->mcu_w ? 1 : 0);
    im->mcu_ymax = t.dim_y / im->mcu_h + (t.dim_y % im->mcu_h ? 1 : 0);
    im->blk_xmax = im->mcu_xmax * hmax; im->blk_ymax = im->mcu_ymax * vmax;
    if (im->blk_xmax == 0 || im->blk_ymax == 0) return false;
    im->img_x = im->mcu_xmax * im->mcu_w; im->img_y = im->mcu_ymax * im->mcu_h;
    im->total_blocks = im->mcu_xmax * im->mcu_ymax * nb;
    d->geom[0] = im->mcu_w; d->geom[1] = im->mcu_h; d->geom[2] = im->mcu_xmax; d->geom[3] = im->mcu_ymax;
    d->geom[4] = im->blk_xmax; d->geom[5] = im->blk_ymax; d->geom[6] = im->img_x; d->geom[7] = im->img_y;
    return true;
}
bool js_describe_image(JsnoopDecoder* d, JsImage* im, JsTableSet* ts, uint32_t file_len, uint32_t scan_start, int display, int quiet, const uint8_t* file)
{
    JsTables& t = d->t;
    if (!js_geometry(d, im)) return false;
    d->head_events = d->head_counted = 0;
    if (file) js_emit_head_events(d, file, file_len, scan_start);   // the reader is filled BEFORE the heading is written (:3007-3019): a marker in the scan's first bytes is reported above it
    if (!quiet) { d->log(0, "*** Decoding SCAN Data ***"); d->log(0, "  OFFSET: 0x%08X", scan_start); }            // :3021-3025
    if (t.num_sof != 1 && t.num_sof != 3) { d->log(1, "  NOTE: Number of Image Components not supported [%u]", t.num_sof); return false; }
    for (uint32_t i = 1; i <= t.num_sos; i++) if (t.dqt_sel[i] < 0) {
        d->log(2, "*** ERROR: Decoding image before DQT Table Selection via JFIF_SOF ***"); return false; }
    bool ready = true;
    for (int cls = 0; cls < 2; cls++) for (uint32_t i = 1; i <= t.num_sos; i++) {
        int sel = t.dht_sel[cls][i];
        if (sel < 0 || sel >= 4 || t.dht_size[cls][sel] == 0) ready = false;
    }
    if (!ready) { d->log(2, "*** ERROR: Decoding image before DHT Table Selection via JFIF_SOS ***"); return false; }
    if (!quiet) {                                                                                                       // :3126-3135
        if (display && d->opt_decode_ac) d->log(0, "  Scan Decode Mode: Full IDCT (AC + DC)");
        else { d->log(0, "  Scan Decode Mode: No IDCT (DC only)");
               d->log(1, "    NOTE: Low-resolution DC component shown. Can decode full-res with [Options->Scan Segment->Full IDCT]"); }
        d->log(0, "");
    }

    im->precision = t.precision; im->rst_en = t.rst_en; im->rst_interval = t.rst_interval;
    im->decode_ac = display ? (uint32_t)d->opt_decode_ac : 0; im->err_max = d->opt_err_max;
    im->file_len = file_len; im->scan_start = scan_start;
    im->preview_mode = d->preview_mode; im->shift_y = d->shift_y; im->shift_cb = d->shift_cb; im->shift_cr = d->shift_cr;
    im->shift_mcu_x = d->shift_mcu_x; im->shift_mcu_y = d->shift_mcu_y;

    // resolve the selected tables per scan component
    memset(ts, 0, sizeof *ts);
    for (uint32_t c = 1; c <= t.num_sos; c++) {
        for (int cls = 0; cls < 2; cls++) {
            const int sel = t.dht_sel[cls][c], slot = (int)(c - 1) * 2 + cls;
            memcpy(ts->fast[slot], t.dht_fast[cls][sel], sizeof ts->fast[slot]);
            ts->size[slot] = t.dht_size[cls][sel]; ts->dest_id[slot] = (uint32_t)sel;
            memcpy(ts->bitlen[slot], t.dht_bitlen[cls][sel], sizeof ts->bitlen[slot]);
            memcpy(ts->bits[slot], t.dht_bits[cls][sel], sizeof ts->bits[slot]);
            memcpy(ts->mask[slot], t.dht_mask[cls][sel], sizeof ts->mask[slot]);
            memcpy(ts->code[slot], t.dht_code[cls][sel], sizeof ts->code[slot]);
        }
        memcpy(ts->qzz[c - 1], t.dqt_zz[t.dqt_sel[c] & 3], sizeof ts->qzz[c - 1]);
    }
    js_build_parallel_luts(ts, t.num_sos);
    return true;
}

// ------------------------------------------------------------------------------ tuning
// The environment supplies DEFAULTS, once per process (the variables of tools/README.md); batches copy them at creation and
// jsnoop_batch_set_tuning replaces the copy.  Nothing below this function reads the environment.
const JsnoopTuning& js_env_tuning()
{
    static const JsnoopTuning env = [] {
        JsnoopTuning t; memset(&t, 0, sizeof t); t.struct_size = (uint32_t)sizeof t;
        auto num = [](const char* name, long long dflt) { const char* e = getenv(name); return e ? atoll(e) : dflt; };
        auto on = [](const char* name) { return getenv(name) != nullptr; };
        { const long long w = num("JSNOOP_SUB_WL", 0); t.sub_wl = (w >= 4 && w <= 8) ? (int32_t)w : 0; }
        if (on("JSNOOP_CAND")) { const long long c = num("JSNOOP_CAND", 0); t.cand_rounds = c > 0 ? (int32_t)std::min<long long>(c, 64) : -1; }
        t.cand_max_walks = (uint64_t)std::max<long long>(0, num("JSNOOP_CAND_LANES", 0));
        t.sync_launches = (int32_t)std::max<long long>(0, num("JSNOOP_SYNC_LAUNCHES", 0));
        t.write_lanes = on("JSNOOP_NO_HALF") ? 1 : 0;
        { const long long sp = num("JSNOOP_SPLIT", 0); t.split = (sp == 1 || sp == 2) ? (int32_t)sp : 0; }
        t.mcus_per_wave = (int32_t)std::max<long long>(0, num("JSNOOP_MPW", 0));
        { const long long v = num("JSNOOP_PG_LANES", 0); t.pg_lanes = (v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 64) ? (int32_t)v : 0; }
        t.cross_checks = (on("JSNOOP_BACKEND_GENERIC") ? JSNOOP_XC_BACKEND_GENERIC : 0u) | (on("JSNOOP_WRITE_V1") ? JSNOOP_XC_WRITE_V1 : 0u) | (on("JSNOOP_NO_TAIL") ? JSNOOP_XC_NO_TAIL : 0u) |
                         (on("JSNOOP_SIDE_EXACT") ? JSNOOP_XC_SIDE_EXACT : 0u) | (on("JSNOOP_CAND_VERIFY") ? JSNOOP_XC_CAND_VERIFY : 0u) |
                         (on("JSNOOP_UNSTUFF_3PASS") ? JSNOOP_XC_UNSTUFF_3PASS : 0u);
        const long long dc = num("JSNOOP_DEBUG_CAND", 0);
        t.debug = (dc >= 1 ? JSNOOP_DBG_CAND : 0u) | (dc >= 2 ? JSNOOP_DBG_CAND_LINKS : 0u) | (on("JSNOOP_DEBUG_TAIL") ? JSNOOP_DBG_TAIL : 0u) | (on("JSNOOP_DEBUG_TIMING") ? JSNOOP_DBG_TIMING : 0u);
        // the same limits as js_check_tuning: a preset out of range falls back to "automatic" (every later set_tuning writes the whole struct back and
        // would be refused for a field its caller never touched)
        if (t.sync_launches > 64) t.sync_launches = 0;
        if (t.mcus_per_wave > 4096) t.mcus_per_wave = 0;
        return t;
    }();
    return env;
}
// struct_size is the caller's sizeof(JsnoopTuning): a caller built against an older (shorter) header gets "automatic" for the fields it does not know,
// nothing is read or written past what it owns; a caller with a LONGER struct than this library knows is refused (its extra fields mean something).
int js_import_tuning(const JsnoopTuning* in, JsnoopTuning* out)
{
    const uint32_t sz = in->struct_size;
    if (sz < 8 || sz > sizeof(JsnoopTuning)) { js_set_error("tuning: struct_size %u, this library has %zu", sz, sizeof(JsnoopTuning)); return -1; }
    memset(out, 0, sizeof *out); memcpy(out, in, sz); out->struct_size = (uint32_t)sizeof(JsnoopTuning);
    return js_check_tuning(*out);
}
void js_export_tuning(const JsnoopTuning& t, JsnoopTuning* out)      // honours out->struct_size when the caller set one (0 / garbage: the full struct, as jsnoop_tuning_defaults always did)
{
    const uint32_t want = out->struct_size;
    const size_t sz = (want >= 8 && want <= sizeof(JsnoopTuning)) ? want : sizeof(JsnoopTuning);
    JsnoopTuning tmp = t; tmp.struct_size = (uint32_t)sz;
    memcpy(out, &tmp, sz);
}
int js_check_tuning(const JsnoopTuning& t)
{
    if (t.struct_size != sizeof(JsnoopTuning)) { js_set_error("tuning: struct_size %u, this library has %zu", t.struct_size, sizeof(JsnoopTuning)); return -1; }
    if (t.sub_wl != 0 && (t.sub_wl < 4 || t.sub_wl > 8)) { js_set_error("tuning: sub_wl must be 0 or 4..8"); return -1; }
    if (t.cand_rounds < -1 || t.cand_rounds > 64) { js_set_error("tuning: cand_rounds must be -1, 0 or 1..64"); return -1; }
    if (t.sync_launches < 0 || t.sync_launches > 64) { js_set_error("tuning: sync_launches must be 0..64"); return -1; }
    if (t.write_lanes < 0 || t.write_lanes > 2) { js_set_error("tuning: write_lanes must be 0, 1 or 2"); return -1; }
    if (t.split < 0 || t.split > 2) { js_set_error("tuning: split must be 0, 1 or 2"); return -1; }
    if (t.mcus_per_wave < 0 || t.mcus_per_wave > 4096) { js_set_error("tuning: mcus_per_wave must be 0..4096"); return -1; }
    if (!(t.pg_lanes == 0 || t.pg_lanes == 1 || t.pg_lanes == 2 || t.pg_lanes == 4 || t.pg_lanes == 8 || t.pg_lanes == 16 || t.pg_lanes == 64)) { js_set_error("tuning: pg_lanes must be 0, 1, 2, 4, 8, 16 or 64"); return -1; }
    return 0;
}

// ------------------------------------------------------------------------------ batch
static inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

JsnoopBatch::JsnoopBatch(void* user_stream)
{
    stream = (hipStream_t)user_stream; own_stream = false; device = g_device;
    opt_decode_ac = 1; opt_want_planes = 0; opt_force_exact = 0;
    memset(&dev, 0, sizeof dev); memset(&cap, 0, sizeof cap);
    pinned = nullptr; pinned_cap = 0; raw_bytes = 0; uploaded = false; sync_launches = 2; sync_rounds = 0;
    tune = js_env_tuning();
    for (auto& e : ev) e = nullptr;
    for (auto& e : ev2) e = nullptr;
    d_lut = nullptr;
}
int JsnoopDecoder::Pinned::ensure(size_t need)
{
    if (need <= cap && p) return 0;
    if (p) { hipHostFree(p); p = nullptr; cap = 0; }
    if (hipHostMalloc(&p, need ? need : 1, hipHostMallocDefault) != hipSuccess) { p = nullptr; js_set_error("page-locked host buffer of %zu bytes: allocation failed", need); return -1; }
    cap = need;
    return 0;
}
JsnoopDecoder::Pinned::~Pinned() { if (p) hipHostFree(p); }
int JsnoopBatch::init()
{
    HIP_TRY(hipSetDevice(device));
    if (!stream) { HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking)); own_stream = true; }
    for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
    for (auto& e : ev2) HIP_TRY(hipEventCreate(&e));
    // PrecalcIdct :2313-2351: built once on the host in fp32, then transposed to [vu][yx] for lane-contiguous reads
    const float pi = (float)3.141592654, rh = (float)0.707106781;
    for (unsigned y = 0; y < 8; y++) for (unsigned x = 0; x < 8; x++) for (unsigned v = 0; v < 8; v++) for (unsigned u = 0; u < 8; u++) {
        float cu = (u == 0) ? rh : 1, cv = (v == 0) ? rh : 1;
        float cp = cosf((2 * x + 1) * u * pi / 16) * cosf((2 * y + 1) * v * pi / 16);
        lut[y * 8 + x][v * 8 + u] = cu * cv * cp;
    }
    std::vector<float> lt(64 * 64);
    for (int yx = 0; yx < 64; yx++) for (int vu = 0; vu < 64; vu++) lt[vu * 64 + yx] = lut[yx][vu];
    HIP_TRY(hipMalloc(&d_lut, 64 * 64 * sizeof(float)));
    HIP_TRY(hipMemcpy(d_lut, lt.data(), 64 * 64 * sizeof(float), hipMemcpyHostToDevice));
    {   // once per device: the write pass's unaligned 16-byte store must arrive as written (k_unaligned_probe)
        static std::atomic<int> ok[JS_MAX_DEVICES];                // 0 = not probed, 1 = fine, -1 = refused
        int st = device >= 0 && device < JS_MAX_DEVICES ? ok[device].load() : 0;
        if (st == 0) {
            // (what the probe can tell: a device that SPLITS or reorders the bytes of such a store.  One that faults on it raises a memory fault, which
            //  ends the process -- there is no error to return then.)
            uint16_t* p = nullptr; uint16_t h[24];
            HIP_TRY(hipMalloc((void**)&p, sizeof h));
            hipError_t pe = hipMemset(p, 0, sizeof h);
            if (pe == hipSuccess) { js_launch_unaligned_probe(stream, p); pe = hipMemcpyAsync(h, p, sizeof h, hipMemcpyDeviceToHost, stream); }
            if (pe == hipSuccess) pe = hipStreamSynchronize(stream);
            hipFree(p);                                          // (on every path)
            if (pe != hipSuccess) { js_set_error("unaligned-store probe: %s", hipGetErrorString(pe)); return -1; }
            st = 1; for (int k = 0; k < 24; k++) if (h[k] != (k >= 3 && k < 11 ? k - 2 : 0)) st = -1;
            if (device >= 0 && device < JS_MAX_DEVICES) ok[device].store(st);
        }
        if (st < 0) { js_set_error("device %d does not take unaligned 16-byte global stores (the write pass relies on them)", device); return -1; }
    }
    return 0;
}
JsnoopBatch::~JsnoopBatch()
{
    hipSetDevice(device);
    if (stream) hipStreamSynchronize(stream);
    for (void** p : { (void**)&dev.raw, (void**)&dev.ustr, (void**)&dev.coef, (void**)&dev.dccum, (void**)&dev.dib, (void**)&dev.planes,
                      (void**)&dev.side, (void**)&dev.imgs, (void**)&dev.tables, (void**)&dev.wg_base, (void**)&dev.sel, (void**)&dev.sums,
                      (void**)&dev.sub, (void**)&dev.probe, (void**)&dev.seg, (void**)&dev.chunk_keep, (void**)&dev.chunk_rst, (void**)&dev.us_base,
                      (void**)&dev.sy_base, (void**)&dev.mcu_rst, (void**)&dev.dc_parts, (void**)&dev.flags, (void**)&dev.ustr_lin, (void**)&dev.events, (void**)&dev.cand, (void**)&dev.cand_req, (void**)&dev.wg_part, (void**)&dev.us_state }) if (*p) hipFree(*p);
    delete helper; helper = nullptr;
    if (d_lut) hipFree(d_lut);
    if (d_side_tmp) hipFree(d_side_tmp);
    if (d_side_all) hipFree(d_side_all);
    if (d_chunk_tmp) hipFree(d_chunk_tmp);
    js_prog_free(this);
    if (pinned) hipHostFree(pinned);
    if (d2h_land) hipHostFree(d2h_land);
    if (h_desc) hipHostFree(h_desc);
    if (ev_up) hipEventDestroy(ev_up);
    for (auto& e : ev) if (e) hipEventDestroy(e);
    for (auto& e : ev2) if (e) hipEventDestroy(e);
    for (auto& e : aux_ev) if (e) hipEventDestroy(e);
    for (auto& a : aux) if (a) hipStreamDestroy(a);
    if (own_stream && stream) hipStreamDestroy(stream);
}
int JsnoopBatch::ensure_aux()
{
    if (aux[0]) return 0;
    HIP_TRY(hipSetDevice(device));
    for (auto& a : aux) HIP_TRY(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    for (auto& e : aux_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return 0;
}
void JsnoopBatch::clear()
{
    imgs.clear(); hinfo.clear(); tables.clear(); raw_bytes = 0; uploaded = false; host_flags.clear(); side_done.clear(); side_mode.clear(); side_anoms.clear(); side_chunk_ok.clear(); side_events.clear(); side_pre.clear();
    js_prog_clear(this);
}
int JsnoopBatch::reserve_pinned(size_t need)
{
    if (need <= pinned_cap) return 0;
    size_t ncap = std::max(need, pinned_cap * 2 + (1u << 20));
    uint8_t* np = nullptr;
    HIP_TRY(hipHostMalloc((void**)&np, ncap, hipHostMallocDefault));
    if (pinned) { memcpy(np, pinned, raw_bytes); hipHostFree(pinned); }
    pinned = np; pinned_cap = ncap;
    return 0;
}
int JsnoopBatch::add(JsnoopDecoder* d, const uint8_t* file, size_t len, unsigned scan_start, int display, int quiet)
{
    if (len >= (1ull << 32) - 64) { js_set_error("file too large for the 32-bit offsets of the reference format"); return -1; }
    if (js_prog_count(this)) { js_set_error("a batch holds either baseline or progressive files, not both"); return -1; }
    JsImage im; JsTableSet* ts = new JsTableSet;
    if (!js_describe_image(d, &im, ts, (uint32_t)len, scan_start, display, quiet, d->log_fn ? file : nullptr)) { delete ts; js_set_error("image not decodable (see log callback)"); return -1; }
    im.decode_ac = display ? (uint32_t)(d->batch == this ? d->opt_decode_ac : opt_decode_ac) : 0;
    // scan length: up to the first marker that is neither stuffing nor RSTn (what pass 1 of the SOS
    // handler skips over, source/JfifDecode.cpp:5207-5265); bytes past `len` read as zero.
    const uint32_t q = js_scan_end(file, scan_start, len);
    im.scan_len = q > scan_start ? q - scan_start : 0;
    // stage the file bytes (the CwindowBuf replacement: whole file in pinned host memory)
    const uint64_t off = align_up(raw_bytes, 16);
    if (reserve_pinned(off + len + 16)) { delete ts; return -1; }
    memset(pinned + raw_bytes, 0, off - raw_bytes);
    memcpy(pinned + off, file, len); memset(pinned + off + len, 0, 16);
    raw_bytes = off + len + 16; im.file_off = off;
    // de-duplicate the table set
    uint32_t tsi = (uint32_t)tables.size();
    for (uint32_t i = 0; i < tables.size(); i++) if (!memcmp(&tables[i], ts, sizeof *ts)) { tsi = i; break; }
    if (tsi == tables.size()) tables.push_back(*ts);
    delete ts;
    im.tableset = tsi;
    imgs.push_back(im); uploaded = false;
    JsImgHost hi; hi.dht_setmax[0] = d->t.dht_setmax[0]; hi.dht_setmax[1] = d->t.dht_setmax[1]; hi.err_max = d->opt_err_max; hi.display = display != 0;
    hinfo.resize(imgs.size() - 1); hinfo.push_back(hi);
    return (int)imgs.size() - 1;
}
// Stages a file whose image descriptor the caller built itself (progressive path): no scan-length search, no decode tables.
int JsnoopBatch::add_described(const JsImage& desc, const uint8_t* file, size_t len)
{
    if (len >= (1ull << 32) - 64) { js_set_error("file too large for the 32-bit offsets of the reference format"); return -1; }
    JsImage im = desc;
    const uint64_t off = align_up(raw_bytes, 16);
    if (reserve_pinned(off + len + 16)) return -1;
    memset(pinned + raw_bytes, 0, off - raw_bytes);
    memcpy(pinned + off, file, len); memset(pinned + off + len, 0, 16);
    raw_bytes = off + len + 16; im.file_off = off; im.file_len = (uint32_t)len;
    if (tables.empty()) { JsTableSet* ts = new JsTableSet; memset(ts, 0, sizeof *ts); tables.push_back(*ts); delete ts; }
    im.tableset = 0;
    imgs.push_back(im); uploaded = false;
    return (int)imgs.size() - 1;
}
int JsnoopBatch::add_clone(const JsnoopBatch* src, uint32_t i, bool through_markers)
{
    if (i >= src->imgs.size()) { js_set_error("add_clone: image index out of range"); return -1; }
    JsImage im = src->imgs[i];
    const JsTableSet& ts = src->tables[im.tableset];
    const uint8_t* file = src->pinned + im.file_off; const size_t len = im.file_len;
    if (through_markers) im.scan_len = im.scan_start < im.file_len ? im.file_len - im.scan_start : 0u;
    const uint64_t off = align_up(raw_bytes, 16);
    if (reserve_pinned(off + len + 16)) return -1;
    memset(pinned + raw_bytes, 0, off - raw_bytes);
    memcpy(pinned + off, file, len); memset(pinned + off + len, 0, 16);
    raw_bytes = off + len + 16; im.file_off = off;
    uint32_t tsi = (uint32_t)tables.size();
    for (uint32_t k = 0; k < tables.size(); k++) if (!memcmp(&tables[k], &ts, sizeof ts)) { tsi = k; break; }
    if (tsi == tables.size()) tables.push_back(ts);
    im.tableset = tsi;
    imgs.push_back(im); uploaded = false;
    hinfo.resize(imgs.size() - 1); hinfo.push_back((size_t)i < src->hinfo.size() ? src->hinfo[i] : JsImgHost());
    return (int)imgs.size() - 1;
}
int JsnoopBatch::tile(int total)
{
    // Replicates the batch PHYSICALLY up to `total` images: every copy gets its own file bytes in the pinned staging area
    // (and so in the raw arena), so that a "1024-image" batch really uploads and reads 1024 files' worth of bytes.
    const size_t n = imgs.size();
    if (!n) { js_set_error("tile: empty batch"); return -1; }
    for (size_t i = n; i < (size_t)total; i++) {
        JsImage im = imgs[i % n];
        const uint64_t off = align_up(raw_bytes, 16);
        if (reserve_pinned(off + im.file_len + 16)) return -1;
        memset(pinned + raw_bytes, 0, off - raw_bytes);
        memcpy(pinned + off, pinned + im.file_off, im.file_len); memset(pinned + off + im.file_len, 0, 16);
        raw_bytes = off + im.file_len + 16; im.file_off = off;
        imgs.push_back(im);
        if (hinfo.size() > i % n) { hinfo.resize(i); hinfo.push_back(hinfo[i % n]); }
        if (js_prog_count(this)) js_prog_dup(this, (uint32_t)(i % n), (uint32_t)i);
    }
    uploaded = false;
    return (int)imgs.size();
}
template <class T> static int grow(T** p, size_t* cap, size_t need_bytes)
{
    if (need_bytes <= *cap && *p) return 0;
    if (*p) { hipFree(*p); *p = nullptr; }
    size_t nb = need_bytes + need_bytes / 16 + 256;
    hipError_t e = hipMalloc((void**)p, nb);
    if (e != hipSuccess) { js_set_error("hipMalloc(%zu) failed: %s", nb, hipGetErrorString(e)); *cap = 0; return -1; }
    *cap = nb; return 0;
}
// Two halves on two streams (the second half a stage behind, decode()): from 8 MB of scan data on -- N x 1080p, ms per decode, one stream | two (tools/small_jobs.py,
// two alternating runs each, round 6): 4: 0.30 | 0.36, 8: 0.38 | 0.43, 16: 0.58 | 0.56, 24: 0.77 | 0.74, 32: 0.93 | 0.88, 40: 1.10 | 1.09, 48: 1.33 | 1.36, 64: 1.50 | 1.50,
// 96: 1.87 | 1.82, 128: 2.30 | 2.20, 160: 2.64 | 2.57, 200: 3.12 | 3.13 .. 3.33; 1024: 12.83 | 12.54.
#define JS_SPLIT_FROM_BYTES (8ull << 20)
int JsnoopBatch::upload()
{
    JsRange r_("jsnoop:upload (pinned H2D + descriptors)");
    HIP_TRY(hipSetDevice(device));
    const size_t n = imgs.size();
    if (!n) { js_set_error("upload: empty batch"); return -1; }
    uint64_t blocks = 0, dibb = 0, plane = 0, side = 0, ustr = 0, subs = 0;
    std::vector<uint32_t> wg(n + 1), usb(2 * (n + 1)), syb(2 * (n + 1));       // syb: write-pass bases, then sync-pass bases; usb: 4 KiB chunks, then super-chunks of four (k_unstuff_fused)
    uint64_t segw = 0, mcub = 0, recw = 0; uint32_t usc = 0, us4 = 0, syw = 0, snw = 0;
    strips_per_wg = 0; uint64_t total_mcus = 0; for (const JsImage& im : imgs) total_mcus += (uint64_t)im.mcu_xmax * im.mcu_ymax;
    // back end: 8 waves per workgroup; enough MCUs per wave to amortise a workgroup's table load; a small job as ONE round of workgroups over the
    // chip's 1024 workgroup slots (one 3840x2160 image: 4 MCUs per wave, 1013 workgroups, 47 us; 3 per wave = 1350 workgroups: 51)
    uint32_t mcus_per_wave = (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(1, (total_mcus + 8 * 1024 - 1) / (8 * 1024)));
    if (tune.mcus_per_wave > 0) mcus_per_wave = (uint32_t)tune.mcus_per_wave;
    // sub-sequence length: long (512 B) when the batch still yields plenty of lanes, short (128 B, 64 B) for small jobs
    uint64_t scan_total = 0; for (const JsImage& im : imgs) scan_total += im.scan_len;
    sub_wl = scan_total >= (96ull << 20) ? 7 : (scan_total < (4ull << 20) ? 4 : (scan_total < (64ull << 20) ? 5 : 6));    // (a single image / a handful: 64-byte pieces give the write pass more lanes; 256-byte pieces between the candidate form and the large batches: 128 x 1080p 2.45 against 2.70 ms with 128-byte pieces)
    // Candidate synchronisation (k_cand_*) wants 64-byte pieces and one walk per piece and block of the MCU.  It beats the rounds of k_sync far beyond
    // what the chip holds at once (~500 k lanes): N x 1080p 4:2:0, ms per decode, candidates | rounds: 1: 0.30 | 0.80, 4: 0.37 | 0.83, 8: 0.45 | 0.95,
    // 16: 0.64 | 1.09, 32: 1.04 | 1.31, 48: 1.44 | 1.58 (2.6 M walks); the two meet near 64 images.
    uint32_t max_blk = 0; for (const JsImage& im : imgs) max_blk = std::max(max_blk, im.blk_per_mcu);
    // (round 6, with the list rounds behind k_sync and 128-byte pieces, N x 1080p candidates | rounds: 24: 0.80 | 0.98, 32: 0.95 | 1.04, 48: 1.29 | 1.25 (2.6 M walks),
    //  64: 1.56 | 1.43, 96: 1.81 (256-byte pieces, plain launches) | 1.76, 128: 2.25 | 2.23 -- profiles/r06_experiments.txt 18)
    const uint64_t cand_lanes = tune.cand_max_walks ? tune.cand_max_walks : 2600000;
    const int cand_want = tune.cand_rounds == 0 ? 16 : tune.cand_rounds;
    const bool cand_fits = cand_want >= 0 && max_blk >= 1 && max_blk <= JS_CAND_MAX_BLK && (scan_total / 64 + 64 * n) * max_blk <= cand_lanes;
    if (cand_fits) sub_wl = 4;
    if (tune.sub_wl) sub_wl = tune.sub_wl;
    sync_launches = tune.sync_launches > 0 ? tune.sync_launches : 2;
    // Synchronisation by rounds = k_sync cut after two rounds, then list rounds over the whole job (js_launch_sync_rounds), unless the tuning struct asks for a
    // number of plain k_sync launches.
    sync_rounds = (tune.sync_launches == 0 && !(cand_fits && sub_wl == 4)) ? JS_SYR_ROUNDS : 0;
    const uint32_t sub_bytes = 4u << sub_wl;
    uint32_t wgs = 0; max_mcu_h = 8; max_mcu_w = 8;
    for (size_t i = 0; i < n; i++) {
        JsImage& im = imgs[i];
        im.want_planes = (uint32_t)opt_want_planes;
        im.coef_off = blocks; blocks += im.total_blocks;
        im.dib_off = dibb; dibb += align_up((uint64_t)im.img_x * im.img_y * 4, 256);
        im.plane_off = plane; if (opt_want_planes) plane += align_up((uint64_t)im.blk_xmax * 8 * im.blk_ymax * 8 * 3, 64);
        im.side_off = side; side += align_up(js_side_words(im.mcu_xmax * im.mcu_ymax, im.blk_xmax * im.blk_ymax), 4);
        im.ustr_off = ustr; im.ustr_cap = (uint32_t)align_up((uint64_t)im.scan_len + 64, 64ull * sub_bytes); ustr += im.ustr_cap;   // whole 64-sub-sequence groups
        im.n_subseq = (im.ustr_cap + sub_bytes - 1) / sub_bytes; im.subseq_off = subs; subs += align_up(im.n_subseq, 256);
        const uint32_t nmcu = im.mcu_xmax * im.mcu_ymax;
        const uint64_t want_seg = im.rst_interval ? (uint64_t)nmcu / im.rst_interval + 2 : 1;
        im.seg_cap = (uint32_t)std::min<uint64_t>((1u << 20) - 1, want_seg * 2 + 16); im.seg_off = segw; segw += align_up(im.seg_cap, 4);   // 20-bit interval index in the state word
        im.mcu_off = mcub; mcub += align_up(nmcu, 16);
        im.rec_off = (uint32_t)recw; recw += align_up((uint64_t)nmcu + 2, 16);
        im.ev_cap = opt_events ? JS_EV_MAX : 0; im.ev_off = (uint64_t)i * (1 + JS_EV_WORDS * JS_EV_MAX);
        { const uint32_t nck = std::max(1u, (uint32_t)(((im.scan_start & 15) + (uint64_t)im.scan_len + JS_US_CHUNK - 1) / JS_US_CHUNK));   // (at least one chunk: its last chunk leaves the image's totals)
          usb[i] = usc; usc += nck; usb[n + 1 + i] = us4; us4 += (nck + 3) / 4; }
        syb[i] = syw; syw += (im.n_subseq + JS_SY_THREADS - 1) / JS_SY_THREADS;
        syb[n + 1 + i] = snw; snw += (im.n_subseq + JS_SY_THREADS - JS_SY_HALO - 1) / (JS_SY_THREADS - JS_SY_HALO);   // sync pass: two threads per workgroup walk a halo
        max_mcu_h = std::max(max_mcu_h, im.mcu_h); max_mcu_w = std::max(max_mcu_w, im.mcu_w);
        wg[i] = wgs; wgs += std::max(1u, (nmcu + 8 * mcus_per_wave - 1) / (8 * mcus_per_wave));     // 8 waves per workgroup, one MCU per wave at a time
    }
    usb[n] = usc; usb[2 * n + 1] = us4; syb[n] = syw; syb[2 * n + 1] = snw; us_chunks = usc; sy_wgs = syw; sn_wgs = snw; seg_words = segw; mcu_bytes = mcub;
    wg[n] = wgs; total_wgs = wgs; rec_words = recw; total_blocks = blocks; dib_bytes = dibb; side_words = side; total_subseq = subs; ustr_bytes = ustr;
    if (grow(&dev.raw, &cap.raw, raw_bytes + 64) || grow(&dev.coef, &cap.coef, blocks * 128) || grow(&dev.dccum, &cap.dccum, blocks * 2 + 64) ||
        grow(&dev.dib, &cap.dib, dibb) || grow(&dev.side, &cap.side, side * 4) || grow(&dev.imgs, &cap.imgs, n * sizeof(JsImage)) ||
        grow(&dev.tables, &cap.tables, tables.size() * sizeof(JsTableSet)) || grow(&dev.wg_base, &cap.wg_base, (n + 1) * 4) ||
        grow(&dev.sel, &cap.sel, n * 4) || grow(&dev.sums, &cap.sums, n * 8) || grow(&dev.ustr, &cap.ustr, ustr + 64) ||
        grow(&dev.sub, &cap.sub, subs * 24 + 64 + js_sync_list_words(subs, (uint32_t)n) * 4) || grow(&dev.probe, &cap.probe, 1024) || grow(&dev.seg, &cap.seg, segw * 4 + 64) ||
        grow(&dev.chunk_keep, &cap.chunk_keep, (size_t)usc * 4 + 64) || grow(&dev.chunk_rst, &cap.chunk_rst, (size_t)usc * 4 + 64) ||
        grow(&dev.us_base, &cap.us_base, 2 * (n + 1) * 4) || grow(&dev.sy_base, &cap.sy_base, 2 * (n + 1) * 4) ||
        grow(&dev.mcu_rst, &cap.mcu_rst, mcub + 64) || grow(&dev.dc_parts, &cap.dc_parts, JS_DC_PARTS_BYTES) || grow(&dev.ustr_lin, &cap.ustr_lin, ustr + 64) || grow(&dev.flags, &cap.flags, n * 8 + 64) ||
        grow(&dev.us_state, &cap.us_state, (size_t)usc * 8 + 64)) return -1;
    HIP_TRY(hipMemsetAsync(dev.us_state, 0, (size_t)usc * 8 + 64, stream)); us_epoch = 0; us_ticket_base[0] = us_ticket_base[1] = 0;   // (the ticket counters sit behind the state words)      // (no word of an earlier layout may look current)
    if (opt_want_planes && grow(&dev.planes, &cap.planes, plane * 2)) return -1;
    { uint32_t most = 0; for (size_t i = 0; i < n; i++) most = std::max(most, wg[i + 1] - wg[i]); if (most > 64 && grow(&dev.wg_part, &cap.wg_part, (size_t)wgs * 16 + 64)) return -1; }
    cand_blk = max_blk; cand_rounds = (cand_fits && sub_wl == 4) ? cand_want : -1;
    // two write lanes per sub-sequence while the job leaves SIMDs idle anyway (one 3840x2160 image: write pass 88 -> 66 us; sixteen 1080p images: 111 -> 123)
    cand_half = cand_rounds >= 0 && tune.write_lanes != 1 && (subs <= 40960 || tune.write_lanes == 2);
    if (cand_rounds >= 0 && (grow(&dev.cand, &cap.cand, js_cand_bytes(subs)) || grow(&dev.cand_req, &cap.cand_req, n * JS_CAND_REQ_WORDS * 4))) return -1;
    event_words = opt_events ? (uint64_t)n * (1 + JS_EV_WORDS * JS_EV_MAX) : 0;
    if (event_words && grow(&dev.events, &cap.events, event_words * 4)) return -1;
    tab_rows = 1; tab_lut2 = 0; uint32_t tdc = 1, tac = 1;
    for (const JsTableSet& t : tables) { tab_rows = std::max(tab_rows, t.n_rows); tab_lut2 = std::max(tab_lut2, t.lut2_used); tdc = std::max(tdc, t.n_dc_rows); tac = std::max(tac, t.n_ac_rows); }
    tab_rows_w = tdc | (tac << 8);                                 // write pass: DC rows (16-bit entries) and AC rows (32-bit pair entries) separately
    HIP_TRY(hipMemcpyAsync(dev.raw, pinned, raw_bytes, hipMemcpyHostToDevice, stream));
    {   // The descriptors go through ONE page-locked staging block of this batch: copies out of pageable vectors are staged by the runtime one by one, and the wait
        // that kept those vectors alive was a round trip of its own (a single-image call: upload 75 -> ~35 us).  The block is rewritten by the next upload only:
        // that one waits for this one's copies first (an event; long done in any real sequence of calls).
        const size_t sz[5] = { n * sizeof(JsImage), tables.size() * sizeof(JsTableSet), (n + 1) * 4, 2 * (n + 1) * 4, 2 * (n + 1) * 4 };
        const void* src[5] = { imgs.data(), tables.data(), wg.data(), usb.data(), syb.data() };
        void* dst[5] = { dev.imgs, dev.tables, dev.wg_base, dev.us_base, dev.sy_base };
        size_t off[5], total = 0;
        for (int k = 0; k < 5; k++) { off[k] = total; total += (sz[k] + 255) & ~(size_t)255; }
        if (!ev_up) HIP_TRY(hipEventCreateWithFlags(&ev_up, hipEventDisableTiming));
        else HIP_TRY(hipEventSynchronize(ev_up));
        if (total > h_desc_cap) {
            if (h_desc) hipHostFree(h_desc);
            h_desc = nullptr; h_desc_cap = 0;
            HIP_TRY(hipHostMalloc((void**)&h_desc, total + total / 4 + 4096, hipHostMallocDefault));
            h_desc_cap = total + total / 4 + 4096;
        }
        for (int k = 0; k < 5; k++) { memcpy(h_desc + off[k], src[k], sz[k]); HIP_TRY(hipMemcpyAsync(dst[k], h_desc + off[k], sz[k], hipMemcpyHostToDevice, stream)); }
        HIP_TRY(hipEventRecord(ev_up, stream));
    }
    h_us_base.assign(usb.begin(), usb.begin() + n + 1); h_us4_base.assign(usb.begin() + n + 1, usb.end()); h_sy_base.assign(syb.begin(), syb.begin() + n + 1); h_sn_base.assign(syb.begin() + n + 1, syb.end()); h_wg_base = wg;
    // two halves on two streams: by default from JS_SPLIT_FROM_BYTES of scan data on (above)
    split_parts = (n >= 2 && (tune.split == 2 || (tune.split == 0 && scan_total >= JS_SPLIT_FROM_BYTES))) ? 2 : 1;
    uploaded = true;
    return 0;
}
static const char* kStageName[JSNOOP_NUM_STAGES] = { "clear", "unstuff", "sync", "blockscan", "write", "dcscan", "exact", "idct_color" };

int JsnoopBatch::decode(bool timed)
{
    HIP_TRY(hipSetDevice(device));
    if (!uploaded && upload()) return -1;
    const uint32_t n = (uint32_t)imgs.size();
    side_done.assign(n, 0); side_mode.assign(n, 0); side_anoms.assign(n, std::vector<uint32_t>()); side_chunk_ok.assign(n, 0); side_events.assign(n, std::vector<uint32_t>()); side_pre.assign(n, 0); side_requests = 0;   // (nothing of an earlier decode's side pass survives)
    JsRange r_("jsnoop:decode (enqueue)");
    if (js_prog_count(this)) return decode_progressive(timed);     // SOF2 files: every scan of every image, one launch per dependency level
    if (timed) HIP_TRY(hipEventRecord(ev[0], stream));
    us_epoch = us_epoch % 255u + 1u;                               // both halves of a split decode share it: their chunks are disjoint
    bool parallel_ok = !opt_force_exact; if (parallel_ok) { parallel_ok = false; for (const JsTableSet& t : tables) parallel_ok = parallel_ok || t.lut_ok; }
    if (!parallel_ok) {           // the exact-mirror kernel stores only what it decodes; the parallel path writes every block whole
        HIP_TRY(hipMemsetAsync(dev.coef, 0, total_blocks * 128, stream));
        HIP_TRY(hipMemsetAsync(dev.dccum, 0, total_blocks * 2, stream));
    }
    // (one launch; all three arenas are allocated with 64 bytes of slack.  The parallel path's decode touches the status words of the side blocks only:
    //  histogram and maps are the side pass's, which clears them itself -- 0.33 MB per 1080p image that need not be written per decode)
    js_launch_clear3(stream, dev.side, side_words * 4, dev.mcu_rst, mcu_bytes, dev.flags, (size_t)n * 8, parallel_ok ? dev.imgs : nullptr, n);
    if (event_words) HIP_TRY(hipMemsetAsync(dev.events, 0, event_words * 4, stream));
    if (timed) HIP_TRY(hipEventRecord(ev[1], stream));
    last_timed_split = false;
    if (split_parts == 2 && parallel_ok && !event_words) {
        // two halves of the batch on two streams, launches interleaved stage by stage
        if (ensure_aux()) return -1;
        const uint32_t n0 = n / 2; hipStream_t s2 = aux[0];
        HIP_TRY(hipEventRecord(aux_ev[0], stream)); HIP_TRY(hipStreamWaitEvent(s2, aux_ev[0], 0));
        // The second half starts when the first one has un-stuffed (round 6): side by side from the first launch on, both halves reach the thinly filled list
        // rounds of their synchronisation at the same time and the chip idles with both; half a stage apart, each one's run under the other's full kernels
        // (12.63 -> 12.40 ms, three alternating pairs of one call; starting it behind the first half's synchronisation: 13.3).
        if (js_parallel_entropy_part(this, stream, 0, n0, timed ? ev : nullptr, aux_ev[2], 1) < 0) return -1;
        HIP_TRY(hipStreamWaitEvent(s2, aux_ev[2], 0));
        if (timed) { HIP_TRY(hipEventRecord(ev2[0], s2)); HIP_TRY(hipEventRecord(ev2[1], s2)); }
        if (js_parallel_entropy_part(this, s2, n0, n - n0, timed ? ev2 : nullptr) < 0) return -1;
        last_used_parallel = true;
        if (timed) { HIP_TRY(hipEventRecord(ev[7], stream)); HIP_TRY(hipEventRecord(ev2[7], s2)); }
        { JsRange r2_("jsnoop:idct+colour"); if (launch_back_end_part(stream, 0, n0) || launch_back_end_part(s2, n0, n - n0)) return -1; }
        if (timed) { HIP_TRY(hipEventRecord(ev[8], stream)); HIP_TRY(hipEventRecord(ev2[8], s2)); last_timed_split = true; }
        HIP_TRY(hipEventRecord(aux_ev[1], s2)); HIP_TRY(hipStreamWaitEvent(stream, aux_ev[1], 0));
        HIP_TRY(hipGetLastError());
        return 0;
    }
    // parallel path stages 1..5 (k_unstuff .. k_dc_scan) are launched by js_parallel_entropy
    int used_parallel = opt_force_exact ? 0 : js_parallel_entropy(this, timed);
    if (used_parallel < 0) return -1;
    last_used_parallel = used_parallel != 0;
    if (timed && !used_parallel) for (int s = 2; s <= 6; s++) HIP_TRY(hipEventRecord(ev[s], stream));
    if (!used_parallel) js_launch_entropy_exact(stream, dev.imgs, nullptr, n, dev.tables, dev.raw, dev.coef, dev.dccum, dev.side, 0, event_words ? dev.events : nullptr);
    if (timed) HIP_TRY(hipEventRecord(ev[7], stream));
    { JsRange r2_("jsnoop:idct+colour"); if (launch_back_end(n)) return -1; }
    if (timed) HIP_TRY(hipEventRecord(ev[8], stream));
    HIP_TRY(hipGetLastError());
    return 0;
}
int JsnoopBatch::launch_back_end(uint32_t nimg) { return launch_back_end_part(stream, 0, nimg); }
int JsnoopBatch::launch_back_end_part(hipStream_t st, uint32_t i0, uint32_t nimg)
{
    uint32_t tile = 16; int layout = -1;                          // layout: the one fast layout all images of the launch share, else 0
    for (uint32_t i = i0; i < i0 + nimg && i < imgs.size(); i++) {
        tile = std::max(tile, js_tile_bytes(imgs[i]));
        const JsImage& im = imgs[i];
        const int l = !js_fast_layout(im) ? 0 : (im.expand_h[2] == 2 ? (im.expand_v[2] == 2 ? 1 : 2) : (im.expand_v[2] == 2 ? 3 : 4));
        layout = layout < 0 ? l : (layout == l ? l : 0);
    }
    if (layout < 0 || (tune.cross_checks & JSNOOP_XC_BACKEND_GENERIC)) layout = 0;   // (cross-check: the all-layouts kernel for every launch)
    const uint32_t wgs = h_wg_base.size() > i0 + nimg ? h_wg_base[i0 + nimg] - h_wg_base[i0] : total_wgs;
    // an image spread over many workgroups: per-workgroup status records and a fold, instead of every workgroup queueing on the image's two status words
    uint32_t most = 0; for (uint32_t i = i0; i < i0 + nimg && i + 1 < h_wg_base.size(); i++) most = std::max(most, h_wg_base[i + 1] - h_wg_base[i]);
    unsigned long long* part = nullptr;
    if (most > 64) part = dev.wg_part;                            // (allocated by upload() when some image has that many)
    const int rc = js_launch_idct_color(st, dev.imgs + i0, dev.wg_base + i0, nimg, wgs, tile, d_lut, dev.coef, dev.dccum, dev.dib, dev.planes, dev.side, layout, part);
    if (rc == -2) { js_set_error("back end: an MCU tile of %u bytes per wave does not fit the 160 KiB LDS", tile); return -1; }
    if (rc) { js_set_error("back end launch failed (%d): %s", rc, hipGetErrorString(hipGetLastError())); return -1; }
    return 0;
}
int JsnoopBatch::sync()
{
    JsRange r_("jsnoop:sync (wait + exact-path fix-up)");
    HIP_TRY(hipSetDevice(device));
    if (js_prog_count(this)) return sync_progressive();
    HIP_TRY(hipStreamSynchronize(stream));
    return js_parallel_fixup(this);     // re-decodes flagged images on the exact path (no-op when none)
}

// ------------------------------------------------------------------------------ C ABI
extern "C" {

int jsnoop_abi_version(void) { return JSNOOP_ABI_VERSION; }
int jsnoop_selftest_tables(unsigned seed, unsigned rounds) { return js_selftest_tables(seed, rounds); }
int jsnoop_selftest_bytes(unsigned seed, unsigned rounds)
{
    // the sixteen-bytes-per-step marker searches of the staging code against byte-at-a-time loops, on buffers dense in FF / 00 / RSTn / other markers
    int bad = 0; uint32_t x = seed * 2654435761u + 12345u;
    auto rnd = [&x]() { x = x * 1664525u + 1013904223u; return x >> 8; };
    for (unsigned r = 0; r < rounds; r++) {
        const size_t n = rnd() % 300, lead = rnd() % 17;
        std::vector<uint8_t> buf(lead + n + 1);
        for (size_t i = 0; i < lead + n; i++) { const uint32_t k = rnd() % 16; buf[i] = k < 5 ? 0xFF : (k < 7 ? 0x00 : (k < 9 ? (uint8_t)(0xD0 + rnd() % 8) : (k == 9 ? 0xD9 : (uint8_t)rnd()))); }
        const uint8_t* f = buf.data() + lead;                     // (every alignment of the first byte)
        for (size_t q0 = 0; q0 <= n; q0 += 1 + rnd() % 7) {
            size_t a = q0; while (a < n && f[a] != 0xFF) a++;
            if (js_next_ff(f, q0, n) != a) bad++;
            uint32_t q = (uint32_t)q0;
            while ((size_t)q + 1 < n) { if (f[q] == 0xFF && f[q + 1] != 0 && !(f[q + 1] >= 0xD0 && f[q + 1] <= 0xD7)) break; q++; }
            if ((size_t)q + 1 >= n) q = (uint32_t)n;
            if (js_scan_end(f, (uint32_t)q0, n) != (q0 >= n ? (uint32_t)n : q)) bad++;
        }
    }
    return bad;
}
const char* jsnoop_last_error(void) { return g_err.c_str(); }
int jsnoop_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
int jsnoop_set_device(int device)
{
    int n = jsnoop_device_count();
    if (device < 0 || device >= n) { js_set_error("device %d not available (%d visible)", device, n); return -1; }
    HIP_TRY(hipSetDevice(device)); g_device = device; return 0;
}

JsnoopDecoder* jsnoop_create(void)
{
    if (jsnoop_device_count() <= 0) { js_set_error("no HIP device visible: libjsnoop_gpu has no CPU fallback"); return nullptr; }
    JsnoopDecoder* d = new JsnoopDecoder();
    d->batch = new JsnoopBatch(nullptr);
    if (d->batch->init()) { delete d->batch; delete d; return nullptr; }
    d->batch->opt_want_planes = 1;        // the reference always keeps m_pPixValY/Cb/Cr
    d->batch->opt_events = 1;             // ... and writes its messages to CDocLog as it goes
    return d;
}
void jsnoop_destroy(JsnoopDecoder* d) { if (!d) return; delete d->batch; delete d; }
void jsnoop_reset(JsnoopDecoder* d)                                             // Reset :49-138
{
    d->have_image = false; d->host_valid = 0; memset(d->geom, 0, sizeof d->geom); d->geom[0] = d->geom[1] = 1; d->warn_ycc_clip = 0;
    if (d->dib_temp_ready) { d->dib_temp.clear(); d->dib_temp.shrink_to_fit(); d->dib_temp_ready = false; }          // :80-83
}
void jsnoop_reset_state(JsnoopDecoder* d) { d->reset_state(); }
void jsnoop_reset_dqt_tables(JsnoopDecoder* d) { d->reset_dqt_tables(); }
void jsnoop_reset_dht_lookup(JsnoopDecoder* d) { d->reset_dht_lookup(); }
void jsnoop_set_image_dimensions(JsnoopDecoder* d, unsigned w, unsigned h) { d->base_w = w; d->base_h = h; }            // :2706-2709
void jsnoop_get_image_dimensions(JsnoopDecoder* d, unsigned* w, unsigned* h) { *w = d->base_w; *h = d->base_h; }
uint8_t* jsnoop_dib_temp_create(JsnoopDecoder* d, unsigned w, unsigned h)
{
    try { d->dib_temp.assign((size_t)w * h * 4, 0); }                                       // (no exception crosses the C ABI: CDIB::CreateDIB returns false, Dib.cpp:53)
    catch (const std::exception&) { d->dib_temp.clear(); d->dib_temp.shrink_to_fit(); js_set_error("jsnoop_dib_temp_create: no memory for %u x %u pixels", w, h); return nullptr; }
    d->have_image = false; d->host_valid = 0;                                               // (the decoded image's DIB is no longer the preview)
    return d->dib_temp.empty() ? nullptr : d->dib_temp.data();
}
void jsnoop_set_dib_temp_ready(JsnoopDecoder* d, int ready) { d->dib_temp_ready = ready != 0; }
int  jsnoop_get_dib_temp_ready(JsnoopDecoder* d) { return d->dib_temp_ready || d->have_image; }
void jsnoop_set_preview_is_jpeg(JsnoopDecoder* d, int is_jpeg) { d->preview_is_jpeg = is_jpeg != 0; }
void jsnoop_set_log_callback(JsnoopDecoder* d, jsnoop_log_fn fn, void* user) { d->log_fn = fn; d->log_user = user; }
void jsnoop_set_options(JsnoopDecoder* d, int ac, int histo, int clip, unsigned err_max)
{ d->opt_decode_ac = ac; d->opt_histo_en = histo; d->opt_stat_clip_en = clip; d->opt_err_max = err_max; }

int jsnoop_set_dqt_entry(JsnoopDecoder* d, unsigned tbl, unsigned nat, unsigned zz, unsigned val)      // :424-453
{
    if (tbl < 4 && nat < 64) { d->t.dqt_nat[tbl][nat] = (uint16_t)val; d->t.dqt_zz[tbl][zz & 63] = (uint16_t)val; return 1; }
    d->log(2, "ERROR: Attempt to set DQT entry out of range (nTblDestId=%u, nCoeffInd=%u, nCoeffVal=%u)", tbl, nat, val); return 0;
}
unsigned jsnoop_get_dqt_entry(JsnoopDecoder* d, unsigned tbl, unsigned nat)                            // :466-490
{
    if (tbl < 4 && nat < 64) return d->t.dqt_nat[tbl][nat];
    d->log(2, "ERROR: GetDqtEntry(nTblDestId=%u, nCoeffInd=%u) out of indexed range", tbl, nat); return 0;
}
int jsnoop_set_dqt_tables(JsnoopDecoder* d, unsigned comp, unsigned tbl)                               // :505-520
{
    if (comp < 256 && tbl < 4) { d->t.dqt_sel[comp] = (int)tbl; return 1; }
    d->log(2, "ERROR: SetDqtTables(Comp ID=%u, Table=%u) out of indexed range", comp, tbl); return 0;
}
int jsnoop_set_dht_tables(JsnoopDecoder* d, unsigned comp, unsigned dc, unsigned ac)                   // :536-553
{
    if (comp >= 1 && comp < 5 && dc < 4 && ac < 4) { d->t.dht_sel[0][comp] = (int)dc; d->t.dht_sel[1][comp] = (int)ac; return 1; }
    d->log(2, "ERROR: SetDhtTables(comp=%u, TblDC=%u TblAC=%u) out of indexed range", comp, dc, ac); return 0;
}
int jsnoop_set_dht_entry(JsnoopDecoder* d, unsigned dest, unsigned cls, unsigned ind, unsigned len,
                         unsigned bits, unsigned mask, unsigned code)                                  // :748-820
{
    if (dest >= 4 || cls >= 2 || ind >= JS_DHT_CODES) { d->log(2, "ERROR: Attempt to set DHT entry out of range"); return 0; }
    JsTables& t = d->t;
    t.dht_bitlen[cls][dest][ind] = len; t.dht_bits[cls][dest][ind] = bits; t.dht_mask[cls][dest][ind] = mask; t.dht_code[cls][dest][ind] = code;
    if (dest > t.dht_setmax[cls]) t.dht_setmax[cls] = dest;
    if (len <= JS_FAST_BITS) {
        unsigned lo = (bits & mask) >> (32 - JS_FAST_BITS), hi = lo + ((1u << (JS_FAST_BITS - len)) - 1);
        for (unsigned i = lo; i <= hi && i < 1024; i++) t.dht_fast[cls][dest][i] = code + (len << 8);
    }
    return 1;
}
int jsnoop_set_dht_size(JsnoopDecoder* d, unsigned dest, unsigned cls, unsigned n)                     // :834-847
{
    if (dest >= 4 || cls >= 2 || n >= JS_DHT_CODES) { d->log(2, "ERROR: Attempt to set DHT table size out of range"); return 0; }
    d->t.dht_size[cls][dest] = n; return 1;
}
void jsnoop_set_sof_samp_factors(JsnoopDecoder* d, unsigned comp, unsigned h, unsigned v) { if (comp < 256) { d->t.samp_h[comp] = h; d->t.samp_v[comp] = v; } }
void jsnoop_set_precision(JsnoopDecoder* d, unsigned p) { d->t.precision = p; }
void jsnoop_set_image_details(JsnoopDecoder* d, unsigned x, unsigned y, unsigned nf, unsigned ns, int rst_en, unsigned rst_int)
{ JsTables& t = d->t; t.details_set = 1; t.dim_x = x; t.dim_y = y; t.num_sof = nf; t.num_sos = ns; t.rst_en = rst_en != 0; t.rst_interval = rst_int; }

int jsnoop_jfif_walk(JsnoopDecoder* d, const uint8_t* file, size_t len, unsigned* scan_start) { return js_jfif_walk(d, file, len, scan_start); }

void jsnoop_pixel_to_mcu(JsnoopDecoder* d, unsigned px, unsigned py, unsigned* mx, unsigned* my)        // :5056-5062
{ *mx = d->geom[0] ? px / d->geom[0] : 0; *my = d->geom[1] ? py / d->geom[1] : 0; }
void jsnoop_pixel_to_blk(JsnoopDecoder*, unsigned px, unsigned py, unsigned* bx, unsigned* by) { *bx = px / 8; *by = py / 8; }   // :5071-5077 (BLK_SZ_X/Y = 8)
unsigned jsnoop_mcu_xy_to_linear(JsnoopDecoder* d, unsigned mx, unsigned my) { return my * d->geom[2] + mx; }                   // :5088-5093
void jsnoop_set_dump_histo_y(JsnoopDecoder* d, int on) { d->opt_dump_histo_y = on != 0; }

int jsnoop_overlay_install(JsnoopDecoder* d, const uint8_t* data, unsigned len, unsigned begin)          // OverlayInstall :516-556
{
    if (d->overlays.size() >= 500) return 0;                     // NUM_OVERLAYS (OverlayAlloc fails)
    if (len >= 500) { d->log(2, "ERROR: CwindowBuf:OverlayInstall() overlay too large"); return 0; }   // MAX_OVERLAY
    JsnoopDecoder::Overlay o; o.data.assign(data, data + len); o.begin = begin;
    d->overlays.push_back(std::move(o)); return 1;
}
void jsnoop_overlay_remove_all(JsnoopDecoder* d) { d->overlays.clear(); }                                 // :585-593
unsigned jsnoop_overlay_get_num(JsnoopDecoder* d) { return (unsigned)d->overlays.size(); }               // :618-621
int jsnoop_overlay_get(JsnoopDecoder* d, unsigned ind, const uint8_t** data, unsigned* len, unsigned* begin)   // :607-617
{
    if (ind >= d->overlays.size()) return 0;
    *data = d->overlays[ind].data.data(); *len = (unsigned)d->overlays[ind].data.size(); *begin = d->overlays[ind].begin; return 1;
}

void jsnoop_decode_scan_img(JsnoopDecoder* d, const uint8_t* file, size_t len, unsigned start, int display, int quiet)
{
    // CwindowBuf::Buf :639-660: an enabled overlay that covers an offset replaces the file's byte there, the last installed one
    // winning -- applied to a private copy of the file image, in installation order
    std::vector<uint8_t> patched;
    if (!d->overlays.empty()) {
        patched.assign(file, file + len);
        for (const JsnoopDecoder::Overlay& o : d->overlays)
            for (size_t i = 0; i < o.data.size(); i++) if ((size_t)o.begin + i < len) patched[(size_t)o.begin + i] = o.data[i];
        file = patched.data();
    }
    d->hist_latched = d->opt_histo_en != 0; d->clip_latched = d->opt_stat_clip_en != 0;      // :2740-2741
    jsnoop_reset(d);
    JsnoopBatch* b = d->batch;
    b->clear();
    b->opt_decode_ac = d->opt_decode_ac;
    d->last_path = 0; d->last_flags = 0;
    const bool dbg_t = (b->tune.debug & JSNOOP_DBG_TIMING) != 0;   // where a call's wall time goes (stderr, one line per call)
    auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tp[10]; int ntp = 0; if (dbg_t) tp[ntp++] = now_us();
    if (b->add(d, file, len, start, display, quiet) < 0) return;   // early returns of DecodeScanImg: no preview
    d->preview_is_jpeg = false; d->dib_temp.clear(); d->dib_temp_ready = false;        // :2976-2978
    d->base_w = d->geom[6]; d->base_h = d->geom[7];                 // m_rectImgBase :2874
    if (display) memset(d->stats, 0, sizeof d->stats);              // :3145-3155
    if (dbg_t) tp[ntp++] = now_us();
    if (b->upload()) { d->log(2, "*** ERROR: device decode failed: %s", g_err.c_str()); return; }
    if (dbg_t) tp[ntp++] = now_us();
    if (d->log_fn) (void)js_side_prepare(b, 0);                     // (the report will ask for the side outputs: the decode's write pass records for them)
    if (b->decode(false)) { b->rec_pos = nullptr; d->log(2, "*** ERROR: device decode failed: %s", g_err.c_str()); return; }
    if (d->log_fn) (void)js_side_prelaunch(b, 0); else b->rec_pos = nullptr;                  // the report will ask for the side outputs: their pass goes behind the decode now, not behind the wait
    if (dbg_t) tp[ntp++] = now_us();
    if (b->sync()) { d->log(2, "*** ERROR: device decode failed: %s", g_err.c_str()); return; }
    if (dbg_t) tp[ntp++] = now_us();
    d->have_image = true; d->host_valid = 0;
    if (display) d->preview_is_jpeg = true;
    d->last_path = (int)b->host_path[0]; d->last_flags = b->host_flags[0];
    d->side_ready = d->last_path == 2;          // the exact-mirror kernel fills the side block as it goes
    // the side block comes back once: with a log callback the report below asks for the side outputs anyway (side pass, then the read-back)
    if (d->log_fn && !d->side_ready) { d->ensure_side(); if (!d->side_ready) d->fetch_side(); } else d->fetch_side();
    if (dbg_t) tp[ntp++] = now_us();
    d->pending_log.clear();
    if (display) d->stats_pass();
    if (dbg_t) tp[ntp++] = now_us();
    if (d->log_fn) {                        // the reference's log text (messages of the decode loop, then the report)
        d->ensure_side();
        js_emit_decode_events(d);
        if (!quiet) d->log(0, "");                                  // :3630-3632
        d->flush_pending_log();                                     // CalcChannelPreview's warnings (:3643)
        js_emit_report(d, display != 0, quiet != 0);
    }
    if (dbg_t) { tp[ntp++] = now_us(); fprintf(stderr, "[timing] add %.0f upload %.0f enqueue %.0f wait+fixup %.0f side outputs %.0f statistics %.0f messages+report %.0f us\n", tp[1] - tp[0], tp[2] - tp[1], tp[3] - tp[2], tp[4] - tp[3], tp[5] - tp[4], tp[6] - tp[5], tp[7] - tp[6]); }
}

int  jsnoop_is_preview_ready(JsnoopDecoder* d) { return d->preview_is_jpeg; }
void jsnoop_get_image_size(JsnoopDecoder* d, unsigned* x, unsigned* y) { *x = d->geom[6]; *y = d->geom[7]; }
void jsnoop_get_geometry(JsnoopDecoder* d, unsigned* o) { memcpy(o, d->geom, sizeof d->geom); }
const uint8_t* jsnoop_get_bitmap_ptr(JsnoopDecoder* d)
{
    if (!d->have_image) return d->dib_temp.empty() ? nullptr : d->dib_temp.data();     // (m_pDibTemp as the PSD path left it, if it did)
    if (!(d->host_valid & 1)) {
        const JsImage& im = d->batch->imgs[d->img];
        if (d->h_dib.ensure((size_t)im.img_x * im.img_y * 4) || d->batch->read_dib(d->img, (uint8_t*)d->h_dib.p)) return nullptr;
        d->host_valid |= 1;
    }
    return (const uint8_t*)d->h_dib.p;
}
const void* jsnoop_get_bitmap_dev(JsnoopDecoder* d) { return d->have_image ? d->batch->dev.dib + d->batch->imgs[d->img].dib_off : nullptr; }
void jsnoop_get_pixmap_ptrs(JsnoopDecoder* d, const int16_t** y, const int16_t** cb, const int16_t** cr)
{
    *y = *cb = *cr = nullptr;
    if (!d->have_image) return;
    const JsImage& im = d->batch->imgs[d->img];
    const size_t psz = (size_t)im.blk_xmax * 8 * im.blk_ymax * 8;
    int16_t* hp = nullptr;
    if (!(d->host_valid & 2)) {
        if (d->h_planes.ensure(psz * 3 * sizeof(int16_t))) return;
        hp = (int16_t*)d->h_planes.p; memset(hp, 0, psz * 3 * sizeof(int16_t));
        if (d->batch->read_planes(d->img, hp, hp + psz, hp + 2 * psz)) return;
        d->host_valid |= 2;
    }
    hp = (int16_t*)d->h_planes.p;
    *y = hp;
    if (im.ncomp == 3) { *cb = hp + psz; *cr = hp + 2 * psz; }
}
void jsnoop_lookup_file_pos_mcu(JsnoopDecoder* d, unsigned mx, unsigned my, unsigned* byte, unsigned* bit)
{
    d->ensure_side();
    *byte = *bit = 0; if (!d->have_image || mx >= d->geom[2] || my >= d->geom[3]) return;
    uint32_t p = d->h_side[JS_SIDE_MCUMAP + mx + my * d->geom[2]]; *bit = p & 7; *byte = p >> 4;     // UnpackFileOffset :5123
}
void jsnoop_lookup_file_pos_pix(JsnoopDecoder* d, unsigned px, unsigned py, unsigned* byte, unsigned* bit)
{ jsnoop_lookup_file_pos_mcu(d, px / d->geom[0], py / d->geom[1], byte, bit); }                      // :5001-5009
const uint32_t* jsnoop_mcu_file_map(JsnoopDecoder* d) { d->ensure_side(); return d->have_image ? d->h_side.data() + JS_SIDE_MCUMAP : nullptr; }
void jsnoop_blk_dc_ptrs(JsnoopDecoder* d, const int16_t** y, const int16_t** cb, const int16_t** cr)
{
    d->ensure_side();
    *y = *cb = *cr = nullptr; if (!d->have_image) return;
    const uint32_t nmcu = d->geom[2] * d->geom[3], nblk = d->geom[4] * d->geom[5], w = 2 * ((nblk + 1) / 2);
    const int16_t* base = (const int16_t*)(d->h_side.data() + JS_SIDE_MCUMAP + nmcu);
    *y = base; if (d->batch->imgs[d->img].ncomp == 3) { *cb = base + w; *cr = base + 2 * w; }
}
void jsnoop_lookup_blk_ycc(JsnoopDecoder* d, unsigned bx, unsigned by, int* y, int* cb, int* cr)      // :5037-5047
{
    const int16_t *py, *pcb, *pcr; jsnoop_blk_dc_ptrs(d, &py, &pcb, &pcr);
    *y = *cb = *cr = 0; if (!py || bx >= d->geom[4] || by >= d->geom[5]) return;
    size_t i = bx + (size_t)by * d->geom[4]; *y = py[i]; if (pcb) { *cb = pcb[i]; *cr = pcr[i]; }
}
const uint32_t* jsnoop_dht_histo(JsnoopDecoder* d) { d->ensure_side(); return d->have_image ? d->h_side.data() + JS_SIDE_HISTO : d->zero_histo; }
void jsnoop_scan_status(JsnoopDecoder* d, unsigned* o) { d->ensure_side(); for (int i = 0; i < 8; i++) o[i] = d->have_image ? d->h_side[i] : 0; }
void jsnoop_bright_avg(JsnoopDecoder* d, int* o)
{
    memset(o, 0, 10 * sizeof(int)); o[1] = o[2] = o[3] = -32768;
    if (!d->have_image || !d->preview_is_jpeg) return;
    const JsImage& im = d->batch->imgs[d->img];
    const uint64_t key = ((uint64_t)d->h_side[13] << 32) | d->h_side[12];
    const uint32_t yk = (uint32_t)(key >> 32), idx = 0xFFFFFFFFu - (uint32_t)key;
    o[0] = 1;
    if (yk != 0) {            // some pixel beat the -32768 start value (:4723-4730)
        const uint32_t px = idx % im.img_x, pyy = idx / im.img_x; const size_t pi = (size_t)pyy * im.blk_xmax * 8 + px;
        o[1] = (int)yk - 32768; o[2] = o[3] = 0; o[7] = (int)(px / im.mcu_w); o[8] = (int)(pyy / im.mcu_h);   // (one component: Cb = Cr = 0, :4709-4715)
        const bool cached = d->report_cache && d->h_bright[3] == d->h_side[12] && d->h_bright[4] == d->h_side[13];
        if (im.ncomp == 3 && cached) { o[2] = (int)d->h_bright[0]; o[3] = (int)d->h_bright[1]; }
        else if (im.ncomp == 3) {
            if (d->host_valid & 2) { const int16_t* hp = (const int16_t*)d->h_planes.p; const size_t psz = (size_t)im.blk_xmax * 8 * im.blk_ymax * 8; o[2] = hp[psz + pi]; o[3] = hp[2 * psz + pi]; }
            else if (d->batch->opt_want_planes) {                   // the two chroma samples of that pixel straight from HBM (not the whole planes)
                const size_t psz = (size_t)im.blk_xmax * 8 * im.blk_ymax * 8; int16_t c2[2] = { 0, 0 };
                hipSetDevice(d->batch->device);
                if (d->batch->d2h_staged(&c2[0], d->batch->dev.planes + im.plane_off + psz + pi, 2) || d->batch->d2h_staged(&c2[1], d->batch->dev.planes + im.plane_off + 2 * psz + pi, 2))
                    d->log(2, "*** ERROR: reading the brightest pixel's chroma back failed: %s", g_err.c_str());
                o[2] = c2[0]; o[3] = c2[1];
            }
        }
    }
    if (d->report_cache && d->h_bright[3] == d->h_side[12] && d->h_bright[4] == d->h_side[13] && (im.ncomp != 3 || d->batch->opt_want_planes || !(d->host_valid & 2))) {
        const uint32_t bgra = d->h_bright[2];                      // (fetched with the side block: k_bright_probe ran the same routine on the same three values)
        o[4] = (bgra >> 16) & 255; o[5] = (bgra >> 8) & 255; o[6] = bgra & 255;
    } else {   // RGB of the brightest pixel through the device colour routine (:4805-4811)
        JsnoopBatch* b = d->batch; uint32_t bgra = 0; hipSetDevice(b->device);
        js_launch_color_probe(b->stream, o[1], o[2], o[3], (uint32_t*)b->dev.probe);
        HIP_NOTE(hipMemcpyAsync(&bgra, b->dev.probe, 4, hipMemcpyDeviceToHost, b->stream)); HIP_NOTE(hipStreamSynchronize(b->stream));
        o[4] = (bgra >> 16) & 255; o[5] = (bgra >> 8) & 255; o[6] = bgra & 255;
    }
    unsigned long npix = (unsigned)((im.img_y + 1) * (im.img_x + 1)); if (!npix) npix = 1;
    o[9] = (int)(d->h_side[15] / npix);
}
const float* jsnoop_idct_lut(JsnoopDecoder* d) { return &d->batch->lut[0][0]; }
const uint32_t* jsnoop_dht_lookupfast(JsnoopDecoder* d) { return &d->t.dht_fast[0][0][0]; }
void jsnoop_idct_block(JsnoopDecoder* d, const int16_t* coef64, float* out64)
{
    JsnoopBatch* b = d->batch; hipSetDevice(b->device);
    size_t c = b->cap.probe; if (grow(&b->dev.probe, &c, 1024)) return; b->cap.probe = c;
    HIP_NOTE(hipMemcpyAsync(b->dev.probe, coef64, 128, hipMemcpyHostToDevice, b->stream));
    js_launch_idct_probe(b->stream, b->d_lut, (const int16_t*)b->dev.probe, (float*)(b->dev.probe + 256));
    HIP_NOTE(hipMemcpyAsync(out64, b->dev.probe + 256, 256, hipMemcpyDeviceToHost, b->stream));
    HIP_NOTE(hipStreamSynchronize(b->stream));
}
int jsnoop_color_sweep(JsnoopDecoder* d, uint32_t* out_bgra)
{
    JsnoopBatch* b = d->batch; hipSetDevice(b->device);
    uint32_t* tmp = nullptr;
    if (hipMalloc(&tmp, (size_t)4 << 24) != hipSuccess) { js_set_error("jsnoop_color_sweep: hipMalloc failed"); return -1; }
    js_launch_color_sweep(b->stream, tmp);
    hipError_t e = hipMemcpyAsync(out_bgra, tmp, (size_t)4 << 24, hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    hipFree(tmp);
    if (e != hipSuccess) { js_set_error("jsnoop_color_sweep: device error"); return -1; }
    return 0;
}
int jsnoop_last_path(JsnoopDecoder* d) { return d->last_path; }
uint32_t jsnoop_last_flags(JsnoopDecoder* d) { return d->last_flags; }
int jsnoop_last_side_mode(JsnoopDecoder* d)
{
    if (!d->have_image) return 0;
    if (d->last_path == 2) return 2;
    return (size_t)d->img < d->batch->side_mode.size() ? (int)d->batch->side_mode[d->img] : 0;
}

void jsnoop_set_preview_mode(JsnoopDecoder* d, unsigned mode) { d->preview_mode = mode; d->rerender(); }      // :633-639
unsigned jsnoop_get_preview_mode(JsnoopDecoder* d) { return d->preview_mode; }
void jsnoop_set_preview_ycc_offset(JsnoopDecoder* d, unsigned mx, unsigned my, int y, int cb, int cr)       // :650-659
{ d->shift_mcu_x = mx; d->shift_mcu_y = my; d->shift_y = y; d->shift_cb = cb; d->shift_cr = cr; d->rerender(); }
void jsnoop_get_preview_ycc_offset(JsnoopDecoder* d, unsigned* mx, unsigned* my, int* y, int* cb, int* cr)                        // :670-677
{ *mx = d->shift_mcu_x; *my = d->shift_mcu_y; *y = d->shift_y; *cb = d->shift_cb; *cr = d->shift_cr; }
void jsnoop_set_preview_mcu_insert(JsnoopDecoder* d, unsigned mx, unsigned my, int len)                                          // :682-690
{ d->ins_mcu_x = mx; d->ins_mcu_y = my; d->ins_mcu_len = (unsigned)len; d->rerender(); }
void jsnoop_get_preview_mcu_insert(JsnoopDecoder* d, unsigned* mx, unsigned* my, unsigned* len) { *mx = d->ins_mcu_x; *my = d->ins_mcu_y; *len = d->ins_mcu_len; }

// ---- batch ---------------------------------------------------------------------------
JsnoopBatch* jsnoop_batch_create(void* stream)
{
    if (jsnoop_device_count() <= 0) { js_set_error("no HIP device visible: libjsnoop_gpu has no CPU fallback"); return nullptr; }
    JsnoopBatch* b = new JsnoopBatch(stream);
    if (b->init()) { delete b; return nullptr; }
    return b;
}
void jsnoop_batch_destroy(JsnoopBatch* b) { delete b; }
void jsnoop_batch_clear(JsnoopBatch* b) { b->clear(); }
void jsnoop_batch_set_options(JsnoopBatch* b, int decode_ac, int want_planes, int force_exact)
{ b->opt_decode_ac = decode_ac; b->opt_want_planes = want_planes; b->opt_force_exact = force_exact; b->uploaded = false; }
int jsnoop_batch_add(JsnoopBatch* b, const JsnoopDecoder* tables, const uint8_t* file, size_t len, unsigned scan_start)
{
    // the geometry step rewrites the sampling factors of single-component scans and records the geometry (as DecodeScanImg does
    // on its own object): done on a private copy here, `tables` really is const and may be shared between batches and threads
    JsnoopDecoder tmp;
    tmp.t = tables->t; tmp.opt_decode_ac = tables->opt_decode_ac; tmp.opt_err_max = tables->opt_err_max;
    tmp.log_fn = tables->log_fn; tmp.log_user = tables->log_user;
    tmp.preview_mode = tables->preview_mode; tmp.shift_y = tables->shift_y; tmp.shift_cb = tables->shift_cb; tmp.shift_cr = tables->shift_cr;
    tmp.shift_mcu_x = tables->shift_mcu_x; tmp.shift_mcu_y = tables->shift_mcu_y;
    return b->add(&tmp, file, len, scan_start, 1);
}
int jsnoop_batch_add_jpeg(JsnoopBatch* b, const uint8_t* file, size_t len)
{
    if (js_is_progressive(file, len)) { JsnoopDecoder tmpd; return b->add_progressive(&tmpd, file, len); }   // beyond the reference (it refuses SOF2)
    JsnoopDecoder tmp; unsigned scan_start = 0;
    if (js_jfif_walk(&tmp, file, len, &scan_start)) return -1;
    return b->add(&tmp, file, len, scan_start, 1);
}
int jsnoop_batch_tile(JsnoopBatch* b, int total) { return b->tile(total); }
static void js_resolve_split(JsnoopBatch* b)
{
    uint64_t scan_total = 0; for (const JsImage& im : b->imgs) scan_total += im.scan_len;
    b->split_parts = (b->imgs.size() >= 2 && (b->tune.split == 2 || (b->tune.split == 0 && scan_total >= JS_SPLIT_FROM_BYTES))) ? 2 : 1;
}
int jsnoop_batch_set_split(JsnoopBatch* b, int parts)
{
    if (!b || parts < 0 || parts > 2) { js_set_error("jsnoop_batch_set_split: parts must be 0 (automatic), 1 or 2"); return -1; }
    b->tune.split = parts; js_resolve_split(b);
    return 0;
}
int jsnoop_batch_split_parts(const JsnoopBatch* b) { return b ? b->split_parts : 1; }
void jsnoop_tuning_defaults(JsnoopTuning* out) { if (out) *out = js_env_tuning(); }      // (writes sizeof(JsnoopTuning) bytes: the struct of THIS header; a caller built against an older, shorter struct uses the sized form)
void jsnoop_tuning_defaults_sized(JsnoopTuning* out, uint32_t struct_size)
{
    if (!out || struct_size < 8) return;
    JsnoopTuning t = js_env_tuning();
    const size_t sz = std::min<size_t>(struct_size, sizeof(JsnoopTuning));
    t.struct_size = (uint32_t)sz;
    memcpy(out, &t, sz);                                         // nothing is written past the caller's struct
}
int jsnoop_batch_set_tuning(JsnoopBatch* b, const JsnoopTuning* t)
{
    if (!b || !t) { js_set_error("jsnoop_batch_set_tuning: null argument"); return -1; }
    JsnoopTuning in;
    if (js_import_tuning(t, &in)) return -1;
    b->tune = in; b->uploaded = false;                           // sub-sequence length, synchronisation form and work split are fixed by upload()
    if (b->helper) b->helper->tune = in;
    js_prog_dirty(b);
    js_resolve_split(b);
    return 0;
}
void jsnoop_batch_get_tuning(const JsnoopBatch* b, JsnoopTuning* out) { if (b && out) js_export_tuning(b->tune, out); }
int jsnoop_set_tuning(JsnoopDecoder* d, const JsnoopTuning* t)
{
    if (!d || !d->batch) { js_set_error("jsnoop_set_tuning: no decoder"); return -1; }
    return jsnoop_batch_set_tuning(d->batch, t);
}
int jsnoop_batch_count(const JsnoopBatch* b) { return (int)b->imgs.size(); }
int jsnoop_batch_upload(JsnoopBatch* b) { return b->upload(); }
int jsnoop_batch_decode(JsnoopBatch* b) { return b->decode(false); }
int jsnoop_batch_sync(JsnoopBatch* b) { return b->sync(); }
const char* jsnoop_stage_name(int s) { return s >= 0 && s < JSNOOP_NUM_STAGES ? kStageName[s] : ""; }
double jsnoop_batch_decode_timed(JsnoopBatch* b, int reps, double* stage_ms)
{
    if (reps < 1) reps = 1;
    double tot[JSNOOP_NUM_STAGES] = {0}, whole = 0;
    for (int r = 0; r < reps; r++) {
        if (b->decode(true)) return -1;
        if (hipStreamSynchronize(b->stream) != hipSuccess) { js_set_error("stream sync failed"); return -1; }
        for (int s = 0; s < JSNOOP_NUM_STAGES; s++) {
            float ms = 0; hipEventElapsedTime(&ms, b->ev[s], b->ev[s + 1]);
            if (b->last_timed_split && s >= 1) { float m2 = 0; hipEventElapsedTime(&m2, b->ev2[s], b->ev2[s + 1]); ms = 0.5f * (ms + m2); }   // two halves side by side: the mean of their stage times
            tot[s] += ms;
        }
        float ms = 0; hipEventElapsedTime(&ms, b->ev[0], b->ev[JSNOOP_NUM_STAGES]);
        if (b->last_timed_split) { float m2 = 0; hipEventElapsedTime(&m2, b->ev[0], b->ev2[JSNOOP_NUM_STAGES]); ms = std::max(ms, m2); }
        whole += ms;
    }
    if (stage_ms) for (int s = 0; s < JSNOOP_NUM_STAGES; s++) stage_ms[s] = tot[s] / reps;
    return whole / reps;
}
int jsnoop_batch_image_info(const JsnoopBatch* b, int i, unsigned* o)
{
    if (i < 0 || (size_t)i >= b->imgs.size()) { js_set_error("jsnoop_batch_image_info: image index out of range"); return -1; }
    const JsImage& im = b->imgs[i];
    unsigned v[16] = { im.dim_x, im.dim_y, im.img_x, im.img_y, im.mcu_w, im.mcu_h, im.mcu_xmax, im.mcu_ymax, im.blk_xmax, im.blk_ymax,
                       im.scan_len, (size_t)i < b->host_flags.size() ? b->host_flags[i] : 0u, (size_t)i < b->host_path.size() ? b->host_path[i] : 0u,
                       im.ncomp, im.file_len, im.total_blocks };
    memcpy(o, v, sizeof v); return 0;
}
const void* jsnoop_batch_dib_dev(const JsnoopBatch* b, int i) { return (i < 0 || (size_t)i >= b->imgs.size() || !b->dev.dib) ? nullptr : b->dev.dib + b->imgs[i].dib_off; }
int jsnoop_batch_read_dib(JsnoopBatch* b, int i, uint8_t* dst) { return b->read_dib(i, dst); }
int jsnoop_batch_read_planes(JsnoopBatch* b, int i, int16_t* y, int16_t* cb, int16_t* cr) { return b->read_planes(i, y, cb, cr); }
int jsnoop_batch_read_coefs(JsnoopBatch* b, int i, int16_t* dst, size_t max_blocks)
{
    if (i < 0 || (size_t)i >= b->imgs.size()) return -1;
    const JsImage& im = b->imgs[i]; size_t nb = std::min<size_t>(max_blocks, im.total_blocks);
    HIP_TRY(hipSetDevice(b->device));
    if (b->d2h_staged(dst, b->dev.coef + im.coef_off * 64, nb * 128)) return -1;
    return (int)nb;
}
int jsnoop_batch_dib_hashes(JsnoopBatch* b, uint64_t* dst)
{
    HIP_TRY(hipSetDevice(b->device));
    const uint32_t n = (uint32_t)b->imgs.size();
    HIP_TRY(hipMemsetAsync(b->dev.sums, 0, n * 8, b->stream));
    js_launch_dib_checksum(b->stream, b->dev.imgs, n, b->dev.dib, (unsigned long long*)b->dev.sums);
    if (b->d2h_staged(dst, b->dev.sums, n * 8)) return -1;
    return 0;
}
// ---- per-image results of a batch beyond pixels (what DoBatchFileProcess's per-file pass produces, source/JPEGsnoopCore.cpp:805-808):
//      a JsnoopDecoder *view* onto image i of the caller's batch runs the very code of the single-image API (side pass on request,
//      statistics pass, event records, report text, TIFF writer) -- one implementation, two entry points.
static int js_batch_view(JsnoopBatch* b, int i, JsnoopDecoder& v, const char* who)
{
    if (!b || i < 0 || (size_t)i >= b->imgs.size()) { js_set_error("%s: image index out of range", who); return -1; }
    if (!b->uploaded || b->host_path.size() != b->imgs.size()) { js_set_error("%s: the batch has not been decoded (upload / decode / sync first)", who); return -1; }
    const JsImage& im = b->imgs[i];
    v.batch = b; v.img = i; v.have_image = true; v.host_valid = 0;
    v.last_path = (int)b->host_path[i]; v.last_flags = b->host_flags[i];
    const bool display = (size_t)i < b->hinfo.size() ? b->hinfo[i].display : true;
    v.preview_is_jpeg = display; v.opt_decode_ac = (int)im.decode_ac; v.opt_err_max = im.err_max;
    if ((size_t)i < b->hinfo.size()) { v.t.dht_setmax[0] = b->hinfo[i].dht_setmax[0]; v.t.dht_setmax[1] = b->hinfo[i].dht_setmax[1]; }
    v.preview_mode = im.preview_mode; v.shift_y = im.shift_y; v.shift_cb = im.shift_cb; v.shift_cr = im.shift_cr; v.shift_mcu_x = im.shift_mcu_x; v.shift_mcu_y = im.shift_mcu_y;
    v.geom[0] = im.mcu_w; v.geom[1] = im.mcu_h; v.geom[2] = im.mcu_xmax; v.geom[3] = im.mcu_ymax; v.geom[4] = im.blk_xmax; v.geom[5] = im.blk_ymax; v.geom[6] = im.img_x; v.geom[7] = im.img_y;
    v.side_ready = v.last_path != 1 || ((size_t)i < b->side_done.size() && b->side_done[i]);   // the exact-mirror kernel fills the side block as it goes; a progressive image has none (zeros + the back end's reductions)
    if (hipSetDevice(b->device) != hipSuccess) { js_set_error("%s: device error", who); return -1; }
    return 0;
}
int jsnoop_batch_enable_log(JsnoopBatch* b, int on)
{
    if (!b) { js_set_error("jsnoop_batch_enable_log: bad argument"); return -1; }
    if (b->opt_events != (on ? 1 : 0)) { b->opt_events = on ? 1 : 0; b->uploaded = false; }
    return 0;
}
int jsnoop_batch_side_outputs(JsnoopBatch* b, int i, uint32_t* mcu_map, int16_t* dc_y, int16_t* dc_cb, int16_t* dc_cr, uint32_t* dht_histo, unsigned* status8, int* bright_avg10)
{
    JsnoopDecoder v;
    if (js_batch_view(b, i, v, "jsnoop_batch_side_outputs")) return -1;
    const JsImage& im = b->imgs[i];
    if (bright_avg10 && im.ncomp == 3 && !b->opt_want_planes) { js_set_error("jsnoop_batch_side_outputs: the brightest pixel's Cb / Cr / RGB need the planes (want_planes)"); return -1; }
    v.want_bright = bright_avg10 != nullptr;                       // (the brightest pixel's chroma + RGB come back with the side block, one round trip)
    v.ensure_side();
    if (!v.side_ready) return -1;
    if (v.h_side.empty()) v.fetch_side();
    const uint32_t nmcu = im.mcu_xmax * im.mcu_ymax, nblk = im.blk_xmax * im.blk_ymax, w = 2 * ((nblk + 1) / 2);
    const uint32_t* sd = v.h_side.data();
    if (mcu_map) memcpy(mcu_map, sd + JS_SIDE_MCUMAP, (size_t)nmcu * 4);
    const int16_t* base = (const int16_t*)(sd + JS_SIDE_MCUMAP + nmcu);
    if (dc_y) memcpy(dc_y, base, (size_t)nblk * 2);
    if (im.ncomp == 3) { if (dc_cb) memcpy(dc_cb, base + w, (size_t)nblk * 2); if (dc_cr) memcpy(dc_cr, base + 2 * w, (size_t)nblk * 2); }
    if (dht_histo) memcpy(dht_histo, sd + JS_SIDE_HISTO, 2 * 4 * 17 * 4);
    if (status8) for (int k = 0; k < 8; k++) status8[k] = sd[k];
    if (bright_avg10) jsnoop_bright_avg(&v, bright_avg10);
    return 0;
}
int jsnoop_batch_log(JsnoopBatch* b, int i, int histo_en, int stat_clip_en, int quiet, jsnoop_log_fn fn, void* user)
{
    JsnoopDecoder v;
    if (js_batch_view(b, i, v, "jsnoop_batch_log")) return -1;
    if (!fn) { js_set_error("jsnoop_batch_log: no log sink"); return -1; }
    if (v.last_path == 3) { js_set_error("jsnoop_batch_log: a progressive image has no DecodeScanImg log (the reference refuses SOF2, source/JfifDecode.cpp:4827-4833)"); return -1; }
    if (!b->event_words) { js_set_error("jsnoop_batch_log: the decoder's event records were not kept (jsnoop_batch_enable_log before upload)"); return -1; }
    const JsImage& im = b->imgs[i];
    const bool display = v.preview_is_jpeg;
    if (display && !b->opt_want_planes && (im.ncomp == 3 || histo_en || stat_clip_en)) { js_set_error("jsnoop_batch_log: the report quotes plane samples (want_planes)"); return -1; }
    v.log_fn = fn; v.log_user = user;
    v.opt_err_max = im.err_max;
    js_emit_head_events(&v, b->pinned + im.file_off, im.file_len, im.scan_start);       // what the reader's first refill logs goes out in front of the heading (:3007-3022)
    if (!quiet) {                                                   // the lines DecodeScanImg writes before its MCU loop (:3021-3025, :3126-3135)
        v.log(0, "*** Decoding SCAN Data ***"); v.log(0, "  OFFSET: 0x%08X", im.scan_start);
        if (display && im.decode_ac) v.log(0, "  Scan Decode Mode: Full IDCT (AC + DC)");
        else { v.log(0, "  Scan Decode Mode: No IDCT (DC only)");
               v.log(1, "    NOTE: Low-resolution DC component shown. Can decode full-res with [Options->Scan Segment->Full IDCT]"); }
        v.log(0, "");
    }
    v.hist_latched = histo_en != 0; v.clip_latched = stat_clip_en != 0;
    v.fetch_side();
    if (display) v.stats_pass();
    v.ensure_side();
    js_emit_decode_events(&v);
    if (!quiet) v.log(0, "");
    v.flush_pending_log();
    js_emit_report(&v, display, quiet != 0);
    return 0;
}
int jsnoop_batch_export_tiff(JsnoopBatch* b, int i, const char* path, int mode)
{
    JsnoopDecoder v;
    if (js_batch_view(b, i, v, "jsnoop_batch_export_tiff")) return -1;
    if (mode == 2 && !b->opt_want_planes) { js_set_error("jsnoop_batch_export_tiff: YCC export needs the planes (want_planes)"); return -1; }
    return jsnoop_export_tiff(&v, path, mode);
}
uint64_t jsnoop_batch_algorithmic_bytes(const JsnoopBatch* b)
{ uint64_t s = 0; for (const JsImage& im : b->imgs) s += (uint64_t)im.scan_len + (uint64_t)im.img_x * im.img_y * 4; return s; }
uint64_t jsnoop_batch_pixels(const JsnoopBatch* b)
{ uint64_t s = 0; for (const JsImage& im : b->imgs) s += (uint64_t)im.dim_x * im.dim_y; return s; }

} // extern "C"

// ------------------------------------------------------------------------------ helpers
int JsnoopBatch::read_dib(int i, uint8_t* dst)
{
    if (i < 0 || (size_t)i >= imgs.size()) { js_set_error("image index out of range"); return -1; }
    const JsImage& im = imgs[i];
    JsRange r_("jsnoop:read_dib (D2H)");
    HIP_TRY(hipSetDevice(device));
    return d2h_staged(dst, dev.dib + im.dib_off, (size_t)im.img_x * im.img_y * 4);
}
// Device -> caller's (pageable) memory through a page-locked landing buffer of this batch, in chunks: a direct asynchronous copy into
// pageable memory is slower and, once the runtime has taken that path, makes every later synchronisation of the process dearer.
int JsnoopBatch::d2h_staged(void* dst, const void* src, size_t bytes)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, dst) == hipSuccess && at.type == hipMemoryTypeHost) {          // already page-locked: straight there
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream)); HIP_TRY(hipStreamSynchronize(stream)); return 0;
    }
    (void)hipGetLastError();                                                                          // (pageable pointers make the query fail: not an error)
    const size_t chunk = 32u << 20;
    if (!d2h_land) { HIP_TRY(hipHostMalloc((void**)&d2h_land, chunk, hipHostMallocDefault)); }
    for (size_t o = 0; o < bytes; o += chunk) {
        const size_t nb = std::min(chunk, bytes - o);
        HIP_TRY(hipMemcpyAsync(d2h_land, (const uint8_t*)src + o, nb, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        memcpy((uint8_t*)dst + o, d2h_land, nb);
    }
    return 0;
}
int JsnoopBatch::read_planes(int i, int16_t* y, int16_t* cb, int16_t* cr)
{
    if (i < 0 || (size_t)i >= imgs.size() || !opt_want_planes) { js_set_error("planes not available"); return -1; }
    const JsImage& im = imgs[i]; const size_t psz = (size_t)im.blk_xmax * 8 * im.blk_ymax * 8;
    int16_t* dst[3] = { y, cb, cr };
    HIP_TRY(hipSetDevice(device));
    for (uint32_t c = 0; c < im.ncomp; c++) if (dst[c] && d2h_staged(dst[c], dev.planes + im.plane_off + c * psz, psz * 2)) return -1;
    return 0;
}
void JsnoopDecoder::ensure_side()
{
    if (!have_image || side_ready) return;
    if (js_side_only(batch, (uint32_t)img) == 0) { side_ready = true; fetch_side(); }
}
void JsnoopDecoder::fetch_side()
{
    const JsImage& im = batch->imgs[img];
    JsnoopBatch* b = batch;
    h_side.assign(js_side_words(im.mcu_xmax * im.mcu_ymax, im.blk_xmax * im.blk_ymax), 0);
    hipSetDevice(b->device);
    report_cache = false;
    // before the side pass has run only the sixteen status words are this decode's (the decode clears nothing else of the side block: histogram and
    // maps would be an earlier decode's, or another image's of an earlier batch) -- they stay zero in the host copy until side_ready
    const size_t words = side_ready ? h_side.size() : (size_t)JS_SIDE_HISTO;
    // With a log callback the report follows: what it reads back besides (event list, restart marks, the brightest pixel's chroma and RGB -- a probe launch) comes
    // in the same round trip, through the page-locked landing buffer: five waits of ~25 us each otherwise.
    const size_t nmcu = (size_t)im.mcu_xmax * im.mcu_ymax, ev_words = b->event_words ? 1 + (size_t)JS_EV_WORDS * JS_EV_MAX : 0;
    const bool with_report = (log_fn != nullptr || want_bright) && side_ready && preview_is_jpeg && b->uploaded && !js_prog_count(b);
    const size_t o_side = 0, o_ev = (words * 4 + 63) & ~(size_t)63, o_rst = o_ev + ((ev_words * 4 + 63) & ~(size_t)63), o_br = o_rst + ((nmcu + 63) & ~(size_t)63), total = o_br + 64;
    if (!with_report || total > (32u << 20)) {
        if (b->d2h_staged(h_side.data(), b->dev.side + im.side_off, words * 4)) log(2, "*** ERROR: reading the side block back failed: %s", g_err.c_str());
        return;
    }
    bool ok = true;
    auto note = [&](hipError_t e) { if (e != hipSuccess) { ok = false; js_set_error("%s", hipGetErrorString(e)); } };
    if (!b->d2h_land) note(hipHostMalloc((void**)&b->d2h_land, 32u << 20, hipHostMallocDefault));
    if (ok) {
        size_t c = b->cap.probe; if (grow(&b->dev.probe, &c, 1024)) ok = false; else b->cap.probe = c;
    }
    if (ok) {
        uint8_t* land = (uint8_t*)b->d2h_land;
        js_launch_bright_probe(b->stream, b->dev.imgs, (uint32_t)img, b->dev.side, b->opt_want_planes ? b->dev.planes : nullptr, (uint32_t*)b->dev.probe);
        note(hipMemcpyAsync(land + o_side, b->dev.side + im.side_off, words * 4, hipMemcpyDeviceToHost, b->stream));
        if (ev_words) note(hipMemcpyAsync(land + o_ev, b->dev.events + im.ev_off, ev_words * 4, hipMemcpyDeviceToHost, b->stream));
        note(hipMemcpyAsync(land + o_rst, b->dev.mcu_rst + im.mcu_off, nmcu, hipMemcpyDeviceToHost, b->stream));
        note(hipMemcpyAsync(land + o_br, b->dev.probe, 32, hipMemcpyDeviceToHost, b->stream));
        note(hipStreamSynchronize(b->stream));
        if (ok) {
            memcpy(h_side.data(), land + o_side, words * 4);
            h_events.assign((const uint32_t*)(land + o_ev), (const uint32_t*)(land + o_ev) + ev_words);
            h_rstf.assign(land + o_rst, land + o_rst + nmcu);
            memcpy(h_bright, land + o_br, 32);
            report_cache = true;
        }
    }
    if (!ok) log(2, "*** ERROR: reading the side block back failed: %s", g_err.c_str());
}
void JsnoopDecoder::rerender()                                  // CalcChannelPreview :4965 on the retained data: colour kernel only
{
    if (!have_image) return;
    JsnoopBatch* b = batch; JsImage& im = b->imgs[0];
    im.preview_mode = preview_mode; im.shift_y = shift_y; im.shift_cb = shift_cb; im.shift_cr = shift_cr; im.shift_mcu_x = shift_mcu_x; im.shift_mcu_y = shift_mcu_y;
    hipSetDevice(b->device);
    HIP_NOTE(hipMemcpyAsync(b->dev.imgs, &im, sizeof im, hipMemcpyHostToDevice, b->stream));
    HIP_NOTE(hipMemsetAsync(b->dev.side + im.side_off + 12, 0, 8, b->stream));       // brightest-pixel key and sum of Y are recomputed (word 14, the block count, stays)
    HIP_NOTE(hipMemsetAsync(b->dev.side + im.side_off + 15, 0, 4, b->stream));
    if (b->launch_back_end(1)) log(2, "*** ERROR: device re-render failed: %s", g_err.c_str());
    hipStreamSynchronize(b->stream);
    host_valid = 0; fetch_side();
    stats_pass(); flush_pending_log();
}

// ConvertYCCtoRGB (:4229) instead of ConvertYCCtoRGBFastFloat when bHistoEn or bStatClipEn (:4742-4747): same pixels,
// plus the statistics -- one reduction kernel over the retained planes per CalcChannelPreview.
void JsnoopDecoder::stats_pass()
{
    if (!have_image || !preview_is_jpeg || !(hist_latched || clip_latched)) return;   // CalcChannelPreview needs the DIB (:4971-4974)
    const bool want_notes = log_fn != nullptr;
    if (want_notes) ensure_side();                                  // the warning text quotes the reader's final position (GetScanBufPos)
    if (batch->color_stats_pass(img, hist_latched, stats, &warn_ycc_clip, want_notes ? &pending_log : nullptr, want_notes ? h_side[4] : 0, want_notes ? h_side[5] : 0))
        log(2, "*** ERROR: colour statistics pass failed: %s", g_err.c_str());
}
void JsnoopDecoder::flush_pending_log()
{
    for (auto& l : pending_log) if (log_fn) log_fn(log_user, l.first, l.second.c_str());
    pending_log.clear();
}
int JsnoopBatch::color_stats_pass(int i, bool hist_en, uint32_t* acc, unsigned* warn_used, std::vector<std::pair<int, std::string>>* notes, uint32_t pos0, uint32_t align)
{
    if (i < 0 || (size_t)i >= imgs.size() || !opt_want_planes || !uploaded) { js_set_error("colour statistics need a decoded image with planes"); return -1; }
    HIP_TRY(hipSetDevice(device));
    size_t c = cap.probe; if (grow(&dev.probe, &c, JS_STATS_DEV_WORDS * 4 + 64)) return -1; cap.probe = c;
    uint32_t* dst = reinterpret_cast<uint32_t*>(dev.probe);
    std::vector<uint32_t> h(JS_STATS_DEV_WORDS);
    HIP_TRY(hipMemsetAsync(dst, 0, JS_STATS_DEV_WORDS * 4, stream));
    js_launch_color_stats(stream, dev.imgs, (uint32_t)i, dev.planes, hist_en, dst);
    HIP_TRY(hipMemcpyAsync(h.data(), dst, JS_STATS_DEV_WORDS * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    int32_t* ai = reinterpret_cast<int32_t*>(acc); const int32_t* hi = reinterpret_cast<const int32_t*>(h.data());
    for (int k = 0; k < 36; k += 3) { ai[k] = std::min(ai[k], hi[k]); ai[k + 1] = std::max(ai[k + 1], hi[k + 1]); acc[k + 2] += h[k + 2]; }
    acc[36] += h[36];
    for (int k = 43; k < 50; k++) acc[k] += h[k];                                 // RGB clip counters
    for (int k = 50; k < JS_STATS_WORDS; k++) acc[k] += h[k];                     // histograms
    // YCC range events: counted -- and reported -- only while fewer than YCC_CLIP_REPORT_MAX (10) warnings were issued (:4372-4378)
    uint32_t total = 0; for (int k = 0; k < 6; k++) total += h[2482 + k];
    const unsigned left = *warn_used < 10 ? 10 - *warn_used : 0;
    if (total <= left && !(notes && total)) { for (int k = 0; k < 6; k++) acc[37 + k] += h[2482 + k]; *warn_used += total; }
    else if (left) {
        const unsigned take = std::min<uint32_t>(left, total);
        js_launch_clip_order(stream, dev.imgs, (uint32_t)i, dev.planes, take, dst);
        uint32_t first[6 + 5 * 10];
        HIP_TRY(hipMemcpyAsync(first, dst, sizeof first, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        for (int k = 0; k < 6; k++) acc[37 + k] += first[k];
        if (notes) {
            static const char* kKind[6] = { "Y Overflow", "Y Underflow", "Cb Overflow", "Cb Underflow", "Cr Overflow", "Cr Underflow" };
            char buf[256];
            for (unsigned k = 0; k < take; k++) {
                const uint32_t* r = first + 6 + 5 * k;
                snprintf(buf, sizeof buf, "*** NOTE: YCC Clipped. MCU=(%4u,%4u) YCC=(%5d,%5d,%5d) %s @ Offset 0x%08X.%u", r[0] & 0xFFFF, r[0] >> 16,
                         (int)r[2], (int)r[3], (int)r[4], kKind[r[1] % 6], pos0, align);
                notes->emplace_back(1, buf);
                if (*warn_used + k + 1 == 10) notes->emplace_back(1, "    Only reported first 10 instances of this message...");
            }
        }
        *warn_used += take;
    }
    return 0;
}
void jsnoop_get_color_stats(JsnoopDecoder* d, uint32_t* out) { memcpy(out, d->stats, sizeof d->stats); }
int jsnoop_batch_color_stats(JsnoopBatch* b, int i, int histo_en, uint32_t* out)
{
    memset(out, 0, JS_STATS_WORDS * 4); unsigned used = 0;
    return b->color_stats_pass(i, histo_en != 0, out, &used);
}
