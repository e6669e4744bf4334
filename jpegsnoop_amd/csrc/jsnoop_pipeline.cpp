// jsnoop_pipeline.cpp -- the CwindowBuf replacement at batch scale: overlapped staging.
//
// The reference reads its input through a 128 KB sliding file window (CwindowBuf::BufLoadWindow, source/WindowBuf.cpp:351-416;
// Buf :639-714) and decodes while it reads.  Here a batch's compressed bytes cross PCIe once, from pinned host memory, and the
// pipeline keeps that transfer off the decode's critical path: two (or more) batch slots, each with its own pinned staging area,
// HBM arenas and stream.  While slot k decodes, slot k+1's bytes are uploaded; when the caller wants the pixels on the host,
// slot k-1's DIBs go back over PCIe at the same time.  No kernel here: the decode is jsnoop_batch_decode on each slot.
//   T1 = decode of a resident batch, T2 = H2D + decode, T3 = H2D + decode + D2H; jsnoop_pipeline_run reports all three
//   from one run: the serial pieces (H2D alone, decode alone, D2H alone) and the steady-state wall time per batch with them overlapped.
#include <hip/hip_runtime.h>
#include <chrono>
#include <vector>
#include "jsnoop_host.h"

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    js_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); return -1; } } while (0)

struct JsnoopPipeline {
    std::vector<JsnoopBatch*> slots;
    uint8_t* d2h_buf = nullptr; size_t d2h_cap = 0;            // pinned landing area of the DIB read-back (one chunk, reused)
    hipStream_t d2h_stream = nullptr;
    ~JsnoopPipeline()
    {
        for (JsnoopBatch* b : slots) delete b;
        if (d2h_buf) hipHostFree(d2h_buf);
        if (d2h_stream) hipStreamDestroy(d2h_stream);
    }
};

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// D2H of every DIB of a decoded slot, in chunks through the pinned landing area (the consumer would pick each chunk up there)
static int read_back(JsnoopPipeline* p, JsnoopBatch* b)
{
    const size_t chunk = p->d2h_cap;
    for (size_t off = 0; off < b->dib_bytes; off += chunk) {
        const size_t n = std::min(chunk, (size_t)b->dib_bytes - off);
        HIP_TRY(hipMemcpyAsync(p->d2h_buf, b->dev.dib + off, n, hipMemcpyDeviceToHost, p->d2h_stream));
        HIP_TRY(hipStreamSynchronize(p->d2h_stream));
    }
    return 0;
}

extern "C" {

JsnoopPipeline* jsnoop_pipeline_create(int slots)
{
    if (slots < 1 || slots > 8) { js_set_error("pipeline: 1..8 slots"); return nullptr; }
    JsnoopPipeline* p = new JsnoopPipeline;
    for (int i = 0; i < slots; i++) {
        JsnoopBatch* b = jsnoop_batch_create(nullptr);
        if (!b) { delete p; return nullptr; }
        p->slots.push_back(b);
    }
    return p;
}
void jsnoop_pipeline_destroy(JsnoopPipeline* p) { delete p; }
JsnoopBatch* jsnoop_pipeline_slot(JsnoopPipeline* p, int i) { return (i >= 0 && (size_t)i < p->slots.size()) ? p->slots[i] : nullptr; }

int jsnoop_pipeline_run(JsnoopPipeline* p, int batches, int d2h, double* out)
{
    const size_t n = p->slots.size();
    for (int i = 0; i < 6; i++) out[i] = 0;
    if (batches < 1) { js_set_error("pipeline: nothing to run"); return -1; }
    for (JsnoopBatch* b : p->slots) if (b->imgs.empty()) { js_set_error("pipeline: an empty slot"); return -1; }
    HIP_TRY(hipSetDevice(p->slots[0]->device));
    // warm-up: arenas allocated, every slot decoded once
    for (JsnoopBatch* b : p->slots) if (b->upload() || b->decode(false) || b->sync()) return -1;
    if (d2h) {
        if (!p->d2h_stream) HIP_TRY(hipStreamCreateWithFlags(&p->d2h_stream, hipStreamNonBlocking));
        if (!p->d2h_buf) { const size_t want = std::min<size_t>((size_t)1 << 30, (size_t)p->slots[0]->dib_bytes);
            HIP_TRY(hipHostMalloc((void**)&p->d2h_buf, want, hipHostMallocDefault)); p->d2h_cap = want; }
    }
    // ---- the pieces on their own (slot 0): H2D, decode, D2H
    JsnoopBatch* b0 = p->slots[0];
    { b0->uploaded = false; const double t = now_ms(); if (b0->upload()) return -1; out[1] = now_ms() - t; }
    { const double t = now_ms(); if (b0->decode(false)) return -1; HIP_TRY(hipStreamSynchronize(b0->stream)); out[2] = now_ms() - t; if (b0->sync()) return -1; }
    if (d2h) { const double t = now_ms(); if (read_back(p, b0)) return -1; out[3] = now_ms() - t; }
    // ---- steady state: slot k % n is re-used once its previous decode (and read-back) is done; its upload blocks this thread while the
    //      other slots' decodes run on their own streams; the decode is only enqueued
    std::vector<int> pending(n, 0);                                       // decode enqueued, results not yet picked up
    const double t0 = now_ms();
    for (int k = 0; k < batches; k++) {
        JsnoopBatch* b = p->slots[(size_t)k % n];
        if (pending[(size_t)k % n]) {
            if (b->sync()) return -1;                                      // previous batch of this slot: decode finished (flagged images re-decoded)
            if (d2h && read_back(p, b)) return -1;                        // ... and, for T3, its DIBs brought to the host while the other slots work
            pending[(size_t)k % n] = 0;
        }
        b->uploaded = false;                                               // new compressed bytes for this slot (the staged ones, re-sent)
        if (b->upload() || b->decode(false)) return -1;
        pending[(size_t)k % n] = 1;
    }
    for (size_t s = 0; s < n; s++) if (pending[s]) { if (p->slots[s]->sync()) return -1; if (d2h && read_back(p, p->slots[s])) return -1; }
    out[0] = (now_ms() - t0) / batches;
    out[4] = (double)b0->raw_bytes; out[5] = (double)b0->dib_bytes;
    return 0;
}

} // extern "C"
