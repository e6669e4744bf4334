// jsnoop_tiff.cpp -- "Export to TIFF" for a decoded image.
//
// Reference: the pixel re-arrangement of CJPEGsnoopDoc::OnToolsExporttiff (source/JPEGsnoopDoc.cpp:2110-2180) and the
// container written by FileTiff::WriteFile / WriteIfd (source/FileTiff.cpp:281-433, :436-538).  The re-arrangement
// (bottom-up BGRA DIB -> top-down RGB, 8 or 16 bit; int16 planes -> clamped 8-bit YCC) runs on the device straight from
// the DIB / planes in HBM (k_tiff_pack); the host only prepends the few hundred header bytes and writes the file.
#include <stdio.h>
#include <string.h>
#include <vector>
#include "jsnoop_host.h"
#include "jsnoop_launch.h"

namespace {

struct Bytes {
    std::vector<uint8_t> v;
    void u8(unsigned x) { v.push_back((uint8_t)x); }
    void u16(unsigned x) { u8(x >> 8); u8(x); }                  // "MM": big-endian
    void u32(unsigned x) { u16(x >> 16); u16(x); }
};

enum { T_SHORT = 3, T_LONG = 4, T_RATIONAL = 5 };

// One IFD in the reference's tag order.  Values that do not fit the 4-byte value field go to `extra`, which the file
// places right behind the IFD; `extra_at` is where that area will start, `strip_at` where the pixel data will.
struct Ifd {
    Bytes dir, extra; unsigned entries = 0, extra_at;
    explicit Ifd(unsigned extra_at_) : extra_at(extra_at_) {}
    void one(unsigned tag, unsigned type, unsigned val)
    { dir.u16(tag); dir.u16(type); dir.u32(1); if (type == T_SHORT) { dir.u16(val); dir.u16(0); } else dir.u32(val); entries++; }
    void many(unsigned tag, unsigned type, std::initializer_list<unsigned> vals)
    {
        const unsigned n = (unsigned)vals.size(), bytes = n * (type == T_SHORT ? 2u : 4u);
        dir.u16(tag); dir.u16(type); dir.u32(type == T_RATIONAL ? n / 2 : n);
        Bytes& dst = bytes > 4 ? extra : dir;
        if (bytes > 4) dir.u32(extra_at + (unsigned)extra.v.size());
        entries++;
        for (unsigned x : vals) { if (type == T_SHORT) dst.u16(x); else dst.u32(x); }
        if (bytes < 4) for (unsigned k = bytes; k < 4; k++) dir.u8(0);
    }
};

Ifd build_ifd(unsigned w, unsigned h, bool ycc, bool b16, unsigned extra_at, unsigned strip_at)
{
    Ifd f(extra_at);
    const unsigned bits = b16 ? 16 : 8;
    f.one(0x0100, T_SHORT, w); f.one(0x0101, T_SHORT, h);                        // the reference writes both as SHORT
    f.many(0x0102, T_SHORT, { bits, bits, bits });
    f.one(0x0103, T_SHORT, 1); f.one(0x0106, T_SHORT, ycc ? 6 : 2);
    f.one(0x0111, T_SHORT, strip_at); f.one(0x0112, T_SHORT, 1); f.one(0x0115, T_SHORT, 3); f.one(0x0116, T_SHORT, h);
    f.one(0x0117, T_LONG, h * w * (b16 ? 6u : 3u));
    f.many(0x011A, T_RATIONAL, { 72, 1 }); f.many(0x011B, T_RATIONAL, { 72, 1 });
    f.one(0x011C, T_SHORT, 1); f.one(0x0128, T_SHORT, 2);
    if (ycc) { f.many(0x0211, T_RATIONAL, { 299, 1000, 587, 1000, 114, 1000 }); f.many(0x0212, T_SHORT, { 1, 1 }); f.one(0x0213, T_SHORT, 1); }
    f.many(0x0214, T_RATIONAL, { 0, 1, 0xFF, 1, 0, 1, 0xFF, 1, 0, 1, 0xFF, 1 });
    return f;
}

}  // namespace

extern "C" int jsnoop_export_tiff(JsnoopDecoder* d, const char* path, int mode)
{
    if (!d || !path || mode < 0 || mode > 2) { js_set_error("jsnoop_export_tiff: bad argument"); return -1; }
    if (!d->have_image || !d->preview_is_jpeg) { js_set_error("jsnoop_export_tiff: no decoded image"); return -1; }
    JsnoopBatch* b = d->batch; const JsImage& im = b->imgs[d->img];
    const bool ycc = mode == 2, b16 = mode == 1;
    if (ycc && im.ncomp != 3) { js_set_error("jsnoop_export_tiff: YCC export needs three components"); return -1; }
    const unsigned w = im.img_x, h = im.img_y;
    // layout: 8-byte header, IFD (count + entries + terminator), extra area, pixel strip -- sizes from a dry run, like the reference's first pass
    Ifd dry = build_ifd(w, h, ycc, b16, 0, 0);
    const unsigned extra_at = 8 + 2 + (unsigned)dry.dir.v.size() + 4, strip_at = extra_at + (unsigned)dry.extra.v.size();
    Ifd ifd = build_ifd(w, h, ycc, b16, extra_at, strip_at);
    Bytes head; head.u32(0x4D4D002A); head.u32(8); head.u16(ifd.entries);
    head.v.insert(head.v.end(), ifd.dir.v.begin(), ifd.dir.v.end()); head.u32(0);
    head.v.insert(head.v.end(), ifd.extra.v.begin(), ifd.extra.v.end());
    const size_t strip = (size_t)w * h * (b16 ? 6 : 3);
    if (hipSetDevice(b->device) != hipSuccess) { js_set_error("jsnoop_export_tiff: device error"); return -1; }
    uint8_t* dpack = nullptr;
    if (hipMalloc((void**)&dpack, strip + 64) != hipSuccess) { js_set_error("jsnoop_export_tiff: hipMalloc failed"); return -1; }
    js_launch_tiff_pack(b->stream, b->dev.imgs, (uint32_t)d->img, b->dev.dib, b->dev.planes, mode, dpack);
    std::vector<uint8_t> host(strip);
    const int rc = b->d2h_staged(host.data(), dpack, strip);
    hipFree(dpack);
    if (rc) return -1;
    FILE* f = fopen(path, "wb");
    if (!f) { js_set_error("ERROR: Couldn't open file for write [%s]", path); return -1; }
    const bool ok = fwrite(head.v.data(), 1, head.v.size(), f) == head.v.size() && fwrite(host.data(), 1, strip, f) == strip;
    fclose(f);
    if (!ok) { js_set_error("jsnoop_export_tiff: short write to [%s]", path); return -1; }
    return 0;
}
