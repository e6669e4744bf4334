// wrapper_demo.cpp -- drives the CimgDecode-shaped C++ wrapper (jpegsnoop_amd/csrc/ImgDecodeGpu.h)
// the way a JPEGsnoop maintainer would: header walk -> DecodeScanImg -> GetBitmapPtr / GetPixMapPtrs,
// then the same file through the batched core.  Prints one line per result for the Python test.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../jpegsnoop_amd/csrc/ImgDecodeGpu.h"

static uint64_t fnv(const uint8_t* p, size_t n) { uint64_t h = 0xcbf29ce484222325ull; for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; } return h; }

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: wrapper_demo file.jpg\n"); return 2; }
    FILE* f = fopen(argv[1], "rb"); if (!f) { perror("open"); return 2; }
    std::vector<uint8_t> bytes; uint8_t buf[65536]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) bytes.insert(bytes.end(), buf, buf + n);
    fclose(f);
    try {
        CwindowBufView wbuf; wbuf.pData = bytes.data(); wbuf.nLen = bytes.size();
        int nerr = 0;
        CimgDecodeGpu dec([&](int lvl, const std::string& s) { if (lvl == 2) { nerr++; fprintf(stderr, "log: %s\n", s.c_str()); } }, &wbuf);
        unsigned nStart = 0;
        if (!dec.WalkJfifHeader(nStart)) { printf("walk_failed %s\n", jsnoop_last_error()); return 1; }
        dec.SetConfig(true);
        dec.DecodeScanImg(nStart, true, false);
        unsigned x = 0, y = 0; dec.GetImageSize(x, y);
        unsigned char* dib = nullptr; dec.GetBitmapPtr(dib);
        short *py, *pcb, *pcr; dec.GetPixMapPtrs(py, pcb, pcr);
        unsigned byte0 = 0, bit0 = 0; dec.LookupFilePosMcu(0, 0, byte0, bit0);
        printf("single ready=%d size=%ux%u dib_fnv=%016llx y0=%d mcu0=%u.%u errs=%d\n", (int)dec.IsPreviewReady(), x, y,
               (unsigned long long)(dib ? fnv(dib, (size_t)x * y * 4) : 0), py ? py[0] : 0, byte0, bit0, nerr);
        CJPEGsnoopCoreGpu core;
        for (int i = 0; i < 3; i++) core.BatchAddFile(bytes.data(), bytes.size());
        if (!core.DoBatchProcess()) { printf("batch_failed %s\n", jsnoop_last_error()); return 1; }
        std::vector<uint8_t> d2; unsigned bx, by;
        core.I_GetBitmap(2, d2, bx, by);
        printf("batch count=%u size=%ux%u dib_fnv=%016llx\n", core.GetBatchFileCount(), bx, by, (unsigned long long)fnv(d2.data(), d2.size()));
    } catch (const std::exception& e) { printf("exception %s\n", e.what()); return 3; }
    return 0;
}
