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
        core.BatchGetBitmap(2, d2, bx, by);
        printf("batch count=%u size=%ux%u dib_fnv=%016llx\n", core.GetBatchFileCount(), bx, by, (unsigned long long)fnv(d2.data(), d2.size()));
        {   // the per-file pass of the batch loop: log text and decoder side outputs of file 1, next to the single-file decoder's
            CJPEGsnoopCoreGpu logcore;
            logcore.BatchSetOptions(true, true); logcore.BatchEnableLog();
            for (int i = 0; i < 2; i++) logcore.BatchAddFile(bytes.data(), bytes.size());
            std::vector<std::string> lines; std::vector<uint32_t> mm((size_t)(x / 8) * (y / 8) + 16); unsigned st8[8] = {0}; int ba[10] = {0};
            const bool ok = logcore.DoBatchProcess() && logcore.BatchGetLog(1, lines) && logcore.BatchGetSideOutputs(1, mm.data(), nullptr, nullptr, nullptr, nullptr, st8, ba);
            size_t fin = 0; for (size_t i = 0; i < lines.size(); i++) if (lines[i].find("Finished Decoding SCAN Data") != std::string::npos) fin = i;
            printf("batchlog ok=%d lines=%zu first=[%s] finished_at=%zu mcu0=%u.%u pixels=%u bright_valid=%d\n", (int)ok, lines.size(), lines.empty() ? "" : lines[0].c_str(), fin,
                   mm[0] >> 4, mm[0] & 7, st8[3], ba[0]);
        }
        // the core facade: AnalyzeFile-shaped entry, I_* accessors, a byte overlay (the reference's fault-injection tool) and the re-decode
        if (!core.AnalyzeFile(argv[1]) || !core.IsAnalyzed()) { printf("analyze_failed %s\n", jsnoop_last_error()); return 1; }
        unsigned cx = 0, cy = 0; core.I_GetImageSize(cx, cy);
        unsigned char* cdib = nullptr; core.I_GetBitmapPtr(cdib);
        unsigned mx, my, kx, ky; core.I_PixelToMcu(100, 50, mx, my); core.I_PixelToBlk(100, 50, kx, ky);
        unsigned fb, fbit; core.I_LookupFilePosPix(100, 50, fb, fbit);
        int ly, lcb, lcr; core.I_LookupBlkYCC(kx, ky, ly, lcb, lcr);
        short *cyp, *ccb, *ccr; core.I_GetPixMapPtrs(cyp, ccb, ccr);
        printf("core ready=%d size=%ux%u dib_fnv=%016llx mcu=%u,%u lin=%u blk=%u,%u pos=%u.%u ycc=%d,%d,%d dqt=%u mode=%u y0=%d\n", (int)core.I_IsPreviewReady(), cx, cy,
               (unsigned long long)(cdib ? fnv(cdib, (size_t)cx * cy * 4) : 0), mx, my, core.I_McuXyToLinear(mx, my), kx, ky, fb, fbit, ly, lcb, lcr,
               core.I_GetDqtEntry(0, 0), core.I_GetPreviewMode(), cyp ? cyp[0] : 0);
        if (argc > 3) {                                          // overlay: argv[2] = file offset, argv[3] = hex bytes
            std::vector<uint8_t> ov; for (const char* h = argv[3]; h[0] && h[1]; h += 2) { unsigned v; sscanf(h, "%2x", &v); ov.push_back((uint8_t)v); }
            const unsigned at = (unsigned)strtoul(argv[2], nullptr, 0);
            const bool ok = core.B_OverlayInstall(0, ov.data(), (unsigned)ov.size(), at);
            const uint8_t seen = core.B_Buf(at), clean = core.B_Buf(at, true);
            core.ReprocessFile();
            core.I_GetBitmapPtr(cdib);
            printf("overlay installed=%d seen=%02x clean=%02x dib_fnv=%016llx\n", (int)ok, seen, clean, (unsigned long long)(cdib ? fnv(cdib, (size_t)cx * cy * 4) : 0));
            core.B_OverlayRemoveAll(); core.ReprocessFile(); core.I_GetBitmapPtr(cdib);
            printf("restored dib_fnv=%016llx\n", (unsigned long long)(cdib ? fnv(cdib, (size_t)cx * cy * 4) : 0));
        }
    } catch (const std::exception& e) { printf("exception %s\n", e.what()); return 3; }
    return 0;
}
