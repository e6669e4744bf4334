// ImgDecodeGpu.h -- the reference's scan-decoder interface on top of the C ABI (include/jsnoop_gpu.h).
//
// `CimgDecodeGpu` keeps the public method names, argument meaning and return conventions of the
// reference's `CimgDecode` (reference source/ImgDecode.h:286-356, :384-385, :407-425) for the
// scan-decode path, so that `CjfifDecode` / `CJPEGsnoopCore` code that drives a `CimgDecode*` can be
// pointed at this class (INTEGRATION.md shows the two-line change).  `CJPEGsnoopCoreGpu` carries the
// `I_*` pass-throughs of `CJPEGsnoopCore` (source/JPEGsnoopCore.h:79-117) that belong to this path,
// plus the batched submit that replaces the sequential DoBatchFileProcess loop
// (source/JPEGsnoopCore.cpp:765-845).
//
// Header-only, no MFC: CString/CDocLog/CwindowBuf are replaced by std::string, a log callback and a
// (pointer, length) view of the file bytes.  Everything below is a thin forwarding layer; the work
// happens in libjsnoop_gpu.so on the GPU.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/jsnoop_gpu.h"

// What CwindowBuf hands the decoder: the whole file image (overlays already applied by the caller).
struct CwindowBufView {
    const uint8_t* pData = nullptr; size_t nLen = 0;
    uint8_t Buf(size_t nOffset) const { return nOffset < nLen ? pData[nOffset] : 0; }     // WindowBuf.cpp:639
};

// The slice of CDIB (source/Dib.h:33-55) that reaches the decoder: the preview DIB CjfifDecode lets the Photoshop decoder fill
// (m_pPsDec->DecodePsd(nStartPos, &m_pImgDec->m_pDibTemp, ...), source/JfifDecode.cpp:7369).  The bits live in the decoder
// (jsnoop_dib_temp_create), so GetBitmapPtr hands them out as the reference's would.
class CDibGpu {
public:
    void  Kill() { if (m_h) jsnoop_dib_temp_create(m_h, 0, 0); m_pBits = nullptr; m_nW = m_nH = 0; }                                     // Dib.cpp:88
    bool  CreateDIB(unsigned long dwWidth, unsigned long dwHeight, unsigned short nBits)                                               // Dib.cpp:53: 32 bits per pixel is what this path creates
    { if (!m_h || nBits != 32) return false; m_pBits = jsnoop_dib_temp_create(m_h, (unsigned)dwWidth, (unsigned)dwHeight); m_nW = (unsigned)dwWidth; m_nH = (unsigned)dwHeight; return m_pBits != nullptr; }
    void* GetDIBBitArray() const { return m_pBits; }                                                                                   // Dib.cpp:131
    unsigned GetWidth() const { return m_nW; }
    unsigned GetHeight() const { return m_nH; }
    void  Attach(JsnoopDecoder* h) { m_h = h; }
    void  Invalidate() { m_pBits = nullptr; m_nW = m_nH = 0; }                                                                         // the decoder dropped the bits (Reset, DecodeScanImg): GetDIBBitArray is NULL as after Kill
private:
    JsnoopDecoder* m_h = nullptr; uint8_t* m_pBits = nullptr; unsigned m_nW = 0, m_nH = 0;
};

// TDib: the type of the public member m_pDibTemp -- CDibGpu here; a build inside the reference's tree may name its own CDIB-shaped class
// (it needs Kill / CreateDIB / GetDIBBitArray, an Attach(JsnoopDecoder*) hook and Invalidate(): called when the decoder has dropped the bits).
template <class TDib>
class CimgDecodeGpuT {
public:
    using LogFn = std::function<void(int /*0 info, 1 warn, 2 err*/, const std::string&)>;

    explicit CimgDecodeGpuT(LogFn pLog = nullptr, const CwindowBufView* pWBuf = nullptr) : m_pWBuf(pWBuf), m_log(std::move(pLog))
    {
        m_h = jsnoop_create();                                   // CimgDecode ctor (ImgDecode.cpp:142)
        if (!m_h) throw std::runtime_error(std::string("jsnoop_create: ") + jsnoop_last_error());
        if (m_log) jsnoop_set_log_callback(m_h, &CimgDecodeGpuT::LogThunk, this);
        m_pDibTemp.Attach(m_h);
    }
    ~CimgDecodeGpuT() { jsnoop_destroy(m_h); }
    CimgDecodeGpuT(const CimgDecodeGpuT&) = delete;
    CimgDecodeGpuT& operator=(const CimgDecodeGpuT&) = delete;

    // ---- the three public members CjfifDecode writes for a preview that is not the scan decoder's (source/ImgDecode.h:507-510, "FIXME (workaround)"
    //      there too; set at source/JfifDecode.cpp:7369-7373).  Plain members like the reference's: SyncPreviewMembers() carries them into the
    //      library -- IsPreviewReady / GetBitmapPtr call it, so poking them and asking is all a caller does.
    bool m_bDibTempReady = false;
    TDib m_pDibTemp;
    bool m_bPreviewIsJpeg = false;
    void SyncPreviewMembers() { jsnoop_set_dib_temp_ready(m_h, m_bDibTempReady); jsnoop_set_preview_is_jpeg(m_h, m_bPreviewIsJpeg); }

    void SetWindowBuf(const CwindowBufView* pWBuf) { m_pWBuf = pWBuf; }

    // ---- lifecycle ------------------------------------------------------------------------------
    void Reset() { SyncPreviewMembers(); jsnoop_reset(m_h); m_bDibTempReady = false; m_pDibTemp.Invalidate(); }   // :49 (kills m_pDibTemp when it was ready, :80-83 -- "ready" is the member the caller may just have set)
    void ResetState() { jsnoop_reset_state(m_h); }              // :286
    void ResetDqtTables() { jsnoop_reset_dqt_tables(m_h); }     // :343 (private in the reference: ResetState calls it)
    void ResetDhtLookup() { jsnoop_reset_dht_lookup(m_h); }     // :373

    // ---- options: the CSnoopConfig fields DecodeScanImg reads (:2730-2741) -------------------------
    void SetConfig(bool bDecodeScanImgAc, bool bHistoEn = false, bool bStatClipEn = false, unsigned nErrMaxDecodeScan = 20, bool bDumpHistoY = false)
    { jsnoop_set_options(m_h, bDecodeScanImgAc, bHistoEn, bStatClipEn, nErrMaxDecodeScan); jsnoop_set_dump_histo_y(m_h, bDumpHistoY); }

    // ---- tables / geometry (same names, same bool returns) -------------------------------------------
    bool SetDqtEntry(unsigned nTblDestId, unsigned nCoeffInd, unsigned nCoeffIndZz, unsigned short nCoeffVal)
    { return jsnoop_set_dqt_entry(m_h, nTblDestId, nCoeffInd, nCoeffIndZz, nCoeffVal) != 0; }                 // :424
    bool SetDqtTables(unsigned nCompInd, unsigned nTbl) { return jsnoop_set_dqt_tables(m_h, nCompInd, nTbl) != 0; }          // :505
    unsigned GetDqtEntry(unsigned nTblDestId, unsigned nCoeffInd) { return jsnoop_get_dqt_entry(m_h, nTblDestId, nCoeffInd); } // :466
    bool SetDhtTables(unsigned nCompInd, unsigned nTblDc, unsigned nTblAc) { return jsnoop_set_dht_tables(m_h, nCompInd, nTblDc, nTblAc) != 0; } // :536
    bool SetDhtEntry(unsigned nDestId, unsigned nClass, unsigned nInd, unsigned nLen, unsigned nBits, unsigned nMask, unsigned nCode)
    { return jsnoop_set_dht_entry(m_h, nDestId, nClass, nInd, nLen, nBits, nMask, nCode) != 0; }                // :748
    bool SetDhtSize(unsigned nDestId, unsigned nClass, unsigned nSize) { return jsnoop_set_dht_size(m_h, nDestId, nClass, nSize) != 0; } // :834
    void SetPrecision(unsigned nPrecision) { jsnoop_set_precision(m_h, nPrecision); }                           // :564
    void SetSofSampFactors(unsigned nCompInd, unsigned nSampFactH, unsigned nSampFactV) { jsnoop_set_sof_samp_factors(m_h, nCompInd, nSampFactH, nSampFactV); } // :619
    void SetImageDetails(unsigned nDimX, unsigned nDimY, unsigned nCompsSOF, unsigned nCompsSOS, bool bRstEn, unsigned nRstInterval)
    { jsnoop_set_image_details(m_h, nDimX, nDimY, nCompsSOF, nCompsSOS, bRstEn, nRstInterval); }                // :590
    void SetImageDimensions(unsigned nWidth, unsigned nHeight) { jsnoop_set_image_dimensions(m_h, nWidth, nHeight); }          // :2706
    void GetImageDimensions(unsigned& nWidth, unsigned& nHeight) { jsnoop_get_image_dimensions(m_h, &nWidth, &nHeight); }      // (m_rectImgBase, read by the view code)

    // ---- minimal header walk: the subset of CjfifDecode::DecodeMarker that feeds this object --------------
    bool WalkJfifHeader(unsigned& nPosScanStart)
    { return m_pWBuf && jsnoop_jfif_walk(m_h, m_pWBuf->pData, m_pWBuf->nLen, &nPosScanStart) == 0; }

    // ---- the hot path -----------------------------------------------------------------------------------
    void DecodeScanImg(unsigned nStart, bool bDisplay, bool bQuiet)                                              // :2723
    {
        if (!m_pWBuf || !m_pWBuf->pData) { if (m_log) m_log(2, "*** ERROR: DecodeScanImg without a file buffer ***"); return; }
        jsnoop_decode_scan_img(m_h, m_pWBuf->pData, m_pWBuf->nLen, nStart, bDisplay, bQuiet);
        m_pDibTemp.Invalidate();                                                                                 // (the decode clears the temporary preview, :2976-2978)
        m_bPreviewIsJpeg = jsnoop_is_preview_ready(m_h) != 0; m_bDibTempReady = jsnoop_get_dib_temp_ready(m_h) != 0;     // :3647-3648
    }

    // ---- results (owned by the decoder, valid until the next Reset / DecodeScanImg / destruction) ------------
    bool IsPreviewReady() { SyncPreviewMembers(); return jsnoop_is_preview_ready(m_h) != 0; }                    // :3753 (returns m_bPreviewIsJpeg)
    void GetImageSize(unsigned& nX, unsigned& nY) { jsnoop_get_image_size(m_h, &nX, &nY); }                      // :4929
    void GetBitmapPtr(unsigned char*& pBitmap) { SyncPreviewMembers(); pBitmap = const_cast<unsigned char*>(jsnoop_get_bitmap_ptr(m_h)); } // :4940
    const void* GetBitmapDevicePtr() { return jsnoop_get_bitmap_dev(m_h); }                                      // HBM copy of the DIB
    void GetPixMapPtrs(short*& pMapY, short*& pMapCb, short*& pMapCr)                                            // :4913
    {
        const int16_t *y, *cb, *cr; jsnoop_get_pixmap_ptrs(m_h, &y, &cb, &cr);
        pMapY = const_cast<short*>(y); pMapCb = const_cast<short*>(cb); pMapCr = const_cast<short*>(cr);
    }
    void LookupFilePosPix(unsigned nPixX, unsigned nPixY, unsigned& nByte, unsigned& nBit) { jsnoop_lookup_file_pos_pix(m_h, nPixX, nPixY, &nByte, &nBit); } // :5001
    void LookupFilePosMcu(unsigned nMcuX, unsigned nMcuY, unsigned& nByte, unsigned& nBit) { jsnoop_lookup_file_pos_mcu(m_h, nMcuX, nMcuY, &nByte, &nBit); } // :5020
    void LookupBlkYCC(unsigned nBlkX, unsigned nBlkY, int& nY, int& nCb, int& nCr) { jsnoop_lookup_blk_ycc(m_h, nBlkX, nBlkY, &nY, &nCb, &nCr); }            // :5037
    // bHistoEn / bStatClipEn statistics (m_sHisto, m_sStatClip, m_anCcHisto_r/g/b, m_anHistoYFull; ImgDecode.h:220-279, :656-661)
    // as one JSNOOP_STATS_WORDS record, see include/jsnoop_gpu.h
    void GetColorStats(uint32_t* pStats) { jsnoop_get_color_stats(m_h, pStats); }
    // Export to TIFF (CJPEGsnoopDoc::OnToolsExporttiff + FileTiff::WriteFile): nMode 0 RGB8, 1 RGB16, 2 YCC8
    bool ExportTiff(const std::string& strFnameOut, int nMode) { return jsnoop_export_tiff(m_h, strFnameOut.c_str(), nMode) == 0; }
    // hover helpers next to LookupFilePosMcu (source/JPEGsnoopViewImg.cpp:294-334); CPoint is a pair of unsigned here
    void PixelToMcu(unsigned nPixX, unsigned nPixY, unsigned& nMcuX, unsigned& nMcuY) { jsnoop_pixel_to_mcu(m_h, nPixX, nPixY, &nMcuX, &nMcuY); }               // :5056
    void PixelToBlk(unsigned nPixX, unsigned nPixY, unsigned& nBlkX, unsigned& nBlkY) { jsnoop_pixel_to_blk(m_h, nPixX, nPixY, &nBlkX, &nBlkY); }               // :5071
    unsigned McuXyToLinear(unsigned nMcuX, unsigned nMcuY) { return jsnoop_mcu_xy_to_linear(m_h, nMcuX, nMcuY); }                                              // :5088
    // CwindowBuf overlays (source/WindowBuf.cpp:516-620): the next DecodeScanImg reads the patched bytes, as CwindowBuf::Buf would hand them out
    bool OverlayInstall(unsigned /*nOvrInd*/, const uint8_t* pOverlay, unsigned nLen, unsigned nBegin) { return jsnoop_overlay_install(m_h, pOverlay, nLen, nBegin) != 0; }
    void OverlayRemoveAll() { jsnoop_overlay_remove_all(m_h); }
    unsigned OverlayGetNum() { return jsnoop_overlay_get_num(m_h); }
    bool OverlayGet(unsigned nOvrInd, const uint8_t*& pOverlay, unsigned& nLen, unsigned& nBegin) { return jsnoop_overlay_get(m_h, nOvrInd, &pOverlay, &nLen, &nBegin) != 0; }
    unsigned PackFileOffset(unsigned nByte, unsigned nBit) const { return (nByte << 4) + nBit; }                 // :5104
    void UnpackFileOffset(unsigned nPacked, unsigned& nByte, unsigned& nBit) const { nBit = nPacked & 0x7; nByte = nPacked >> 4; } // :5123

    // ---- preview re-render on the retained data --------------------------------------------------------------
    void SetPreviewMode(unsigned nMode) { jsnoop_set_preview_mode(m_h, nMode); }                                 // :633
    unsigned GetPreviewMode() { return jsnoop_get_preview_mode(m_h); }
    void SetPreviewYccOffset(unsigned nMcuX, unsigned nMcuY, int nY, int nCb, int nCr) { jsnoop_set_preview_ycc_offset(m_h, nMcuX, nMcuY, nY, nCb, nCr); } // :650

    void GetPreviewYccOffset(unsigned& nMcuX, unsigned& nMcuY, int& nY, int& nCb, int& nCr) { jsnoop_get_preview_ycc_offset(m_h, &nMcuX, &nMcuY, &nY, &nCb, &nCr); }   // :670
    void SetPreviewMcuInsert(unsigned nMcuX, unsigned nMcuY, int nLen) { jsnoop_set_preview_mcu_insert(m_h, nMcuX, nMcuY, nLen); }                                        // :682
    void GetPreviewMcuInsert(unsigned& nMcuX, unsigned& nMcuY, unsigned& nLen) { jsnoop_get_preview_mcu_insert(m_h, &nMcuX, &nMcuY, &nLen); }                             // :693

    JsnoopDecoder* Handle() { return m_h; }

private:
    static void LogThunk(void* user, int level, const char* text) { auto* self = static_cast<CimgDecodeGpuT*>(user); if (self->m_log) self->m_log(level, text); }
    JsnoopDecoder* m_h = nullptr;
    const CwindowBufView* m_pWBuf;
    LogFn m_log;
};
using CimgDecodeGpu = CimgDecodeGpuT<CDibGpu>;

// The slice of CJPEGsnoopCore that belongs to the scan-decode path (source/JPEGsnoopCore.h:79-117, JPEGsnoopCore.cpp:1211-1399):
// the data-bearing I_* accessors over the core's one CimgDecode (source/JPEGsnoopCore.cpp:46), an AnalyzeFile-shaped entry
// (source/JPEGsnoopCore.cpp:325: open, analyse, close) fed from a file or from bytes, the B_Overlay* pass-throughs of its
// CwindowBuf, and a batched replacement for the strictly sequential per-file loop (DoBatchFileProcess :765-845).  Not carried:
// the status-bar / zoom / marker / GDI accessors (I_SetStatusBar, I_*Zoom*, I_*Marker*, I_ViewOnDraw, I_*StatusText) -- Windows
// GUI state that never reaches the decoder -- and the J_* metadata-report accessors (out of scope).
class CJPEGsnoopCoreGpu {
public:
    explicit CJPEGsnoopCoreGpu(CimgDecodeGpu::LogFn pLog = nullptr) : m_pImgDec(new CimgDecodeGpu(std::move(pLog), &m_view))
    { m_b = jsnoop_batch_create(nullptr); if (!m_b) { delete m_pImgDec; throw std::runtime_error(std::string("jsnoop_batch_create: ") + jsnoop_last_error()); } }
    ~CJPEGsnoopCoreGpu() { jsnoop_batch_destroy(m_b); delete m_pImgDec; }
    CJPEGsnoopCoreGpu(const CJPEGsnoopCoreGpu&) = delete;
    CJPEGsnoopCoreGpu& operator=(const CJPEGsnoopCoreGpu&) = delete;

    // ---- AnalyzeFile :325 (AnalyzeOpen, AnalyzeFileDo, AnalyzeClose): header walk with the built-in front end, then DecodeScanImg(nStart, true, false) as
    //      CjfifDecode::DecodeMarker issues it (source/JfifDecode.cpp:5299).  The file image stays owned by the core until the next Analyze*.
    bool AnalyzeBuffer(const uint8_t* pFile, size_t nLen)
    {
        m_file.assign(pFile, pFile + nLen); m_view.pData = m_file.data(); m_view.nLen = m_file.size(); m_bFileAnalyzed = false;
        if (nLen == 0) return false;                             // "ERROR: File length is zero, no decoding done." (:301)
        m_pImgDec->ResetState(); m_pImgDec->Reset();             // CjfifDecode::Reset (source/JfifDecode.cpp:7306-7308)
        unsigned nStart = 0;
        if (!m_pImgDec->WalkJfifHeader(nStart)) return false;
        m_pImgDec->DecodeScanImg(nStart, true, false);
        m_bFileAnalyzed = true;
        return true;
    }
    bool AnalyzeFile(const std::string& strFname)
    {
        FILE* f = fopen(strFname.c_str(), "rb");                 // AnalyzeOpen :157: FALSE when the file cannot be opened
        if (!f) return false;
        std::vector<uint8_t> buf; uint8_t tmp[65536]; size_t n;
        while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
        fclose(f);
        m_strPathName = strFname;
        AnalyzeBuffer(buf.data(), buf.size());
        return true;                                             // the status of opening the file, like the reference
    }
    bool IsAnalyzed() const { return m_bFileAnalyzed; }          // :138

    // ---- I_* accessor wrappers for CimgDecode (:1211-1399) ---------------------------------------------------------------------
    unsigned I_GetDqtEntry(unsigned nTblDestId, unsigned nCoeffInd) { return m_pImgDec->GetDqtEntry(nTblDestId, nCoeffInd); }                       // :1216
    void     I_SetPreviewMode(unsigned nMode) { m_pImgDec->SetPreviewMode(nMode); }                                                                 // :1221
    unsigned I_GetPreviewMode() { return m_pImgDec->GetPreviewMode(); }                                                                             // :1226
    void     I_SetPreviewYccOffset(unsigned nMcuX, unsigned nMcuY, int nY, int nCb, int nCr) { m_pImgDec->SetPreviewYccOffset(nMcuX, nMcuY, nY, nCb, nCr); }   // :1231
    void     I_GetPreviewYccOffset(unsigned& nMcuX, unsigned& nMcuY, int& nY, int& nCb, int& nCr) { m_pImgDec->GetPreviewYccOffset(nMcuX, nMcuY, nY, nCb, nCr); } // :1236
    void     I_SetPreviewMcuInsert(unsigned nMcuX, unsigned nMcuY, int nLen) { m_pImgDec->SetPreviewMcuInsert(nMcuX, nMcuY, nLen); }               // :1241
    void     I_GetPreviewMcuInsert(unsigned& nMcuX, unsigned& nMcuY, unsigned& nLen) { m_pImgDec->GetPreviewMcuInsert(nMcuX, nMcuY, nLen); }       // :1246
    void     I_PixelToMcu(unsigned nPixX, unsigned nPixY, unsigned& nMcuX, unsigned& nMcuY) { m_pImgDec->PixelToMcu(nPixX, nPixY, nMcuX, nMcuY); } // :1276
    void     I_PixelToBlk(unsigned nPixX, unsigned nPixY, unsigned& nBlkX, unsigned& nBlkY) { m_pImgDec->PixelToBlk(nPixX, nPixY, nBlkX, nBlkY); } // :1281
    unsigned I_McuXyToLinear(unsigned nMcuX, unsigned nMcuY) { return m_pImgDec->McuXyToLinear(nMcuX, nMcuY); }                                    // :1286
    void     I_GetImageSize(unsigned& nX, unsigned& nY) { m_pImgDec->GetImageSize(nX, nY); }                                                       // :1291
    void     I_GetPixMapPtrs(short*& pMapY, short*& pMapCb, short*& pMapCr) { m_pImgDec->GetPixMapPtrs(pMapY, pMapCb, pMapCr); }                   // :1296
    void     I_GetBitmapPtr(unsigned char*& pBitmap) { m_pImgDec->GetBitmapPtr(pBitmap); }                                                         // :1361: the decoder-owned pointer, nothing copied
    void     I_LookupFilePosMcu(unsigned nMcuX, unsigned nMcuY, unsigned& nByte, unsigned& nBit) { m_pImgDec->LookupFilePosMcu(nMcuX, nMcuY, nByte, nBit); } // :1366
    void     I_LookupFilePosPix(unsigned nPixX, unsigned nPixY, unsigned& nByte, unsigned& nBit) { m_pImgDec->LookupFilePosPix(nPixX, nPixY, nByte, nBit); } // :1371
    void     I_LookupBlkYCC(unsigned nBlkX, unsigned nBlkY, int& nY, int& nCb, int& nCr) { m_pImgDec->LookupBlkYCC(nBlkX, nBlkY, nY, nCb, nCr); } // :1376
    bool     I_IsPreviewReady() { return m_pImgDec->IsPreviewReady(); }                                                                           // :1396
    const void* I_GetBitmapDevicePtr() { return m_pImgDec->GetBitmapDevicePtr(); }                                                                 // (HBM copy, no reference counterpart)

    // ---- B_* accessor wrappers for CwindowBuf that reach the decoder: byte fetch and overlays (:1180-1207) ------------------------
    uint8_t  B_Buf(unsigned long nOffset, bool bClean = false)                                                                                     // :1180 -> CwindowBuf::Buf (source/WindowBuf.cpp:639)
    {
        if (!bClean) {
            const uint8_t* p; unsigned n, b; uint8_t v = 0; bool hit = false;
            for (unsigned i = 0; i < m_pImgDec->OverlayGetNum(); i++)
                if (m_pImgDec->OverlayGet(i, p, n, b) && nOffset >= b && nOffset < (unsigned long)b + n) { v = p[nOffset - b]; hit = true; }
            if (hit) return v;
        }
        return m_view.Buf(nOffset);
    }
    bool     B_OverlayInstall(unsigned nOvrInd, const uint8_t* pOverlay, unsigned nLen, unsigned nBegin, unsigned /*nMcuX*/ = 0, unsigned /*nMcuY*/ = 0,
                              unsigned /*nMcuLen*/ = 0, unsigned /*nMcuLenIns*/ = 0, int /*nAdjY*/ = 0, int /*nAdjCb*/ = 0, int /*nAdjCr*/ = 0)
    { return m_pImgDec->OverlayInstall(nOvrInd, pOverlay, nLen, nBegin); }                                                                         // :1191
    void     B_OverlayRemoveAll() { m_pImgDec->OverlayRemoveAll(); }                                                                              // :1198
    bool     B_OverlayGet(unsigned nOvrInd, const uint8_t*& pOverlay, unsigned& nLen, unsigned& nBegin) { return m_pImgDec->OverlayGet(nOvrInd, pOverlay, nLen, nBegin); } // :1203
    // re-run the scan decode of the analysed file on the (re-)patched bytes: what the reference's "Tools -> File overlay" does after an install
    bool     ReprocessFile() { if (!m_bFileAnalyzed) return false; const std::vector<uint8_t> copy(m_file); return AnalyzeBuffer(copy.data(), copy.size()); }

    // ---- batch: N files -> N DIBs resident in HBM (per-file semantics of DoBatchFileProcess :765-845 preserved) -------------------
    void     BatchClear() { jsnoop_batch_clear(m_b); }
    int      BatchAddFile(const uint8_t* pFile, size_t nLen) { return jsnoop_batch_add_jpeg(m_b, pFile, nLen); }
    unsigned GetBatchFileCount() const { return (unsigned)jsnoop_batch_count(m_b); }                                                             // :680
    bool     DoBatchProcess() { return jsnoop_batch_upload(m_b) == 0 && jsnoop_batch_decode(m_b) == 0 && jsnoop_batch_sync(m_b) == 0; }
    bool     BatchSetSplit(int nParts) { return jsnoop_batch_set_split(m_b, nParts) == 0; }   // 0: the library decides (default), 1: one stream, 2: the halves of the batch on two streams side by side (same results)
    bool     BatchSetTuning(const JsnoopTuning& t) { return jsnoop_batch_set_tuning(m_b, &t) == 0; }   // how the batch decodes, never what it produces (jsnoop_tuning_defaults fills a struct)
    const void* BatchBitmapDevicePtr(int nFileInd) const { return jsnoop_batch_dib_dev(m_b, nFileInd); }
    bool     BatchGetBitmap(int nFileInd, std::vector<uint8_t>& dib, unsigned& nX, unsigned& nY)
    {
        unsigned info[16]; if (jsnoop_batch_image_info(m_b, nFileInd, info)) return false;
        nX = info[2]; nY = info[3]; dib.resize((size_t)nX * nY * 4);
        return jsnoop_batch_read_dib(m_b, nFileInd, dib.data()) == 0;
    }
    // ---- what the per-file pass of DoBatchFileProcess leaves behind besides pixels (:805-808 DoLogSave; log body ImgDecode.cpp:3021-3745)
    void     BatchSetOptions(bool bDecodeScanImgAc, bool bKeepPixMaps) { jsnoop_batch_set_options(m_b, bDecodeScanImgAc, bKeepPixMaps, 0); }   // before DoBatchProcess
    bool     BatchEnableLog(bool bOn = true) { return jsnoop_batch_enable_log(m_b, bOn) == 0; }                                               // before DoBatchProcess
    // the DecodeScanImg text of file nFileInd, one std::string per CDocLog line ("W:" / "E:" prefixed like AddLineWarn / AddLineErr colour them)
    bool     BatchGetLog(int nFileInd, std::vector<std::string>& lines, bool bHistoEn = false, bool bStatClipEn = false, bool bQuiet = false)
    {
        lines.clear();
        auto thunk = [](void* u, int lvl, const char* txt) { static_cast<std::vector<std::string>*>(u)->push_back(std::string(lvl == 2 ? "E:" : lvl == 1 ? "W:" : "") + txt); };
        return jsnoop_batch_log(m_b, nFileInd, bHistoEn, bStatClipEn, bQuiet, thunk, &lines) == 0;
    }
    // DoLogSave (:807 -> source/JPEGsnoopCore.cpp:150-220) for the scan-decode part of the report: the lines above into a text file
    bool     BatchLogSave(int nFileInd, const std::string& strLogName, bool bHistoEn = false)
    {
        std::vector<std::string> lines; if (!BatchGetLog(nFileInd, lines, bHistoEn)) return false;
        FILE* f = fopen(strLogName.c_str(), "w"); if (!f) return false;
        for (const std::string& l : lines) fprintf(f, "%s\n", l.c_str());
        return fclose(f) == 0;
    }
    // m_pMcuFileMap / m_pBlkDcVal* / m_anDhtHisto / scan status / brightest pixel + average Y of file nFileInd (any pointer may be null)
    bool     BatchGetSideOutputs(int nFileInd, uint32_t* pMcuFileMap, int16_t* pBlkDcY, int16_t* pBlkDcCb, int16_t* pBlkDcCr, uint32_t* pDhtHisto, unsigned* pStatus8, int* pBrightAvg10)
    { return jsnoop_batch_side_outputs(m_b, nFileInd, pMcuFileMap, pBlkDcY, pBlkDcCb, pBlkDcCr, pDhtHisto, pStatus8, pBrightAvg10) == 0; }
    bool     BatchExportTiff(int nFileInd, const std::string& strFname, int nMode = 0) { return jsnoop_batch_export_tiff(m_b, nFileInd, strFname.c_str(), nMode) == 0; }
    CimgDecodeGpu* ImgDec() { return m_pImgDec; }
    JsnoopBatch* Handle() { return m_b; }
private:
    CwindowBufView m_view; std::vector<uint8_t> m_file; std::string m_strPathName; bool m_bFileAnalyzed = false;
    CimgDecodeGpu* m_pImgDec; JsnoopBatch* m_b = nullptr;
};
