// shim_impl.cpp -- TEST INFRASTRUCTURE ONLY.
// Platform-side definitions that the UNMODIFIED reference translation units
// (ImgDecode.cpp, WindowBuf.cpp, General.cpp, compiled in place from
// /root/reference/source) link against: a log sink, the config object with the
// two decode options switched to the north-star mode, a plain heap DIB, and the
// AfxGetApp() singleton.  No reference code is restated here.
#include "stdafx.h"
#include "JPEGsnoop.h"
#include "DocLog.h"
#include "Dib.h"

// ---- log sink: keep lines in memory so the driver can hand them back -------
static std::vector<std::string> g_log;
std::vector<std::string>& ShimLog() { return g_log; }
CDocLog::CDocLog() : m_bUseDoc(false), m_pDoc(nullptr), m_bEn(true), m_bLogQuickMode(false) {}
CDocLog::~CDocLog() {}
void CDocLog::AddLine(CString s)        { if (m_bEn) g_log.push_back(s.s); }
void CDocLog::AddLineHdr(CString s)     { if (m_bEn) g_log.push_back(s.s); }
void CDocLog::AddLineHdrDesc(CString s) { if (m_bEn) g_log.push_back(s.s); }
void CDocLog::AddLineWarn(CString s)    { if (m_bEn) g_log.push_back("W:" + s.s); }
void CDocLog::AddLineErr(CString s)     { if (m_bEn) g_log.push_back("E:" + s.s); }
void CDocLog::AddLineGood(CString s)    { if (m_bEn) g_log.push_back(s.s); }
void CDocLog::Enable()  { m_bEn = true; }
void CDocLog::Disable() { m_bEn = false; }
void CDocLog::SetQuickMode(bool b) { m_bLogQuickMode = b; }
bool CDocLog::GetQuickMode() { return m_bLogQuickMode; }

// ---- config: only the fields the scan decoder reads ------------------------
CSnoopConfig::CSnoopConfig(void)
{
    bInteractive = false; bGuiMode = false;
    bDecodeScanImg = true;
    bDecodeScanImgAc = true;     // "Full IDCT" mode (north-star path)
    bHistoEn = false;            // fast colour path
    bStatClipEn = false;
    bDumpHistoY = false;
    nErrMaxDecodeScan = 20;
    bOutputScanDump = false; bOutputDHTexpand = false; bRelaxedParsing = true;
    bDebugLogEnable = false; fpDebugLog = nullptr;
}
CSnoopConfig::~CSnoopConfig(void) {}
bool CSnoopConfig::DebugLogAdd(CString) { return true; }

// ---- application singleton -------------------------------------------------
static CJPEGsnoopApp g_app;
static CSnoopConfig  g_cfg;
CWinApp* AfxGetApp() { g_app.m_pAppConfig = &g_cfg; return &g_app; }
CSnoopConfig* ShimConfig() { return &g_cfg; }
int AfxMessageBox(const char*, unsigned) { return 0; }

// ---- heap DIB with the Win32 BITMAPINFO layout ------------------------------
CDIB::CDIB() : m_pDIB(nullptr) {}
CDIB::~CDIB() { Kill(); }
void CDIB::Kill() { if (m_pDIB) { delete[] (BYTE*)m_pDIB; m_pDIB = nullptr; } }
bool CDIB::CreateDIB(DWORD w, DWORD h, unsigned short bits)
{
    if (m_pDIB) return false;
    size_t bytes = sizeof(BITMAPINFOHEADER) + sizeof(RGBQUAD) + (size_t)w * h * sizeof(RGBQUAD) + 4;
    BYTE* p = new BYTE[bytes]();
    m_pDIB = (LPBITMAPINFO)p;
    BITMAPINFOHEADER& bh = m_pDIB->bmiHeader;
    bh.biSize = sizeof(BITMAPINFOHEADER); bh.biWidth = (LONG)w; bh.biHeight = (LONG)h;
    bh.biPlanes = 1; bh.biBitCount = bits; bh.biCompression = BI_RGB;
    bh.biXPelsPerMeter = bh.biYPelsPerMeter = 1000;
    return true;
}
int CDIB::GetDIBCols() const { return 0; }          // >8 bpp: no palette
void* CDIB::GetDIBBitArray() const { return m_pDIB ? (BYTE*)m_pDIB + m_pDIB->bmiHeader.biSize : nullptr; }
bool CDIB::CopyDIB(CDC*, int, int, float) { return true; }
bool CDIB::CopyDibDblBuf(CDC*, int, int, CRect*, float) { return true; }
bool CDIB::CopyDIBsmall(CDC*, int, int, float) { return true; }
bool CDIB::CopyDibPart(CDC*, CRect, CRect*, float) { return true; }
