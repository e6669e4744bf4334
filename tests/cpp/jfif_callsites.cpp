// Boundary check: every way CjfifDecode reaches into its CimgDecode -- the twelve live methods and the three
// public members it writes -- expressed against CimgDecodeGpu with the argument types the reference's call sites have (its locals are `unsigned`,
// `bool`, `unsigned short` table entries).  Each function below names the call site it stands for (reference source/JfifDecode.cpp:line); the one
// commented-out call (ResetImageContent, :4889) is not part of the surface.  Built by tests/test_cpp_wrapper.py with plain g++; run with a file
// argument on a GPU box it also drives the members through a PSD-style preview and a scan decode.
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../jpegsnoop_amd/csrc/ImgDecodeGpu.h"

struct CDecodePsStub {               // CDecodePs::DecodePsd(nPos, CDIB* pDibTemp, nWidth, nHeight), source/DecodePs.cpp:90: fills the preview DIB it is handed
    bool DecodePsd(unsigned long /*nPos*/, CDibGpu* pDibTemp, unsigned& nWidth, unsigned& nHeight)
    {
        nWidth = 5; nHeight = 3;
        pDibTemp->Kill();
        if (!pDibTemp->CreateDIB(nWidth, nHeight, 32)) return false;
        unsigned char* p = static_cast<unsigned char*>(pDibTemp->GetDIBBitArray());
        for (unsigned i = 0; i < nWidth * nHeight * 4; i++) p[i] = (unsigned char)(i * 7 + 1);
        return true;
    }
};

struct CjfifDecodeStub {
    CimgDecodeGpu* m_pImgDec; CDecodePsStub* m_pPsDec;
    unsigned short m_anImgDqtTbl[4][64]; unsigned m_anSofQuantTblSel_Tqi[256], m_anSofHorzSampFact_Hi[256], m_anSofVertSampFact_Vi[256];
    unsigned m_nSofPrecision_P = 8, m_nSofSampsPerLine_X = 0, m_nSofNumLines_Y = 0, m_nSofNumComps_Nf = 3, m_nSosNumCompScan_Ns = 3, m_nImgRstInterval = 0;
    bool m_nImgRstEn = false;
    void Site115() { m_pImgDec->Reset(); }
    bool Site3581(unsigned nDhtHuffTblId_Th, unsigned nDhtClass_Tc, unsigned nDhtLookupInd, unsigned nBitLen, unsigned nTmpBits, unsigned nTmpMask, unsigned nTmpCode)
    { bool bRet = m_pImgDec->SetDhtEntry(nDhtHuffTblId_Th, nDhtClass_Tc, nDhtLookupInd, nBitLen, nTmpBits, nTmpMask, nTmpCode); return bRet; }
    bool Site3600(unsigned nDhtHuffTblId_Th, unsigned nDhtClass_Tc, unsigned nTmpSize) { bool bRet = m_pImgDec->SetDhtSize(nDhtHuffTblId_Th, nDhtClass_Tc, nTmpSize); return bRet; }
    bool Site4648(unsigned nDqtQuantDestId_Tq, unsigned nCoeffInd, const unsigned* glb_anUnZigZag)
    { bool bRet = m_pImgDec->SetDqtEntry(nDqtQuantDestId_Tq, nCoeffInd, glb_anUnZigZag[nCoeffInd], m_anImgDqtTbl[nDqtQuantDestId_Tq][nCoeffInd]); return bRet; }
    bool Site5008(unsigned nCompInd, unsigned nCompIdent) { bool bRet = m_pImgDec->SetDqtTables(nCompInd, m_anSofQuantTblSel_Tqi[nCompIdent]); return bRet; }
    void Site5012() { m_pImgDec->SetPrecision(m_nSofPrecision_P); }
    void Site5025(unsigned nCompInd, unsigned nCompIdent) { m_pImgDec->SetSofSampFactors(nCompInd, m_anSofHorzSampFact_Hi[nCompIdent], m_anSofVertSampFact_Vi[nCompIdent]); }
    bool Site5161(unsigned nScanCompInd, unsigned nSosHuffTblSelDc_Td, unsigned nSosHuffTblSelAc_Ta) { bool bRet = m_pImgDec->SetDhtTables(nScanCompInd, nSosHuffTblSelDc_Td, nSosHuffTblSelAc_Ta); return bRet; }
    void Site5291() { m_pImgDec->SetImageDetails(m_nSofSampsPerLine_X, m_nSofNumLines_Y, m_nSofNumComps_Nf, m_nSosNumCompScan_Ns, m_nImgRstEn, m_nImgRstInterval); }
    void Site5299(unsigned long nPosScanStart) { m_pImgDec->DecodeScanImg(nPosScanStart, true, false); }
    bool Site6859(unsigned nDqtQuantDestId_Tq, unsigned nX, unsigned nY, const unsigned* glb_anUnZigZag)
    { bool bRet = m_pImgDec->SetDqtEntry(nDqtQuantDestId_Tq, nY * 8 + nX, glb_anUnZigZag[nY * 8 + nX], m_anImgDqtTbl[nDqtQuantDestId_Tq][nY * 8 + nX]); return bRet; }
    void Site7307() { m_pImgDec->ResetState(); }
    bool Site7369(unsigned long nStartPos)                            // :7369-7379, the Photoshop preview
    {
        unsigned nWidth = 0, nHeight = 0;
        bool bDecPsdOk = m_pPsDec->DecodePsd(nStartPos, &m_pImgDec->m_pDibTemp, nWidth, nHeight);
        if (bDecPsdOk) {
            m_pImgDec->m_bDibTempReady = true;
            m_pImgDec->m_bPreviewIsJpeg = false;
            m_pImgDec->SetImageDimensions(nWidth, nHeight);
            m_pImgDec->SetImageDetails(0, 0, 0, 0, false, 0);
        }
        return bDecPsdOk;
    }
};

int main(int argc, char** argv)
{
    if (argc < 2) { printf("built\n"); return 0; }                  // (the CPU suite only builds and links it)
    try {
        CimgDecodeGpu dec; CDecodePsStub ps; CjfifDecodeStub j; j.m_pImgDec = &dec; j.m_pPsDec = &ps;
        j.Site115(); j.Site7307();
        // a PSD-style preview: the members and SetImageDimensions as :7369-7379 leave them
        if (!j.Site7369(0)) { printf("psd_failed\n"); return 1; }
        unsigned char* bits = nullptr; dec.GetBitmapPtr(bits);
        unsigned w = 0, h = 0; dec.GetImageDimensions(w, h);
        printf("psd ready=%d bits=%d first=%u dims=%ux%u\n", (int)dec.IsPreviewReady(), bits != nullptr, bits ? bits[0] : 0u, w, h);
        // then a JPEG through the very same object: header walk = the setter call sites, DecodeScanImg = :5299
        FILE* f = fopen(argv[1], "rb"); if (!f) return 2;
        std::vector<uint8_t> buf; uint8_t tmp[65536]; size_t n; while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n); fclose(f);
        CwindowBufView view; view.pData = buf.data(); view.nLen = buf.size(); dec.SetWindowBuf(&view);
        j.Site115(); j.Site7307();
        unsigned nStart = 0; if (!dec.WalkJfifHeader(nStart)) { printf("walk_failed\n"); return 1; }
        j.Site5299(nStart);
        dec.GetBitmapPtr(bits); dec.GetImageDimensions(w, h);
        printf("jpeg ready=%d temp_ready=%d is_jpeg=%d bits=%d dims=%ux%u\n", (int)dec.IsPreviewReady(), (int)dec.m_bDibTempReady, (int)dec.m_bPreviewIsJpeg, bits != nullptr, w, h);
    } catch (const std::exception& e) { printf("%s\n", e.what()); return 3; }
    return 0;
}
