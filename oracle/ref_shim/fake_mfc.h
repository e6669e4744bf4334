// fake_mfc.h -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// A minimal stand-in for the slice of MFC/Win32 that the reference's
// ImgDecode.cpp / WindowBuf.cpp / General.cpp touch, so that those files can
// be compiled UNMODIFIED, in place, from /root/reference/source with g++
// (see oracle/Makefile, target `ref`).  Nothing here restates reference code;
// it only supplies the platform types the reference expects (CString, CRect,
// CDC no-ops, a memory-backed CFile, BITMAPINFO ...).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdarg>
#include <cstdint>
#include <cassert>
#include <cmath>
#include <string>
#include <vector>

typedef unsigned char  BYTE;
typedef unsigned char  byte;
typedef BYTE*          PBYTE;
typedef unsigned short WORD;
typedef uint32_t       DWORD;
typedef int32_t        LONG;
typedef unsigned int   UINT;
typedef int            BOOL;
typedef char           TCHAR;
typedef const char*    LPCTSTR;
typedef char*          LPTSTR;
typedef uint32_t       COLORREF;
typedef char*          LPSTR;
typedef const wchar_t* LPCWSTR;
inline wchar_t* lstrcpyW(wchar_t* d, const wchar_t* s) { wchar_t* r = d; while ((*d++ = *s++)) {} return r; }
#define _tcstoul strtoul
typedef unsigned long long ULONGLONG;
#define _T(x) x
#ifndef TRUE
#define TRUE 1
#define FALSE 0
#endif
#define ASSERT(x) ((void)0)
#define RGB(r,g,b) ((COLORREF)(((BYTE)(r)|((WORD)((BYTE)(g))<<8))|(((DWORD)(BYTE)(b))<<16)))
#define PS_DOT 2
#define TRANSPARENT 1
#define DT_CALCRECT 0x400
#define DT_NOPREFIX 0x800
#define DT_WORDBREAK 0x10
#define DT_LEFT 0
#define DT_TOP 0
#define DT_CENTER 1
#define DT_SINGLELINE 0x20
#define DT_VCENTER 4
#define BI_RGB 0
#define _istprint(c) isprint(c)
#ifndef max
#define max(a,b) (((a)>(b))?(a):(b))
#define min(a,b) (((a)<(b))?(a):(b))
#endif

class CString {
public:
    std::string s;
    CString() {}
    CString(const char* p) : s(p ? p : "") {}
    CString(const std::string& p) : s(p) {}
    operator LPCTSTR() const { return s.c_str(); }
    void Format(const char* fmt, ...) {
        va_list ap; va_start(ap, fmt);
        char buf[4096]; vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
        s = buf;
    }
    void AppendFormat(const char* fmt, ...) {
        va_list ap; va_start(ap, fmt);
        char buf[4096]; vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
        s += buf;
    }
    void Append(const char* p) { s += p; }
    void Append(const CString& o) { s += o.s; }
    CString& operator+=(const CString& o) { s += o.s; return *this; }
    CString& operator+=(const char* p) { s += p; return *this; }
    CString& operator+=(char c) { s += c; return *this; }
    CString& operator=(const char* p) { s = p ? p : ""; return *this; }
    int GetLength() const { return (int)s.size(); }
    bool IsEmpty() const { return s.empty(); }
    void Empty() { s.clear(); }
    CString Mid(int a, int n) const { if (a >= (int)s.size()) return CString(); return CString(s.substr(a, n)); }
    CString Mid(int a) const { if (a >= (int)s.size()) return CString(); return CString(s.substr(a)); }
    CString Left(int n) const { return CString(s.substr(0, n)); }
    CString Right(int n) const { return n >= (int)s.size() ? *this : CString(s.substr(s.size() - n)); }
    int Insert(int i, const char* p) { if (i > (int)s.size()) i = (int)s.size(); s.insert(i, p); return (int)s.size(); }
    char GetAt(int i) const { return s[i]; }
    char operator[](int i) const { return s[i]; }
    int Find(const char* p) const { size_t r = s.find(p); return r == std::string::npos ? -1 : (int)r; }
    int Compare(const char* p) const { return s.compare(p); }
    CString SpanIncluding(const char* set) const { size_t n = strspn(s.c_str(), set); return CString(s.substr(0, n)); }
    CString& operator=(const wchar_t* w) { s.clear(); while (w && *w) s += (char)*w++; return *this; }
    void MakeLower() { for (auto& c : s) c = (char)tolower(c); }
    void MakeUpper() { for (auto& c : s) c = (char)toupper(c); }
};
inline CString operator+(const CString& a, const CString& b) { return CString(a.s + b.s); }
inline CString operator+(const CString& a, const char* b) { return CString(a.s + b); }
inline CString operator+(const char* a, const CString& b) { return CString(std::string(a) + b.s); }
inline bool operator==(const CString& a, const char* b) { return a.s == b; }
inline bool operator==(const CString& a, const CString& b) { return a.s == b.s; }
inline bool operator!=(const CString& a, const char* b) { return a.s != b; }

struct CSize { int cx, cy; CSize() : cx(0), cy(0) {} CSize(int x, int y) : cx(x), cy(y) {} };
struct CPoint { int x, y; CPoint() : x(0), y(0) {} CPoint(int a, int b) : x(a), y(b) {}
    void Offset(int a, int b) { x += a; y += b; } };
struct RECT { LONG left, top, right, bottom; };
struct CRect : public RECT {
    CRect() { left = top = right = bottom = 0; }
    CRect(int l, int t, int r, int b) { left = l; top = t; right = r; bottom = b; }
    CRect(CPoint p, CSize s) { left = p.x; top = p.y; right = p.x + s.cx; bottom = p.y + s.cy; }
    CRect(CPoint a, CPoint b) { left = a.x; top = a.y; right = b.x; bottom = b.y; }
    int Width() const { return right - left; }
    int Height() const { return bottom - top; }
    CSize Size() const { return CSize(Width(), Height()); }
    CPoint TopLeft() const { return CPoint(left, top); }
    CPoint BottomRight() const { return CPoint(right, bottom); }
    void OffsetRect(int x, int y) { left += x; right += x; top += y; bottom += y; }
    void OffsetRect(CPoint p) { OffsetRect(p.x, p.y); }
    void InflateRect(int x, int y) { left -= x; right += x; top -= y; bottom += y; }
    void InflateRect(int l, int t, int r, int b) { left -= l; top -= t; right += r; bottom += b; }
    void DeflateRect(int x, int y) { InflateRect(-x, -y); }
    void SetRect(int l, int t, int r, int b) { left = l; top = t; right = r; bottom = b; }
    void SetRectEmpty() { left = top = right = bottom = 0; }
    bool IsRectEmpty() const { return Width() <= 0 || Height() <= 0; }
    bool PtInRect(CPoint p) const { return p.x >= left && p.x < right && p.y >= top && p.y < bottom; }
    bool IntersectRect(const RECT* a, const RECT* b) {
        left = max(a->left, b->left); right = min(a->right, b->right);
        top = max(a->top, b->top); bottom = min(a->bottom, b->bottom);
        if (left >= right || top >= bottom) { SetRectEmpty(); return false; } return true; }
    operator RECT*() { return this; }
};

class CObject { public: virtual ~CObject() {} };
class CGdiObject : public CObject {};
class CBrush : public CGdiObject { public: CBrush() {} CBrush(COLORREF) {} };
class CPen : public CGdiObject { public: CPen() {} CPen(int, int, COLORREF) {} };
class CFont : public CGdiObject {};
class CBitmap : public CGdiObject { public:
    bool CreateCompatibleBitmap(void*, int, int) { return true; } void DeleteObject() {} };
class CDC : public CObject {
public:
    void FillRect(const RECT*, CBrush*) {}
    void FrameRect(const RECT*, CBrush*) {}
    CFont* SelectObject(CFont* f) { return f; }
    CPen* SelectObject(CPen* p) { return p; }
    CBrush* SelectObject(CBrush* p) { return p; }
    CBitmap* SelectObject(CBitmap* p) { return p; }
    int DrawText(const CString&, int, RECT*, unsigned) { return 0; }
    int DrawText(const char*, int, RECT*, unsigned) { return 0; }
    int GetBkMode() { return 0; }
    int SetBkMode(int) { return 0; }
    COLORREF SetBkColor(COLORREF c) { return c; }
    COLORREF SetTextColor(COLORREF c) { return c; }
    void MoveTo(int, int) {}
    void LineTo(int, int) {}
    bool CreateCompatibleDC(CDC*) { return true; }
    bool BitBlt(int, int, int, int, CDC*, int, int, DWORD) { return true; }
    void* GetSafeHdc() { return nullptr; }
    void* m_hDC = nullptr;
};
class CStatusBar : public CObject { public: void SetPaneText(int, const CString&) {} void SetPaneText(int, const char*) {} };
class CDocument : public CObject {};
class CStringArray { public: std::vector<CString> v; int Add(const CString& s) { v.push_back(s); return (int)v.size() - 1; }
    int GetCount() const { return (int)v.size(); } int GetSize() const { return (int)v.size(); } void RemoveAll() { v.clear(); }
    CString GetAt(int i) const { return v[i]; } };
class CUIntArray { public: std::vector<unsigned> v; int Add(unsigned s) { v.push_back(s); return (int)v.size() - 1; }
    int GetCount() const { return (int)v.size(); } void RemoveAll() { v.clear(); } unsigned GetAt(int i) const { return v[i]; } };

// Memory-backed CFile: the reference's CwindowBuf reads its 128 KB windows
// through Seek/Read/GetLength only.
class CFileException : public CObject {
public:
    BOOL GetErrorMessage(TCHAR* msg, unsigned n) { if (n) msg[0] = 0; return TRUE; }
    void Delete() { delete this; }
};
class CFile : public CObject {
public:
    enum { begin = 0, current = 1, end = 2, modeRead = 0, shareDenyNone = 0, typeBinary = 0, modeCreate = 0x1000, modeWrite = 0x0001 };
    const BYTE* m_p = nullptr; ULONGLONG m_n = 0, m_pos = 0;
    FILE* m_out = nullptr;                                   // write mode (FileTiff): a real file
    CFile() {}
    CFile(const BYTE* p, ULONGLONG n) : m_p(p), m_n(n) {}
    CFile(CString name, unsigned flags) { if (flags & modeWrite) { m_out = fopen(name.s.c_str(), "wb"); if (!m_out) throw new CFileException(); } }
    ~CFile() { if (m_out) fclose(m_out); }
    ULONGLONG GetLength() const { return m_n; }
    ULONGLONG Seek(long long off, unsigned from) {
        long long base = from == begin ? 0 : from == current ? (long long)m_pos : (long long)m_n;
        long long np = base + off; if (np < 0) np = 0; m_pos = (ULONGLONG)np; return m_pos; }
    ULONGLONG SeekToBegin() { m_pos = 0; return 0; }
    unsigned Read(void* dst, unsigned n) {
        if (m_pos >= m_n) return 0; ULONGLONG k = m_n - m_pos; if (k > n) k = n;
        memcpy(dst, m_p + m_pos, (size_t)k); m_pos += k; return (unsigned)k; }
    void Write(const void* src, unsigned n) { if (m_out) fwrite(src, 1, n, m_out); }
    ULONGLONG GetPosition() const { return m_pos; }
    void Close() { if (m_out) { fclose(m_out); m_out = nullptr; } }
};
class CStdioFile : public CFile { public: void WriteString(const char*) {} };

#pragma pack(push, 1)
struct RGBQUAD { BYTE rgbBlue, rgbGreen, rgbRed, rgbReserved; };
struct BITMAPINFOHEADER { DWORD biSize; LONG biWidth; LONG biHeight; WORD biPlanes; WORD biBitCount;
    DWORD biCompression; DWORD biSizeImage; LONG biXPelsPerMeter; LONG biYPelsPerMeter; DWORD biClrUsed; DWORD biClrImportant; };
struct BITMAPINFO { BITMAPINFOHEADER bmiHeader; RGBQUAD bmiColors[1]; };
#pragma pack(pop)
typedef BITMAPINFO* LPBITMAPINFO;
typedef RGBQUAD* LPRGBQUAD;

class CWinApp : public CObject {};
class CWinAppEx : public CWinApp {};
CWinApp* AfxGetApp();
int AfxMessageBox(const char* msg, unsigned type = 0);
inline void OutputDebugString(const char*) {}
