// TEST INFRASTRUCTURE: stands in for the reference's application header so that
// ImgDecode.cpp's `((CJPEGsnoopApp*)AfxGetApp())->m_pAppConfig` resolves.
#pragma once
#include "fake_mfc.h"
#include "snoop.h"
#include "SnoopConfig.h"
class CJPEGsnoopApp : public CWinApp { public: CSnoopConfig* m_pAppConfig = nullptr; };
