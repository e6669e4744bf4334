// TEST INFRASTRUCTURE: redirects the MFC umbrella include to the fake-MFC shim.
#pragma once
#include "fake_mfc.h"
