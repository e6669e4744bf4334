// jsnoop_parallel.cpp -- host side of the parallel (self-synchronising) entropy path.
#include <string.h>
#include "jsnoop_host.h"
#include "jsnoop_launch.h"

void js_build_parallel_luts(JsTableSet* ts, uint32_t ncomp) { (void)ncomp; ts->lut_ok = 0; }
int  js_parallel_entropy(JsnoopBatch* b, bool timed) { (void)b; (void)timed; return 0; }
int  js_parallel_fixup(JsnoopBatch* b) { (void)b; return 0; }
