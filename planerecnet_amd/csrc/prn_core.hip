// Library plumbing: version + thread-local error string.
#include <stdarg.h>
#include "prn_common.h"

static thread_local char g_err[512] = "";

void prn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int prn_version(void) { return 100; }
extern "C" const char* prn_last_error(void) { return g_err; }

// Measurement aid, never set by the package: bit mask of helper launches to leave out so that a tool can measure what a step would gain if their work came for
// free from a producer's epilogue (tools/ablation_bounds.py).  RESULTS ARE WRONG while a bit is set (stale sums are consumed).
int prn_skip_mask = 0;
extern "C" int prn_debug_skip_launches(int mask) {
  const int old = prn_skip_mask;
  prn_skip_mask = mask;
  return old;
}
