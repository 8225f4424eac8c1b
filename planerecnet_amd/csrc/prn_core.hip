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
