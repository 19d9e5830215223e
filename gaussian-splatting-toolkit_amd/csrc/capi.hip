// capi.hip -- library-level entry points of libgsraster (version, errors).
#include <stdarg.h>

#include "gsr_common.h"

namespace {
thread_local char g_err[512] = {0};
}

void gsr_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

GSR_EXPORT int gsr_version(void) { return GSR_VERSION; }

GSR_EXPORT const char *gsr_last_error(void) { return g_err; }
