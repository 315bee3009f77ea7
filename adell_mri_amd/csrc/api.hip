// Error reporting + version of the C ABI.
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void adell_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* adell_last_error(void) { return g_err; }
extern "C" int adell_abi_version(void) { return ADELL_ABI_VERSION; }
