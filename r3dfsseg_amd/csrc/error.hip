// Thread-local error string behind the C ABI (no C++ exceptions cross it).
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void r3d_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* r3d_last_error_string(void) { return g_err; }

extern "C" int r3d_abi_version(void) { return 2; }
