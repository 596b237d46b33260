// Library-level entry points: ABI version + thread-local error string.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/daft_exprt_hip.h"

static thread_local char g_err[512] = "";

void dx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int dx_abi_version(void) { return DX_ABI_VERSION; }
extern "C" const char* dx_last_error(void) { return g_err; }
