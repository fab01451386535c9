// Error plumbing + version for the C-ABI (include/medplib_hip.h).  No global state besides the
// thread-local last-error string.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void mp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int mp_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    mp_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return MP_ERR_LAUNCH;
  }
  return MP_OK;
}

extern "C" const char* mp_last_error_string() { return g_err; }
extern "C" int mp_version() { return 100; }  // 0.1.0
extern "C" const char* mp_arch() { return "gfx950"; }
