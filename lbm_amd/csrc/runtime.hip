// Error reporting shared by every entry point of the library.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void lbmk_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int lbmk_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    lbmk_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return LBMK_ERR_LAUNCH;
  }
  return LBMK_OK;
}

extern "C" const char* lbmk_last_error(void) { return g_err; }
extern "C" int lbmk_version(void) { return 1; }
