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

int lbmk_num_cus(void) {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n;
  }
  return cached[dev];
}

extern "C" const char* lbmk_last_error(void) { return g_err; }
extern "C" int lbmk_version(void) { return 1; }
