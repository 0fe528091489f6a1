// Error plumbing and device queries shared by every C-ABI entry point.
#include "common.h"
#include <string.h>

namespace {
thread_local char g_err[512] = "";
}

void nvl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int nvl_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    nvl_set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return NVL_ELAUNCH;
  }
  return NVL_OK;
}

extern "C" int nvl_abi_version(void) { return 2; }   // 2: kv_dtype arguments, collectives, lm_head sampler

extern "C" const char* nvl_last_error(void) { return g_err; }

extern "C" int nvl_device_cu_count(void) {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
  return n;
}
