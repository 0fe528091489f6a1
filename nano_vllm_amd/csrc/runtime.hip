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

extern "C" int nvl_abi_version(void) { return 6; }   // 6: nvl_allreduce_status_async (a latched collective timeout travels with the step's ids), decode attention for every group size 1 ... 16; 5: nvl_decode_plan takes the shared-prefix block count (shared-prefix attention pass); 4: fused lm_head sampler retired; qkv split-K slabs into the fused decode attention (3: per-step decode plan, LSE outputs)

extern "C" const char* nvl_last_error(void) { return g_err; }

int nvl_device_slot(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 0;
  return dev < NVL_MAX_DEVICES ? dev : NVL_MAX_DEVICES - 1;
}

extern "C" int nvl_device_cu_count(void) {
  static int cache[NVL_MAX_DEVICES] = {};
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  const int slot = dev >= 0 && dev < NVL_MAX_DEVICES ? dev : -1;
  if (slot >= 0 && cache[slot] > 0) return cache[slot];
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
  if (slot >= 0) cache[slot] = n;
  return n;
}
