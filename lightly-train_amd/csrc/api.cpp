// Error plumbing and device diagnostics of the C ABI (include/lt_amd.h).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/lt_amd.h"

static thread_local char g_err[512] = "";

void lt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* lt_last_error(void) { return g_err; }
extern "C" int lt_abi_version(void) { return 5; }

extern "C" int lt_device_info(char* name, int name_len, int* compute_units, int* clock_khz) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  hipDeviceProp_t p;
  if (e == hipSuccess) e = hipGetDeviceProperties(&p, dev);
  if (e != hipSuccess) {
    lt_set_error("lt_device_info: %s", hipGetErrorString(e));
    return LT_ERR_HIP;
  }
  if (name && name_len > 0) {
    snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
  }
  if (compute_units) *compute_units = p.multiProcessorCount;
  if (clock_khz) *clock_khz = p.clockRate;
  return LT_OK;
}
