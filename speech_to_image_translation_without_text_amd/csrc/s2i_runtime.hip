// Error reporting, version and device check of the C-ABI (include/s2i_hip.h).
#include "s2i_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void s2i_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* s2i_last_error(void) { return g_err; }

extern "C" int s2i_version(void) { return S2I_ABI_VERSION; }

extern "C" int s2i_check_device(void) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) S2I_FAIL("check_device: hipGetDevice: %s", hipGetErrorString(e));
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) S2I_FAIL("check_device: hipGetDeviceProperties: %s", hipGetErrorString(e));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    S2I_FAIL("check_device: kernels are built for gfx950 only, device %d is %s", dev, prop.gcnArchName);
  return 0;
}
