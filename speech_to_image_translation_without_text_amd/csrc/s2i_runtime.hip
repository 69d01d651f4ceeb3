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

// ---- tuning knobs ----------------------------------------------------------------------------------------------------
#include <stdlib.h>
static const char* const g_tune_names[S2I_TUNE_COUNT] = {"fwd_bm", "fwd_min_cps", "b16_v2", "b16_persist", "finalize_threads", "b16_dbg",
                                                            "wgrad16_bm", "wgrad_bm"};
static int g_tune_val[S2I_TUNE_COUNT];
static bool g_tune_set[S2I_TUNE_COUNT];

static int tune_index(const char* key, size_t len) {
  for (int i = 0; i < S2I_TUNE_COUNT; ++i)
    if (strlen(g_tune_names[i]) == len && strncmp(g_tune_names[i], key, len) == 0) return i;
  return -1;
}

// S2I_TUNE="key=value,key=value": the library's only environment variable, read once when it is loaded
namespace {
struct TuneInit {
  TuneInit() {
    const char* e = getenv("S2I_TUNE");
    while (e && *e) {
      const char* eq = strchr(e, '=');
      if (!eq) break;
      const int i = tune_index(e, (size_t)(eq - e));
      if (i >= 0) { g_tune_val[i] = atoi(eq + 1); g_tune_set[i] = true; }
      const char* c = strchr(eq, ',');
      e = c ? c + 1 : nullptr;
    }
  }
} g_tune_init;
}  // namespace

int s2i_tune(int key, int def) { return g_tune_set[key] ? g_tune_val[key] : def; }

extern "C" int s2i_set_tuning(const char* key, int value) {
  const int i = key ? tune_index(key, strlen(key)) : -1;
  if (i < 0) S2I_FAIL("set_tuning: unknown key '%s'", key ? key : "(null)");
  g_tune_val[i] = value;
  g_tune_set[i] = value >= 0;      // knobs are non-negative; a negative value restores the built-in default
  return 0;
}

extern "C" int s2i_get_tuning(const char* key, int* value) {
  const int i = key ? tune_index(key, strlen(key)) : -1;
  if (i < 0 || !value) S2I_FAIL("get_tuning: unknown key '%s'", key ? key : "(null)");
  *value = g_tune_set[i] ? g_tune_val[i] : -1;
  return 0;
}
