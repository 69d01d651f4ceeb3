// Shared helpers for the gfx950 kernels (internal; the public surface is include/s2i_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/s2i_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void s2i_set_error(const char* fmt, ...);

#define S2I_FAIL(...)            \
  do {                           \
    s2i_set_error(__VA_ARGS__);  \
    return 1;                    \
  } while (0)

#define S2I_REQUIRE(cond, ...)   \
  do {                           \
    if (!(cond)) S2I_FAIL(__VA_ARGS__); \
  } while (0)

#define S2I_LAUNCH_CHECK(name)                                          \
  do {                                                                  \
    hipError_t e__ = hipGetLastError();                                 \
    if (e__ != hipSuccess) S2I_FAIL("%s: launch failed: %s", name, hipGetErrorString(e__)); \
  } while (0)

static inline int s2i_ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
static inline bool s2i_is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static inline int s2i_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- tuning knobs (s2i_set_tuning / S2I_TUNE, parsed once at load; s2i_runtime.hip) ----
enum {
  S2I_TUNE_FWD_BM = 0,      // fwd_bm: force the tile height of the fp32 matrix kernel (0 = planner)
  S2I_TUNE_FWD_MIN_CPS,     // fwd_min_cps: fewest 32-deep K chunks a split-K block keeps
  S2I_TUNE_B16_V2,          // b16_v2: 0 never / 1 where eligible and >= 224 tiles / 2 wherever eligible
  S2I_TUNE_B16_PERSIST,     // b16_persist: 0 off / 1 one block per CU / n block slots (tests)
  S2I_TUNE_FINALIZE_THREADS, // finalize_threads: thread cap of a BatchNorm finalize block (default 256)
  S2I_TUNE_B16_DBG,         // b16_dbg: diagnostic instantiation of the bf16 convolution kernels (libs2i_hip_diag.so only)
  S2I_TUNE_WGRAD16_BM,      // wgrad16_bm: tile height of the bf16 weight-gradient kernel where 256 divides K (0 = cost model, 128, 256, 512 = 256 x 256)
  S2I_TUNE_WGRAD_BM,        // wgrad_bm: tile of the fp32 weight-gradient kernel where 256 divides K (0 = cost model, 1 = model without 256 x 256, 128, 256, 512 = 256 x 256)
  S2I_TUNE_COUNT
};
int s2i_tune(int key, int def);
