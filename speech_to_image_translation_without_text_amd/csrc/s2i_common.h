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
