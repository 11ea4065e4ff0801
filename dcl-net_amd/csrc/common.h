// common.h -- shared helpers of libdclnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/dclnet_hip.h"

#define DCL_API extern "C" __attribute__((visibility("default")))

void dcl_set_error(const char *fmt, ...);

#define DCL_CHECK_ARG(cond)                                                          \
  do {                                                                               \
    if (!(cond)) {                                                                   \
      dcl_set_error("%s: invalid argument: %s", __func__, #cond);                    \
      return DCL_EINVAL;                                                             \
    }                                                                                \
  } while (0)

#define DCL_LAUNCH_CHECK()                                                           \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess) {                                                         \
      dcl_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__));      \
      return (int)e__;                                                               \
    }                                                                                \
  } while (0)

static inline int dcl_div_up(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline int dcl_grid_1d(long long work, int block, int max_blocks = 256 * 16) {
  long long g = (work + block - 1) / block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return (int)g;
}

// Squared distance under the pinned contraction policy (DESIGN.md "Floating point"):
// dx*dx + dy*dy + dz*dz evaluated as fma(dz,dz, fma(dx,dx, dy*dy)).
__device__ __forceinline__ float dcl_dist2(float ax, float ay, float az, float bx, float by, float bz) {
  float dx = ax - bx, dy = ay - by, dz = az - bz;
  return __fmaf_rn(dz, dz, __fmaf_rn(dx, dx, dy * dy));
}
// a*b + c*d + e*f under the same policy.
__device__ __forceinline__ float dcl_wsum3(float a, float b, float c, float d, float e, float f) {
  return __fmaf_rn(e, f, __fmaf_rn(a, b, c * d));
}
