// gsr_common.h -- shared host/device helpers of libgsraster (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gsraster.h"

#define GSR_EXPORT extern "C" __attribute__((visibility("default")))

// ---- error reporting -------------------------------------------------------
void gsr_set_error(const char *fmt, ...);

#define GSR_REQUIRE(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      gsr_set_error(__VA_ARGS__);     \
      return GSR_EINVAL;              \
    }                                 \
  } while (0)

// checks the launch itself (asynchronous execution errors surface later)
#define GSR_CHECK_LAUNCH(what)                                               \
  do {                                                                       \
    hipError_t e_ = hipGetLastError();                                       \
    if (e_ != hipSuccess) {                                                  \
      gsr_set_error("%s: %s", what, hipGetErrorString(e_));                  \
      return GSR_ELAUNCH;                                                    \
    }                                                                        \
  } while (0)

#define GSR_CHECK_HIP(expr)                                                  \
  do {                                                                       \
    hipError_t e_ = (expr);                                                  \
    if (e_ != hipSuccess) {                                                  \
      gsr_set_error("%s: %s", #expr, hipGetErrorString(e_));                 \
      return GSR_ELAUNCH;                                                    \
    }                                                                        \
  } while (0)

static inline unsigned gsr_cdiv(unsigned a, unsigned b) { return (a + b - 1) / b; }

// ---- device helpers --------------------------------------------------------
#define GSR_WAVE 64

// thresholds of the compositing rule (forward.cu:360-366, backward.cu:232-233)
#define GSR_ALPHA_MIN (1.f / 255.f)
#define GSR_ALPHA_MAX_FWD 0.999f
#define GSR_ALPHA_MAX_BWD 0.99f
#define GSR_T_EPS 1e-4f

__device__ __forceinline__ int gsr_clampi(int v, int lo, int hi) {
  return min(max(v, lo), hi);
}

// tile bbox of a splat: inclusive min / exclusive max, clamped to the grid
// (helpers.cuh:11-34).  float->int conversion truncates and saturates.
__device__ __forceinline__ void gsr_tile_bbox(float cx, float cy, float radius,
                                              int tiles_x, int tiles_y,
                                              float inv_unused, int bw,
                                              int &minx, int &miny, int &maxx,
                                              int &maxy) {
  (void)inv_unused;
  const float fbw = (float)bw;
  const float tcx = cx / fbw, tcy = cy / fbw, tr = radius / fbw;
  minx = gsr_clampi((int)(tcx - tr), 0, tiles_x);
  maxx = gsr_clampi((int)(tcx + tr + 1.f), 0, tiles_x);
  miny = gsr_clampi((int)(tcy - tr), 0, tiles_y);
  maxy = gsr_clampi((int)(tcy + tr + 1.f), 0, tiles_y);
}

// conic (inverse cov2d) + 3-sigma radius (helpers.cuh:36-59)
__device__ __forceinline__ bool gsr_cov2d_bounds(float c0, float c1, float c2,
                                                 float &k0, float &k1,
                                                 float &k2, float &radius) {
  const float det = c0 * c2 - c1 * c1;
  if (det == 0.f) return false;
  const float inv_det = 1.f / det;
  k0 = c2 * inv_det;
  k1 = -c1 * inv_det;
  k2 = c0 * inv_det;
  const float b = 0.5f * (c0 + c2);
  const float disc = sqrtf(fmaxf(0.1f, b * b - det));
  radius = ceilf(3.f * sqrtf(fmaxf(b + disc, b - disc)));
  return true;
}

// XCD-aware workgroup -> work-item remap.  Workgroup b is dispatched to XCD
// b % 8 (observed, used for speed only): give every XCD one contiguous slab of
// the work range so neighbouring tiles (which share Gaussians) hit the same L2.
__device__ __forceinline__ unsigned gsr_xcd_remap(unsigned b, unsigned n) {
  const unsigned per = n / 8u, rem = n % 8u;
  const unsigned main = per * 8u;
  if (b >= main) return b;  // ragged tail keeps identity order
  (void)rem;
  return (b % 8u) * per + (b / 8u);
}
