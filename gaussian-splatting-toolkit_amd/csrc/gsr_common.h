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

// Zero `bytes` bytes (a multiple of 4) at a 4-byte aligned device address with a KERNEL
// rather than hipMemsetAsync: a kernel node is captured into a HIP graph like every other
// launch of this library (gs_fused.ViewGraph replays whole views), whereas memset nodes
// recorded during stream capture were observed to leave the gradient accumulators
// un-zeroed on replay (ROCm 7.2: garbage gradients depending on the pool layout).
int gsr_zero_async(void *ptr, size_t bytes, hipStream_t s);

// ---- device helpers --------------------------------------------------------
// Streams that are touched once per step (SH coefficient rows, their gradients, the optimizer's state): non-temporal
// accesses (`global_load ... nt`) -- measured, 1 M Gaussians: sh16_fwd 57.8 -> 40.5 us for the same 216 MB.
typedef float gsr_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 gsr_load_stream(const float4 *p) {
  const gsr_v4f v = __builtin_nontemporal_load(reinterpret_cast<const gsr_v4f *>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void gsr_store_stream(float4 *p, const float4 v) {
  const gsr_v4f t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<gsr_v4f *>(p));
}
__device__ __forceinline__ float gsr_load_stream(const float *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void gsr_store_stream(float *p, const float v) { __builtin_nontemporal_store(v, p); }

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

// deep_tile_threshold carries one flag bit besides the threshold (include/gsraster.h, GSR_DEEP_ORDERED): the buffer
// behind tile_bins holds the launch's JOB ORDER (raster_common.h), built by the entry point itself.
// deep_tile_threshold's bits (include/gsraster.h): 0-21 the threshold; 22-27 GSR_DEEP_TAIL_64THS; 28 GSR_DEEP_SECOND
// (the second of the two job arrays behind tile_bins: the backward's); 29 GSR_DEEP_PREBUILT (gsr_tile_jobs_build has
// written the array: do not build); 30 GSR_DEEP_ORDERED.
#define GSR_DEEP_THRESHOLD_MASK 0x3FFFFF
__host__ __device__ __forceinline__ int gsr_deep_threshold(int v) { return v > 0 ? (v & GSR_DEEP_THRESHOLD_MASK) : 0; }
__host__ __device__ __forceinline__ bool gsr_deep_ordered(int v) { return v > 0 && (v & GSR_DEEP_ORDERED) != 0; }
__host__ __device__ __forceinline__ int gsr_deep_tail64(int v) { return gsr_deep_ordered(v) ? ((v >> 22) & 63) : 0; }
__host__ __device__ __forceinline__ int gsr_deep_second(int v) { return gsr_deep_ordered(v) && (v & GSR_DEEP_SECOND) ? 1 : 0; }

// XCD-aware workgroup -> tile remap.  Workgroup b is dispatched to XCD b % 8
// (observed, used for speed only).  The tile grid is cut into blocks of 8 x 4 tiles
// that are dealt to the XCDs round-robin: the tiles of a block -- which share most
// of their Gaussians -- hit the same L2, and every XCD sees all parts of the frame,
// so a scene concentrated in part of it still loads all eight evenly (one contiguous
// band of tile rows per XCD left two of them nearly idle on the trainer's scene:
// -7 % iterations/s), while no XCD gets more than one block above the average
// (whole tile rows dealt round-robin: 9 vs 8 rows at 1080p, +4 % on a uniform scene).
// Launch gsr_xcd_grid(tiles_x, tiles_y) workgroups; a result < 0 means "no tile".
#define GSR_XCD_BW 8
#define GSR_XCD_BH 4
__host__ __device__ __forceinline__ unsigned gsr_xcd_grid(int tiles_x, int tiles_y) {
  const unsigned blocks = (unsigned)((tiles_x + GSR_XCD_BW - 1) / GSR_XCD_BW) *
                          (unsigned)((tiles_y + GSR_XCD_BH - 1) / GSR_XCD_BH);
  return (blocks + 7u) / 8u * 8u * (GSR_XCD_BW * GSR_XCD_BH);
}
__device__ __forceinline__ int gsr_xcd_remap(unsigned b, int tiles_x, int tiles_y) {
  constexpr unsigned per = GSR_XCD_BW * GSR_XCD_BH;
  const unsigned xcd = b % 8u, slot = b / 8u;
  const unsigned k = (slot / per) * 8u + xcd, within = slot % per;
  const unsigned bx_count = (unsigned)(tiles_x + GSR_XCD_BW - 1) / GSR_XCD_BW;
  const int tx = (int)((k % bx_count) * GSR_XCD_BW + within % GSR_XCD_BW);
  const int ty = (int)((k / bx_count) * GSR_XCD_BH + within / GSR_XCD_BW);
  return (tx < tiles_x && ty < tiles_y) ? ty * tiles_x + tx : -1;
}
