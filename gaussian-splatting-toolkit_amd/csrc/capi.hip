// capi.hip -- library-level entry points of libgsraster (version, errors).
#include <stdarg.h>

#include "gsr_common.h"

namespace {
thread_local char g_err[512] = {0};
}

void gsr_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

GSR_EXPORT int gsr_version(void) { return GSR_VERSION; }

GSR_EXPORT const char *gsr_last_error(void) { return g_err; }

namespace {
__global__ __launch_bounds__(256) void zero_kernel(uint32_t *__restrict__ p, const size_t words) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += stride) p[i] = 0u;
}
}  // namespace

int gsr_zero_async(void *ptr, size_t bytes, hipStream_t s) {
  if (bytes == 0) return GSR_OK;
  if ((bytes & 3) || (reinterpret_cast<uintptr_t>(ptr) & 3)) {
    gsr_set_error("gsr_zero_async: pointer and size must be multiples of 4");
    return GSR_EINVAL;
  }
  const size_t words = bytes >> 2;
  const unsigned blocks = (unsigned)((words + 1023) / 1024 < 4096 ? (words + 1023) / 1024 : 4096);
  hipLaunchKernelGGL(zero_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, static_cast<uint32_t *>(ptr), words);
  GSR_CHECK_LAUNCH("zero");
  return GSR_OK;
}

// ---- box calibration (bench.py `calibration`): two fixed workloads timed next to the bench's own steps, so that a
// move of the headline between two leases can be attributed to the box (clocks, memory) or to the code.
//   gsr_calibrate_valu: every lane runs `iters` rounds of 8 independent fma chains (non-packed fp32, the instruction
//     mix the compositing kernels are bound by): 16 * iters * lanes flops = 8 * iters * lanes VALU lane-ops.
//   gsr_calibrate_copy: a float4 grid-stride copy of `bytes` (reads bytes, writes bytes).
namespace {
__global__ __launch_bounds__(256) void calib_valu_kernel(const int iters, const float seed, float *__restrict__ out) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
  float a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  const float m = 0.999f, c = 0.001f;
  for (int i = 0; i < iters; i += 8) {  // (iters is rounded up to a multiple of 8 by the entry point)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a0 = __builtin_fmaf(a0, m, c);
      a1 = __builtin_fmaf(a1, m, c);
      a2 = __builtin_fmaf(a2, m, c);
      a3 = __builtin_fmaf(a3, m, c);
      a4 = __builtin_fmaf(a4, m, c);
      a5 = __builtin_fmaf(a5, m, c);
      a6 = __builtin_fmaf(a6, m, c);
      a7 = __builtin_fmaf(a7, m, c);
    }
  }
  const float s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  if (s == 123.456f) out[blockIdx.x] = s;  // never true: keeps the chains alive
}
__global__ __launch_bounds__(256) void calib_copy_kernel(const float4 *__restrict__ a, float4 *__restrict__ b,
                                                         const size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
}  // namespace

GSR_EXPORT long long gsr_calibrate_valu(int iters, int workgroups, float *scratch, gsr_stream_t stream) {
  if (iters <= 0 || workgroups <= 0 || !scratch) {
    gsr_set_error("calibrate_valu: bad arguments");
    return -1;
  }
  iters = (iters + 7) & ~7;
  hipLaunchKernelGGL(calib_valu_kernel, dim3((unsigned)workgroups), dim3(256), 0, (hipStream_t)stream, iters, 1.0f,
                     scratch);
  if (hipGetLastError() != hipSuccess) {
    gsr_set_error("calibrate_valu: launch failed");
    return -1;
  }
  return 8LL * iters * 256LL * workgroups;  // VALU lane-operations (one fma = one)
}

GSR_EXPORT int gsr_calibrate_copy(const void *src, void *dst, size_t bytes, gsr_stream_t stream) {
  GSR_REQUIRE(src && dst && bytes >= 16 && (bytes & 15) == 0, "calibrate_copy: need 16-byte multiples");
  GSR_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "calibrate_copy: need 16-byte alignment");
  hipLaunchKernelGGL(calib_copy_kernel, dim3(8192), dim3(256), 0, (hipStream_t)stream, (const float4 *)src,
                     (float4 *)dst, bytes / 16);
  GSR_CHECK_LAUNCH("calibrate_copy");
  return GSR_OK;
}

// ---- job-order statistics (raster_common.h: JobStats[2][8]): 1 KB per device, owned by the library, allocated on first
// use -- never inside a stream capture (hipMalloc / hipMemset would invalidate it): a launch whose stream is
// capturing before the buffer exists gets nullptr (the order kernel keys by the default ratio, the waves report nothing)
// and the next eager launch allocates it.
#include <mutex>
#include <stdlib.h>
namespace gsr {
float *gsr_job_stats_buffer(hipStream_t s) {
  static std::mutex mu;
  static float *table[64] = {nullptr};
  static bool failed[64] = {false};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (table[dev] || failed[dev]) return table[dev];
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess) {
    (void)hipGetLastError();  // (e.g. the legacy stream while another one captures)
    return nullptr;
  }
  if (cap != hipStreamCaptureStatusNone) return nullptr;  // (not `failed`: the next eager launch allocates)
  void *p = nullptr;
  if (hipMalloc(&p, 1024) != hipSuccess || hipMemset(p, 0, 1024) != hipSuccess) {
    (void)hipGetLastError();
    failed[dev] = true;
    return nullptr;
  }
  table[dev] = static_cast<float *>(p);
  return table[dev];
}
float gsr_job_split_ratio() {
  static const float v = [] {
    const char *e = getenv("GSR_DEEP_SPLIT_KEY");  // a fixed key ratio for a split tile's jobs (A/B); unset: measured
    const float f = e ? (float)atof(e) : 0.f;
    return f > 0.f && f <= 1.f ? f : 0.f;
  }();
  return v;
}
}  // namespace gsr
