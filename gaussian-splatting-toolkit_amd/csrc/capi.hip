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
