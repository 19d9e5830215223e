// bwbench.hip -- what can a streaming kernel reach on this MI355X?  Pure read (dwordx4 loads summed into one
// value per workgroup), pure write, copy; at the sizes of the path's streaming kernels (100 / 216 MB) and at 1 GB
// (beyond the 256 MB Infinity Cache).  The ceilings the streaming kernels of DESIGN.md section 4 are held against.
// hipcc --offload-arch=gfx950 -O3 -o bwbench bwbench.hip && ./bwbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 ld(const float4 *p) {
  if (NT) {
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
  }
  return *p;
}
template <bool NT>
__device__ __forceinline__ void st(float4 *p, float4 v) {
  if (NT) {
    const v4f t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(p));
  } else {
    *p = v;
  }
}
template <bool NT>
__global__ __launch_bounds__(256) void read_kernel(const float4 *__restrict__ p, size_t n4, float *__restrict__ out) {
  float4 acc = make_float4(0, 0, 0, 0);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = ld<NT>(p + i);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  const float s = acc.x + acc.y + acc.z + acc.w;
  if (s == 123.456f) out[blockIdx.x] = s;  // never true: keeps the loads alive without a store stream
}
template <bool NT>
__global__ __launch_bounds__(256) void write_kernel(float4 *__restrict__ p, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) st<NT>(p + i, make_float4(1.f, 2.f, 3.f, 4.f));
}
template <bool NT>
__global__ __launch_bounds__(256) void copy_kernel(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) st<NT>(b + i, ld<NT>(a + i));
}

int main() {
  const size_t maxb = (size_t)1 << 30;
  float4 *a, *b; float *out;
  CK(hipMalloc(&a, maxb)); CK(hipMalloc(&b, maxb)); CK(hipMalloc(&out, 1 << 20));
  CK(hipMemset(a, 0, maxb)); CK(hipMemset(b, 0, maxb));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t sizes[] = {(size_t)100e6, (size_t)216e6, maxb};
  const int grids[] = {2048, 8192, 32768};
  for (size_t bytes : sizes) {
    const size_t n4 = bytes / 16;
    for (int nt = 0; nt < 2; ++nt)
    for (int mode = 0; mode < 3; ++mode) {
      float best = 1e9f; int bestg = 0;
      for (int g : grids) {
        float tot = 0;
        for (int r = 0; r < 12; ++r) {
          CK(hipEventRecord(e0));
          if (mode == 0 && !nt) hipLaunchKernelGGL(read_kernel<false>, dim3(g), dim3(256), 0, 0, a, n4, out);
          if (mode == 1 && !nt) hipLaunchKernelGGL(write_kernel<false>, dim3(g), dim3(256), 0, 0, b, n4);
          if (mode == 2 && !nt) hipLaunchKernelGGL(copy_kernel<false>, dim3(g), dim3(256), 0, 0, a, b, n4 / 2);
          if (mode == 0 && nt) hipLaunchKernelGGL(read_kernel<true>, dim3(g), dim3(256), 0, 0, a, n4, out);
          if (mode == 1 && nt) hipLaunchKernelGGL(write_kernel<true>, dim3(g), dim3(256), 0, 0, b, n4);
          if (mode == 2 && nt) hipLaunchKernelGGL(copy_kernel<true>, dim3(g), dim3(256), 0, 0, a, b, n4 / 2);
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          if (r >= 2) tot += ms;
        }
        if (tot / 10 < best) { best = tot / 10; bestg = g; }
      }
      const double moved = mode == 2 ? (double)(n4 / 2) * 32 : (double)n4 * 16;
      printf("%7.0f MB  %-5s %-3s %7.1f us  %6.2f TB/s  (grid %d)\n", bytes / 1e6, mode == 0 ? "read" : mode == 1 ? "write" : "copy", nt ? "nt" : "",
             best * 1e3, moved / (best * 1e-3) / 1e12, bestg);
    }
  }
  return 0;
}
