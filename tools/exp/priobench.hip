// priobench.hip -- does s_setprio buy a wave of a VALU-bound kernel a larger share of its SIMD?  Eight single-wave
// workgroups per SIMD run the same loop (the instruction mix of one compositing pixel block); every `period`-th one
// raises its priority first.  Printed: mean cycles per loop step of the preferred waves and of the others, against the
// all-equal launch.   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o priobench priobench.hip && ./priobench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float step(float a, const float m, const float c) {
  const float vis = __builtin_amdgcn_exp2f(-a);
  const float alpha = fminf(0.99f, m * vis);
  const float ra = __builtin_amdgcn_rcpf(1.f - alpha);
  const float Tn = c * ra;
  const bool valid = alpha >= 0.004f;
  const float d = m * a + c;
  const float va = Tn * d + ra * a;
  const float w = valid ? vis * va : 0.f;
  const float fac = valid ? alpha * Tn : 0.f;
  return a + 1e-9f * (w + fac * d);
}

template <int ILP>
__global__ __launch_bounds__(64) void k(const int iters, const int long_iters, const int period, const int prio, const float m,
                                        const float c, float *out, unsigned long long *cycles) {
  const bool special = period > 0 && (blockIdx.x / 8) % period == 0;  // (blocks b, b + 8, ... share an XCD)
  if (special && prio == 1) __builtin_amdgcn_s_setprio(1);
  if (special && prio == 2) __builtin_amdgcn_s_setprio(2);
  if (special && prio == 3) __builtin_amdgcn_s_setprio(3);
  const int n = special ? long_iters : iters;
  float a[ILP];
#pragma unroll
  for (int q = 0; q < ILP; ++q) a[q] = 1.0f + 0.001f * (threadIdx.x + q);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int q = 0; q < ILP; ++q) a[q] = step(a[q], m, c);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < ILP; ++q) s += a[q];
  if (s == 123.456f) out[blockIdx.x] = s;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
  float *d_out;
  unsigned long long *d_cyc;
  const int grid = 256 * 4 * 8;  // eight resident waves per SIMD, one round
  CK(hipMalloc(&d_out, grid * 4));
  CK(hipMalloc(&d_cyc, grid * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<unsigned long long> cyc(grid);
  const int iters = 4096;
  for (int long_factor : {1, 4}) {
    for (int period : {0, 8, 2}) {
      for (int prio : {0, 1, 3}) {
        if (period == 0 && prio) continue;
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipEventRecord(e0));
          hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 0, 0, iters, iters * long_factor, period, prio, 0.999f, 0.001f, d_out, d_cyc);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          best = ms < best ? ms : best;
        }
        CK(hipMemcpy(cyc.data(), d_cyc, grid * 8, hipMemcpyDeviceToHost));
        double sp = 0, ot = 0;
        int nsp = 0, no = 0;
        for (int b = 0; b < grid; ++b) {
          const bool special = period > 0 && (b / 8) % period == 0;
          if (special) sp += (double)cyc[b] / (iters * long_factor), ++nsp;
          else ot += (double)cyc[b] / iters, ++no;
        }
        printf("long x%d  every %d-th wave special, prio %d : kernel %.3f ms; s_memtime ticks per step: special %.2f (n %d), others %.2f (n %d)\n",
               long_factor, period, prio, best, nsp ? sp / nsp : 0.0, nsp, no ? ot / no : 0.0, no);
      }
    }
  }
  return 0;
}
