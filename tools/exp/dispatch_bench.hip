// How fast does gfx950 START workgroups?  The compositing launches on small tile grids are tens of thousands of one-wave
// workgroups that live a few microseconds each (tools/exp/wave_trace.py: 17 005 waves of 17.8 us in a 133-us launch of 49 152
// workgroups, 2.5 resident waves per SIMD of 8): is that launch paced by the dispatcher?
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/dispatch_bench tools/exp/dispatch_bench.hip && tools/exp/dispatch_bench
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int THREADS, int LDS_BYTES>
__global__ __launch_bounds__(THREADS) void wg_kernel(int *out, const int spin_ticks, const int live_every) {
  __shared__ int lds[LDS_BYTES / 4 > 0 ? LDS_BYTES / 4 : 1];
  if (live_every > 1 && (blockIdx.x % live_every) != 0) return;  // (the segment grids are two thirds empty)
  if (LDS_BYTES) lds[threadIdx.x] = threadIdx.x;
  if (spin_ticks > 0) {
    const unsigned long long t0 = wall_clock64();  // 100 MHz
    while (wall_clock64() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
  }
  if (out && LDS_BYTES && lds[(threadIdx.x + 1) % THREADS] == -1) out[0] = 1;
}

// the same with ~64 live VGPRs per lane (the compositing kernels' register footprint): does a wave with more state start slower?
__global__ __launch_bounds__(64) void wg_kernel_regs(float *out, const int spin_ticks, const int live_every, const float seed) {
  __shared__ int lds[768];
  if (live_every > 1 && (blockIdx.x % live_every) != 0) return;
  lds[threadIdx.x] = threadIdx.x;
  float acc[56];
#pragma unroll
  for (int i = 0; i < 56; ++i) acc[i] = seed * (float)(i + 1) + (float)threadIdx.x;
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)spin_ticks) {
#pragma unroll
    for (int i = 0; i < 56; ++i) acc[i] = acc[i] * 1.0001f + acc[(i + 7) % 56];
    __builtin_amdgcn_s_sleep(4);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 56; ++i) s += acc[i];
  if (out && s == 12345.678f) out[0] = s + lds[(threadIdx.x + 1) & 63];
}

void run_regs(int blocks, int spin_ticks, int live_every) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(wg_kernel_regs, dim3(blocks), dim3(64), 0, 0, nullptr, spin_ticks, live_every, 0.5f);
  hipDeviceSynchronize();
  const int reps = 20;
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(wg_kernel_regs, dim3(blocks), dim3(64), 0, 0, nullptr, spin_ticks, live_every, 0.5f);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  printf("%-34s blocks %7d x  64 threads, lds  3072 B, live 1/%d, each %5.1f us: %8.1f us per launch = %7.1f workgroups/us\n",
         "one wave, ~64 VGPRs, alive", blocks, live_every, spin_ticks / 100.0, us, blocks / us);
}

template <int THREADS, int LDS_BYTES>
void run(const char *name, int blocks, int spin_ticks, int live_every) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((wg_kernel<THREADS, LDS_BYTES>), dim3(blocks), dim3(THREADS), 0, 0, nullptr, spin_ticks, live_every);
  hipDeviceSynchronize();
  const int reps = 20;
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((wg_kernel<THREADS, LDS_BYTES>), dim3(blocks), dim3(THREADS), 0, 0, nullptr, spin_ticks, live_every);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  printf("%-34s blocks %7d x %3d threads, lds %5d B, live 1/%d, each %5.1f us: %8.1f us per launch = %7.1f workgroups/us\n", name, blocks,
         THREADS, LDS_BYTES, live_every, spin_ticks / 100.0, us, blocks / us);
}

int main() {
  for (int blocks : {12288, 49152, 196608}) {
    run<64, 0>("empty", blocks, 0, 1);
    run<64, 3072>("lds 3 KB", blocks, 0, 1);
    run<256, 12288>("4 waves, lds 12 KB", blocks / 4, 0, 1);
  }
  for (int ticks : {200, 500, 1800}) {  // 2, 5, 18 us of life per workgroup
    run<64, 3072>("one wave, alive", 49152, ticks, 1);
    run<64, 3072>("one wave, alive, 2/3 empty", 49152, ticks, 3);
    run<256, 12288>("four waves, alive", 12288, ticks, 1);
    run<256, 12288>("four waves, alive, 2/3 empty", 12288, ticks, 3);
  }
  for (int ticks : {500, 1800}) {
    run_regs(49152, ticks, 1);
    run_regs(49152, ticks, 3);
  }
  return 0;
}
