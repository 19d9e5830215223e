// valubench.hip -- what the VALU of one gfx950 SIMD really issues per cycle, by instruction kind, by the number of
// independent chains per wave (ILP) and by the number of resident waves per SIMD (TLP).  The compositing kernels are
// VALU-shaped (no MFMA, ~0.1 % memory instructions): this is the ceiling they are held against.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o valubench valubench.hip && ./valubench
// Every workgroup is ONE wave (like the wave-per-tile kernels); `waves` single-wave workgroups per SIMD are launched
// (256 CUs x 4 SIMDs x waves) and the kernel's wall time is taken with events; s_memtime around the loop gives the
// cycles one wave saw.  Output: wave-instructions per cycle per SIMD (1 / "issue cycles"), lane-ops/s chip-wide.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

enum Op { FMA = 0, MUL_ADD, EXP2, RCP, CNDMASK, DPP_ADD, PERMLANE32, MIX_BWD, LDS_B128, MAXF, FMA_PK };
static const char *kNames[] = {"v_fma_f32", "v_mul+v_add", "v_exp_f32", "v_rcp_f32", "v_cndmask", "v_add dpp",
                               "permlane32_swap+add", "mix(exp,rcp,6fma,2cnd,min)", "ds_read_b128 bcast + fma",
                               "v_max_f32", "v_pk_fma_f32"};

template <int OP>
__device__ __forceinline__ float step(float a, const float m, const float c, const float4 *lds, int i) {
  if constexpr (OP == FMA) return __builtin_fmaf(a, m, c);
  if constexpr (OP == MUL_ADD) return (a * m) + c;
  if constexpr (OP == EXP2) return __builtin_amdgcn_exp2f(a);
  if constexpr (OP == RCP) return __builtin_amdgcn_rcpf(a);
  if constexpr (OP == CNDMASK) return a > c ? m : a;
  if constexpr (OP == MAXF) return fmaxf(a, c) ;
  if constexpr (OP == DPP_ADD)
    return a + __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(a), 0xB1, 0xf, 0xf, true));
  if constexpr (OP == PERMLANE32) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(c), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  if constexpr (OP == MIX_BWD) {
    // the shape of one (pixel, splat) of the compositing backward: exp, min, rcp, selects, fmas
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
  if constexpr (OP == LDS_B128) {
    const float4 v = lds[i & 63];  // wave-uniform address: broadcast read
    return __builtin_fmaf(a, v.x, v.y);
  }
  return a;
}

template <int OP, int ILP>
__global__ __launch_bounds__(64) void bench_kernel(const int iters, const float m, const float c, float *out,
                                                   unsigned long long *cycles) {
  __shared__ float4 lds[64];
  lds[threadIdx.x] = make_float4(m, c, m, c);
  __syncthreads();
  float a[ILP];
#pragma unroll
  for (int k = 0; k < ILP; ++k) a[k] = 1.0f + 0.001f * (threadIdx.x + k);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int k = 0; k < ILP; ++k) a[k] = step<OP>(a[k], m, c, lds, i + j);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < ILP; ++k) s += a[k];
  if (s == 123.456f) out[blockIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x < 4096) cycles[blockIdx.x] = t1 - t0;
}

typedef float v2f __attribute__((ext_vector_type(2)));
template <int ILP>
__global__ __launch_bounds__(64) void pk_kernel(const int iters, const float m, const float c, float *out,
                                                unsigned long long *cycles) {
  v2f a[ILP];
#pragma unroll
  for (int k = 0; k < ILP; ++k) a[k] = v2f{1.0f + 0.001f * (threadIdx.x + k), 2.f};
  const v2f mm = {m, m}, cc = {c, c};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int k = 0; k < ILP; ++k) a[k] = __builtin_elementwise_fma(a[k], mm, cc);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < ILP; ++k) s += a[k].x + a[k].y;
  if (s == 123.456f) out[blockIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x < 4096) cycles[blockIdx.x] = t1 - t0;
}

static float *d_out;
static unsigned long long *d_cyc;
static hipEvent_t e0, e1;

template <int OP, int ILP>
void run(int waves_per_simd, double insts_per_step) {
  const int iters = 8192, grid = 256 * 4 * waves_per_simd;
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    if constexpr (OP == FMA_PK) hipLaunchKernelGGL((pk_kernel<ILP>), dim3(grid), dim3(64), 0, 0, iters, 0.999f, 0.001f, d_out, d_cyc);
    else hipLaunchKernelGGL((bench_kernel<OP, ILP>), dim3(grid), dim3(64), 0, 0, iters, 0.999f, 0.001f, d_out, d_cyc);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) best = ms < best ? ms : best;
  }
  std::vector<unsigned long long> cyc(std::min(grid, 4096));
  CK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double mean = 0;
  for (auto v : cyc) mean += (double)v;
  mean /= cyc.size();
  const double wave_insts = (double)iters * ILP * insts_per_step;       // per wave
  const double total = wave_insts * grid;                               // wave-instructions chip-wide
  const double per_simd_per_s = total / 1024.0 / (best * 1e-3);
  printf("%-30s ilp %d waves/SIMD %d : %8.3f ms  wave-cycles/inst (s_memtime) %6.2f  "
         "inst/s/SIMD %.3f G  (= %.2f cyc/inst at 2.4 GHz)  lane-ops %.1f T/s\n",
         kNames[OP], ILP, waves_per_simd, best, mean / wave_insts, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s,
         total * 64.0 / (best * 1e-3) / 1e12);
}

template <int OP>
void sweep(double insts_per_step) {
  for (int w : {1, 2, 4, 8}) {
    run<OP, 1>(w, insts_per_step);
    run<OP, 2>(w, insts_per_step);
    run<OP, 4>(w, insts_per_step);
    run<OP, 8>(w, insts_per_step);
  }
}

int main() {
  CK(hipMalloc(&d_out, 1 << 20));
  CK(hipMalloc(&d_cyc, 4096 * sizeof(unsigned long long)));
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  sweep<FMA>(1);
  sweep<FMA_PK>(1);
  sweep<MUL_ADD>(2);
  sweep<EXP2>(1);
  sweep<RCP>(1);
  sweep<CNDMASK>(2);   // v_cmp + v_cndmask
  sweep<MAXF>(1);
  sweep<DPP_ADD>(1);
  sweep<PERMLANE32>(2);  // swap + add
  sweep<MIX_BWD>(16);    // ~16 VALU per step (counted from the ISA; see the printed cycles per step = 16 x cyc/inst)
  sweep<LDS_B128>(1);
  return 0;
}
