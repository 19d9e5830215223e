// valubench2.hip -- issue cost of single gfx950 VALU instructions by ENCODING and operand kind, pinned with inline asm
// (valubench.hip showed v_mul/v_add at ~2 cycles per wave64 instruction, the compiler's VOP3 v_fma_f32 at ~4, v_exp /
// v_rcp at ~8: which property makes an instruction a 2-cycle one decides how the compositing loops should be written).
//   hipcc --offload-arch=gfx950 -O3 -o valubench2 valubench2.hip && ./valubench2
// 8 single-wave workgroups per SIMD, 8 independent destination registers per instruction kind, 8192 x 8 x 8 issues.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// One asm block = 8 instructions on 8 independent accumulators a0..a7 (inputs x, y VGPR; s SGPR).
#define BODY8(INS)                                                                                          \
  asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                       \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
               : "v"(x), "v"(y), "s"(sm)                                                                      \
               : "vcc");

#define I_FMA_VOP3(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n"          // d = d*x + y   (VOP3, 3 VGPR sources)
#define I_FMAC_VOP2(k) "v_fmac_f32_e32 %" #k ", %8, %9\n"             // d += x*y      (VOP2)
#define I_FMAMK(k) "v_fmamk_f32 %" #k ", %" #k ", 0x3f7fbe77, %9\n"   // d = d*K + y   (VOP2 + literal)
#define I_FMAAK(k) "v_fmaak_f32 %" #k ", %" #k ", %8, 0x3a83126f\n"   // d = d*x + K
#define I_MUL_VOP2(k) "v_mul_f32_e32 %" #k ", %" #k ", %8\n"
#define I_MUL_VOP3(k) "v_mul_f32_e64 %" #k ", %" #k ", %8\n"
#define I_MUL_SGPR(k) "v_mul_f32_e32 %" #k ", %10, %" #k "\n"
#define I_ADD_VOP2(k) "v_add_f32_e32 %" #k ", %" #k ", %8\n"
#define I_SUB_VOP2(k) "v_sub_f32_e32 %" #k ", %" #k ", %8\n"
#define I_MIN_VOP2(k) "v_min_f32_e32 %" #k ", %" #k ", %8\n"
#define I_MAX_VOP2(k) "v_max_f32_e32 %" #k ", %" #k ", %8\n"
#define I_MOV(k) "v_mov_b32_e32 %" #k ", %8\n"
#define I_EXP(k) "v_exp_f32_e32 %" #k ", %" #k "\n"
#define I_RCP(k) "v_rcp_f32_e32 %" #k ", %" #k "\n"
#define I_RSQ(k) "v_rsq_f32_e32 %" #k ", %" #k "\n"
#define I_CMP(k) "v_cmp_lt_f32_e32 vcc, %" #k ", %8\n"
#define I_CMP_SGPR(k) "v_cmp_lt_f32_e64 s[20:21], %" #k ", %8\n"
#define I_CNDMASK(k) "v_cndmask_b32_e32 %" #k ", %" #k ", %8, vcc\n"
#define I_CNDMASK_E64(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, s[20:21]\n"
#define I_ADD_DPP(k) "v_add_f32_dpp %" #k ", %" #k ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_MOV_DPP(k) "v_mov_b32_dpp %" #k ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_ADD_DPP_ROR(k) "v_add_f32_dpp %" #k ", %" #k ", %8 row_ror:8 row_mask:0xf bank_mask:0xf\n"
#define I_PERMLANE32(k) "v_permlane32_swap_b32_e32 %" #k ", %8\n"
#define I_PERMLANE16(k) "v_permlane16_swap_b32_e32 %" #k ", %8\n"
#define I_PK_FMA(k) "v_pk_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define I_PK_MUL(k) "v_pk_mul_f32 %" #k ", %" #k ", %8\n"
#define I_PK_ADD(k) "v_pk_add_f32 %" #k ", %" #k ", %8\n"
#define I_AND(k) "v_and_b32_e32 %" #k ", %" #k ", %8\n"
#define I_ADD_U32(k) "v_add_u32_e32 %" #k ", %" #k ", %8\n"
#define I_LSHL_ADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 1, %8\n"
#define I_CVT(k) "v_cvt_f32_i32_e32 %" #k ", %" #k "\n"
#define I_FMA_NEG(k) "v_fma_f32 %" #k ", -%" #k ", %8, %9\n"
#define I_MUL_NEG(k) "v_mul_f32_e64 %" #k ", -%" #k ", %8\n"
#define I_READLANE(k) "v_readfirstlane_b32 s22, %" #k "\n"

template <int WHICH>
__global__ __launch_bounds__(64) void k1(const int iters, float *out, const float sm) {
  float a[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = 1.0f + 0.001f * (threadIdx.x + k);
  const float x = 0.999f + 1e-6f * threadIdx.x, y = 0.001f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if constexpr (WHICH == 0) BODY8(I_FMA_VOP3)
      if constexpr (WHICH == 1) BODY8(I_FMAC_VOP2)
      if constexpr (WHICH == 2) BODY8(I_FMAMK)
      if constexpr (WHICH == 3) BODY8(I_FMAAK)
      if constexpr (WHICH == 4) BODY8(I_MUL_VOP2)
      if constexpr (WHICH == 5) BODY8(I_MUL_VOP3)
      if constexpr (WHICH == 6) BODY8(I_MUL_SGPR)
      if constexpr (WHICH == 7) BODY8(I_ADD_VOP2)
      if constexpr (WHICH == 8) BODY8(I_SUB_VOP2)
      if constexpr (WHICH == 9) BODY8(I_MIN_VOP2)
      if constexpr (WHICH == 10) BODY8(I_MAX_VOP2)
      if constexpr (WHICH == 11) BODY8(I_MOV)
      if constexpr (WHICH == 12) BODY8(I_EXP)
      if constexpr (WHICH == 13) BODY8(I_RCP)
      if constexpr (WHICH == 14) BODY8(I_RSQ)
      if constexpr (WHICH == 15) BODY8(I_CMP)
      if constexpr (WHICH == 16) BODY8(I_CMP_SGPR)
      if constexpr (WHICH == 17) BODY8(I_CNDMASK)
      if constexpr (WHICH == 18) BODY8(I_CNDMASK_E64)
      if constexpr (WHICH == 19) BODY8(I_ADD_DPP)
      if constexpr (WHICH == 20) BODY8(I_MOV_DPP)
      if constexpr (WHICH == 21) BODY8(I_ADD_DPP_ROR)
      if constexpr (WHICH == 22) BODY8(I_PERMLANE32)
      if constexpr (WHICH == 23) BODY8(I_PERMLANE16)
      if constexpr (WHICH == 24) BODY8(I_AND)
      if constexpr (WHICH == 25) BODY8(I_ADD_U32)
      if constexpr (WHICH == 26) BODY8(I_LSHL_ADD)
      if constexpr (WHICH == 27) BODY8(I_CVT)
      if constexpr (WHICH == 28) BODY8(I_FMA_NEG)
      if constexpr (WHICH == 29) BODY8(I_MUL_NEG)
      if constexpr (WHICH == 30) BODY8(I_READLANE)
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += a[k];
  if (s == 123.456f) out[blockIdx.x] = s;
}

// packed: 64-bit register pairs
typedef float v2f __attribute__((ext_vector_type(2)));
#define BODY8P(INS)                                                                                         \
  asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                       \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
               : "v"(x), "v"(y));
template <int WHICH>
__global__ __launch_bounds__(64) void k2(const int iters, float *out) {
  v2f a[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = v2f{1.0f + 0.001f * (threadIdx.x + k), 1.5f};
  const v2f x = {0.999f, 0.998f}, y = {0.001f, 0.002f};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if constexpr (WHICH == 0) BODY8P(I_PK_FMA)
      if constexpr (WHICH == 1) BODY8P(I_PK_MUL)
      if constexpr (WHICH == 2) BODY8P(I_PK_ADD)
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += a[k].x + a[k].y;
  if (s == 123.456f) out[blockIdx.x] = s;
}

static float *d_out;
static hipEvent_t e0, e1;

template <typename F>
void timeit(const char *name, F launch, int waves) {
  const int iters = 4096;
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    launch(iters, 1024 * waves);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) best = ms < best ? ms : best;
  }
  const double per_simd = (double)iters * 64.0 * waves;  // wave-instructions per SIMD
  printf("%-44s waves/SIMD %d : %7.3f ms   %.2f cycles per wave-instruction per SIMD (2.4 GHz)\n", name, waves, best,
         best * 1e-3 * 2.4e9 / per_simd);
}

#define RUN1(W, NAME)                                                                                              \
  for (int w : {1, 4, 8}) timeit(NAME, [](int it, int grid) { hipLaunchKernelGGL((k1<W>), dim3(grid), dim3(64), 0, 0, it, d_out, 0.5f); }, w);
#define RUN2(W, NAME)                                                                                              \
  for (int w : {1, 4, 8}) timeit(NAME, [](int it, int grid) { hipLaunchKernelGGL((k2<W>), dim3(grid), dim3(64), 0, 0, it, d_out); }, w);

int main() {
  CK(hipMalloc(&d_out, 1 << 20));
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  RUN1(0, "v_fma_f32 (VOP3) d,d,v,v")
  RUN1(28, "v_fma_f32 (VOP3) d,-d,v,v")
  RUN1(1, "v_fmac_f32_e32 (VOP2) d+=v*v")
  RUN1(2, "v_fmamk_f32 d=d*K+v")
  RUN1(3, "v_fmaak_f32 d=d*v+K")
  RUN1(4, "v_mul_f32_e32")
  RUN1(5, "v_mul_f32_e64 (VOP3)")
  RUN1(29, "v_mul_f32_e64 -d (VOP3 neg)")
  RUN1(6, "v_mul_f32_e32 sgpr operand")
  RUN1(7, "v_add_f32_e32")
  RUN1(8, "v_sub_f32_e32")
  RUN1(9, "v_min_f32_e32")
  RUN1(10, "v_max_f32_e32")
  RUN1(11, "v_mov_b32_e32")
  RUN1(12, "v_exp_f32_e32")
  RUN1(13, "v_rcp_f32_e32")
  RUN1(14, "v_rsq_f32_e32")
  RUN1(15, "v_cmp_lt_f32_e32 -> vcc")
  RUN1(16, "v_cmp_lt_f32_e64 -> sgpr pair")
  RUN1(17, "v_cndmask_b32_e32 (vcc)")
  RUN1(18, "v_cndmask_b32_e64 (sgpr pair)")
  RUN1(19, "v_add_f32_dpp quad_perm")
  RUN1(20, "v_mov_b32_dpp quad_perm")
  RUN1(21, "v_add_f32_dpp row_ror:8")
  RUN1(22, "v_permlane32_swap_b32")
  RUN1(23, "v_permlane16_swap_b32")
  RUN1(24, "v_and_b32_e32")
  RUN1(25, "v_add_u32_e32")
  RUN1(26, "v_lshl_add_u32 (VOP3)")
  RUN1(27, "v_cvt_f32_i32_e32")
  RUN1(30, "v_readfirstlane_b32")
  RUN2(0, "v_pk_fma_f32 (2 lanes-ops per lane)")
  RUN2(1, "v_pk_mul_f32")
  RUN2(2, "v_pk_add_f32")
  return 0;
}
