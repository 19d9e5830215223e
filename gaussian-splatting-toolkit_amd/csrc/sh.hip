// sh.hip -- view-dependent colour from real spherical harmonics, and its VJP
// w.r.t. the coefficients (gfx950).
//
// Behaviour restated from rasterizer/cuda/csrc/sh.cuh:33-224: svox2 sign
// convention, view direction normalised in-kernel, 3 channels, coefficient
// layout [n, K, 3] (basis-major), bands above `degrees_to_use` ignored
// (forward) / zero (backward).
//
// HBM-bound streaming op; the coefficient tensor is the largest per-Gaussian
// read of the whole path (192 B at degree 3).  For K == 16 the work is mapped
// 16 lanes per Gaussian: lane (g,k) moves exactly one contiguous 12-B
// coefficient triple, so a wave-wide load/store covers 768 contiguous bytes;
// the 16-term dot product is a DPP row reduction (a row is 16 lanes on CDNA).
// Other degrees use one lane per Gaussian.
#include "gsr_common.h"

namespace {

#define C0 0.28209479177387814f
#define C1 0.4886025119029199f
#define C2_0 1.0925484305920792f
#define C2_1 -1.0925484305920792f
#define C2_2 0.31539156525252005f
#define C2_3 -1.0925484305920792f
#define C2_4 0.5462742152960396f
#define C3_0 -0.5900435899266435f
#define C3_1 2.890611442640554f
#define C3_2 -0.4570457994644658f
#define C3_3 0.3731763325901154f
#define C3_4 -0.4570457994644658f
#define C3_5 1.445305721320277f
#define C3_6 -0.5900435899266435f
#define C4_0 2.5033429417967046f
#define C4_1 -1.7701307697799304f
#define C4_2 0.9461746957575601f
#define C4_3 -0.6690465435572892f
#define C4_4 0.10578554691520431f
#define C4_5 -0.6690465435572892f
#define C4_6 0.47308734787878004f
#define C4_7 -1.7701307697799304f
#define C4_8 0.6258357354491761f

// basis vector up to `deg` for direction d (normalised here); B[k] = 0 above.
template <int KMAX>
__device__ __forceinline__ void sh_basis(unsigned deg, float dx, float dy, float dz,
                                         float (&B)[KMAX]) {
#pragma unroll
  for (int k = 0; k < KMAX; ++k) B[k] = 0.f;
  B[0] = C0;
  if (deg < 1 || KMAX < 4) return;
  const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
  const float x = dx / nrm, y = dy / nrm, z = dz / nrm;
  if constexpr (KMAX >= 4) {
    B[1] = -C1 * y;
    B[2] = C1 * z;
    B[3] = -C1 * x;
  }
  if (deg < 2 || KMAX < 9) return;
  const float xx = x * x, xy = x * y, xz = x * z, yy = y * y, yz = y * z, zz = z * z;
  if constexpr (KMAX >= 9) {
    B[4] = C2_0 * xy;
    B[5] = C2_1 * yz;
    B[6] = C2_2 * (2.f * zz - xx - yy);
    B[7] = C2_3 * xz;
    B[8] = C2_4 * (xx - yy);
  }
  if (deg < 3 || KMAX < 16) return;
  if constexpr (KMAX >= 16) {
    B[9] = C3_0 * y * (3.f * xx - yy);
    B[10] = C3_1 * xy * z;
    B[11] = C3_2 * y * (4.f * zz - xx - yy);
    B[12] = C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
    B[13] = C3_4 * x * (4.f * zz - xx - yy);
    B[14] = C3_5 * z * (xx - yy);
    B[15] = C3_6 * x * (xx - 3.f * yy);
  }
  if (deg < 4 || KMAX < 25) return;
  if constexpr (KMAX >= 25) {
    B[16] = C4_0 * xy * (xx - yy);
    B[17] = C4_1 * yz * (3.f * xx - yy);
    B[18] = C4_2 * xy * (7.f * zz - 1.f);
    B[19] = C4_3 * yz * (7.f * zz - 3.f);
    B[20] = C4_4 * (zz * (35.f * zz - 30.f) + 3.f);
    B[21] = C4_5 * xz * (7.f * zz - 3.f);
    B[22] = C4_6 * (xx - yy) * (7.f * zz - 1.f);
    B[23] = C4_7 * xz * (xx - 3.f * yy);
    B[24] = C4_8 * (xx * (xx - 3.f * yy) - yy * (3.f * xx - yy));
  }
}

// select B[k] for a lane-varying k without indexing registers dynamically
template <int KMAX>
__device__ __forceinline__ float pick(const float (&B)[KMAX], int k) {
  float r = B[0];
#pragma unroll
  for (int j = 1; j < KMAX; ++j) r = (k == j) ? B[j] : r;
  return r;
}

// sum over the 16 lanes of a DPP row; the total lands in lane 15 of the row
__device__ __forceinline__ float row_sum16(float v) {
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x111, 0xf, 0xf, true));  // row_shr:1
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x112, 0xf, 0xf, true));  // row_shr:2
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x114, 0xf, 0xf, true));  // row_shr:4
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x118, 0xf, 0xf, true));  // row_shr:8
  return v;
}

// ---- K == 16: 16 lanes per Gaussian ---------------------------------------
__global__ __launch_bounds__(256) void sh16_fwd_kernel(
    const unsigned n, const unsigned deg_use, const float *__restrict__ viewdirs,
    const float *__restrict__ coeffs, float *__restrict__ colors) {
  const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;  // (g,k) flat
  const unsigned g = e >> 4;
  const int k = (int)(e & 15u);
  const bool live = g < n;
  const unsigned gs = live ? g : (n - 1);
  float B[16];
  sh_basis<16>(deg_use, viewdirs[3 * gs], viewdirs[3 * gs + 1], viewdirs[3 * gs + 2], B);
  const float bk = pick<16>(B, k);
  const float *c = coeffs + (size_t)gs * 48 + 3 * k;
  const float r = row_sum16(bk * c[0]);
  const float gr = row_sum16(bk * c[1]);
  const float b = row_sum16(bk * c[2]);
  if (live && k == 15) {
    colors[3 * g] = r;
    colors[3 * g + 1] = gr;
    colors[3 * g + 2] = b;
  }
}

__global__ __launch_bounds__(256) void sh16_bwd_kernel(
    const unsigned n, const unsigned deg_use, const float *__restrict__ viewdirs,
    const float *__restrict__ v_colors, float *__restrict__ v_coeffs) {
  const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned g = e >> 4;
  const int k = (int)(e & 15u);
  if (g >= n) return;
  float B[16];
  sh_basis<16>(deg_use, viewdirs[3 * g], viewdirs[3 * g + 1], viewdirs[3 * g + 2], B);
  const float bk = pick<16>(B, k);
  float *o = v_coeffs + (size_t)g * 48 + 3 * k;
  o[0] = bk * v_colors[3 * g];
  o[1] = bk * v_colors[3 * g + 1];
  o[2] = bk * v_colors[3 * g + 2];
}

// ---- any degree: one lane per Gaussian -------------------------------------
template <int K>
__global__ __launch_bounds__(256) void sh_fwd_kernel(
    const unsigned n, const unsigned deg_use, const float *__restrict__ viewdirs,
    const float *__restrict__ coeffs, float *__restrict__ colors) {
  const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  float B[K];
  sh_basis<K>(deg_use, viewdirs[3 * g], viewdirs[3 * g + 1], viewdirs[3 * g + 2], B);
  const float *c = coeffs + (size_t)g * K * 3;
  float r = 0.f, gr = 0.f, b = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    r += B[k] * c[3 * k];
    gr += B[k] * c[3 * k + 1];
    b += B[k] * c[3 * k + 2];
  }
  colors[3 * g] = r;
  colors[3 * g + 1] = gr;
  colors[3 * g + 2] = b;
}

template <int K>
__global__ __launch_bounds__(256) void sh_bwd_kernel(
    const unsigned n, const unsigned deg_use, const float *__restrict__ viewdirs,
    const float *__restrict__ v_colors, float *__restrict__ v_coeffs) {
  const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  float B[K];
  sh_basis<K>(deg_use, viewdirs[3 * g], viewdirs[3 * g + 1], viewdirs[3 * g + 2], B);
  const float vr = v_colors[3 * g], vg = v_colors[3 * g + 1], vb = v_colors[3 * g + 2];
  float *o = v_coeffs + (size_t)g * K * 3;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    o[3 * k] = B[k] * vr;
    o[3 * k + 1] = B[k] * vg;
    o[3 * k + 2] = B[k] * vb;
  }
}

}  // namespace

GSR_EXPORT int gsr_sh_forward(unsigned num_points, unsigned degree, unsigned degrees_to_use,
                              const float *viewdirs, const float *coeffs, float *colors,
                              gsr_stream_t stream) {
  GSR_REQUIRE(degree <= 4, "sh_forward: degree must be in [0,4]");
  GSR_REQUIRE(degrees_to_use <= degree, "sh_forward: degrees_to_use > degree");
  if (num_points == 0) return GSR_OK;
  GSR_REQUIRE(viewdirs && coeffs && colors, "sh_forward: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const dim3 blk(256), grd(gsr_cdiv(num_points, 256));
  switch (degree) {
    case 0: hipLaunchKernelGGL(sh_fwd_kernel<1>, grd, blk, 0, s, num_points, degrees_to_use, viewdirs, coeffs, colors); break;
    case 1: hipLaunchKernelGGL(sh_fwd_kernel<4>, grd, blk, 0, s, num_points, degrees_to_use, viewdirs, coeffs, colors); break;
    case 2: hipLaunchKernelGGL(sh_fwd_kernel<9>, grd, blk, 0, s, num_points, degrees_to_use, viewdirs, coeffs, colors); break;
    case 3:
      hipLaunchKernelGGL(sh16_fwd_kernel, dim3(gsr_cdiv(num_points, 16)), blk, 0, s, num_points,
                         degrees_to_use, viewdirs, coeffs, colors);
      break;
    default: hipLaunchKernelGGL(sh_fwd_kernel<25>, grd, blk, 0, s, num_points, degrees_to_use, viewdirs, coeffs, colors); break;
  }
  GSR_CHECK_LAUNCH("sh_forward");
  return GSR_OK;
}

GSR_EXPORT int gsr_sh_backward(unsigned num_points, unsigned degree, unsigned degrees_to_use,
                               const float *viewdirs, const float *v_colors, float *v_coeffs,
                               gsr_stream_t stream) {
  GSR_REQUIRE(degree <= 4, "sh_backward: degree must be in [0,4]");
  GSR_REQUIRE(degrees_to_use <= degree, "sh_backward: degrees_to_use > degree");
  if (num_points == 0) return GSR_OK;
  GSR_REQUIRE(viewdirs && v_colors && v_coeffs, "sh_backward: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const dim3 blk(256), grd(gsr_cdiv(num_points, 256));
  switch (degree) {
    case 0: hipLaunchKernelGGL(sh_bwd_kernel<1>, grd, blk, 0, s, num_points, degrees_to_use, viewdirs, v_colors, v_coeffs); break;
    case 1: hipLaunchKernelGGL(sh_bwd_kernel<4>, grd, blk, 0, s, num_points, degrees_to_use, viewdirs, v_colors, v_coeffs); break;
    case 2: hipLaunchKernelGGL(sh_bwd_kernel<9>, grd, blk, 0, s, num_points, degrees_to_use, viewdirs, v_colors, v_coeffs); break;
    case 3:
      hipLaunchKernelGGL(sh16_bwd_kernel, dim3(gsr_cdiv(num_points, 16)), blk, 0, s, num_points,
                         degrees_to_use, viewdirs, v_colors, v_coeffs);
      break;
    default: hipLaunchKernelGGL(sh_bwd_kernel<25>, grd, blk, 0, s, num_points, degrees_to_use, viewdirs, v_colors, v_coeffs); break;
  }
  GSR_CHECK_LAUNCH("sh_backward");
  return GSR_OK;
}
