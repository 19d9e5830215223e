// sh.hip -- view-dependent colour from real spherical harmonics, and its VJP
// w.r.t. the coefficients (gfx950).
//
// Behaviour restated from rasterizer/cuda/csrc/sh.cuh:33-224: svox2 sign
// convention, view direction normalised in-kernel, 3 channels, coefficient
// layout [n, K, 3] (basis-major), bands above `degrees_to_use` ignored
// (forward) / zero (backward).
//
// HBM-bound streaming op; the coefficient tensor is the largest per-Gaussian
// read of the whole path (192 B at degree 3).  One lane per Gaussian; at K == 16
// with 16-byte-aligned buffers each lane moves its 192 contiguous bytes as 12
// dwordx4 accesses (every fetched 128-B line is fully consumed by the same lane
// a few instructions later, so L1/L2 turn the stride into full-line traffic).
// Design notes from tools/exp/shbench.hip on MI355X, 3 M Gaussians (576 MB,
// larger than the 256 MiB Infinity Cache): lane-per-Gaussian dwordx4 4.5 TB/s,
// 16-lanes-per-Gaussian dwordx3 + DPP row reduction 3.6 TB/s; the backward is
// write-bound at 2.2-2.4 TB/s either way.
#include "gsr_common.h"

#include <stdlib.h>

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

// Basis vector for direction d (normalised here); bands above `deg` are zero.
// Written branch-free over constant indices on purpose: with early returns the
// compiler keeps B[] as a memory object and promotes it to LDS, which turned
// this streaming op into an LDS-bank-conflict-bound one (measured 0.9 TB/s
// instead of 3.6-4.5 TB/s).
template <int KMAX>
__device__ __forceinline__ void sh_basis(unsigned deg, float dx, float dy, float dz,
                                         float (&B)[KMAX]) {
  const float inv = rsqrtf(dx * dx + dy * dy + dz * dz);
  const float x = dx * inv, y = dy * inv, z = dz * inv;
  const float xx = x * x, xy = x * y, xz = x * z, yy = y * y, yz = y * z, zz = z * z;
  const float m1 = deg >= 1 ? 1.f : 0.f, m2 = deg >= 2 ? 1.f : 0.f;
  const float m3 = deg >= 3 ? 1.f : 0.f, m4 = deg >= 4 ? 1.f : 0.f;
  B[0] = C0;
  if constexpr (KMAX >= 4) {
    B[1] = m1 * (-C1 * y);
    B[2] = m1 * (C1 * z);
    B[3] = m1 * (-C1 * x);
  }
  if constexpr (KMAX >= 9) {
    B[4] = m2 * (C2_0 * xy);
    B[5] = m2 * (C2_1 * yz);
    B[6] = m2 * (C2_2 * (2.f * zz - xx - yy));
    B[7] = m2 * (C2_3 * xz);
    B[8] = m2 * (C2_4 * (xx - yy));
  }
  if constexpr (KMAX >= 16) {
    B[9] = m3 * (C3_0 * y * (3.f * xx - yy));
    B[10] = m3 * (C3_1 * xy * z);
    B[11] = m3 * (C3_2 * y * (4.f * zz - xx - yy));
    B[12] = m3 * (C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy));
    B[13] = m3 * (C3_4 * x * (4.f * zz - xx - yy));
    B[14] = m3 * (C3_5 * z * (xx - yy));
    B[15] = m3 * (C3_6 * x * (xx - 3.f * yy));
  }
  if constexpr (KMAX >= 25) {
    B[16] = m4 * (C4_0 * xy * (xx - yy));
    B[17] = m4 * (C4_1 * yz * (3.f * xx - yy));
    B[18] = m4 * (C4_2 * xy * (7.f * zz - 1.f));
    B[19] = m4 * (C4_3 * yz * (7.f * zz - 3.f));
    B[20] = m4 * (C4_4 * (zz * (35.f * zz - 30.f) + 3.f));
    B[21] = m4 * (C4_5 * xz * (7.f * zz - 3.f));
    B[22] = m4 * (C4_6 * (xx - yy) * (7.f * zz - 1.f));
    B[23] = m4 * (C4_7 * xz * (xx - 3.f * yy));
    B[24] = m4 * (C4_8 * (xx * (xx - 3.f * yy) - yy * (3.f * xx - yy)));
  }
}
// (a zero direction gives NaN in the bands >= 1 when they are in use, exactly
//  like the reference's x / norm; with deg == 0 the masks multiply NaN by 0 ->
//  guard: the degree-0 term never touches the direction)

// ---- K == 16, 16-byte aligned coefficients ----------------------------------
// One lane per Gaussian for the arithmetic, but the coefficients move as the
// wave's 64 x 192 contiguous bytes in 12 fully coalesced 1-KB dwordx4 rows and are
// transposed through LDS (per-Gaussian rows padded to 13 float4: 2-way bank
// conflicts at most).  A lane streaming its own 192 bytes with 12 dwordx4 accesses
// touches every 64-byte sector from four different instructions and, for the
// backward's stores, writes four partial sectors: tools/exp/shbench.hip on 3 M
// Gaussians (576 MB): forward 4.36 -> 5.18 TB/s, backward 2.2 -> 5.4 TB/s.
constexpr int kShRow = 13;  // float4 per Gaussian row in LDS (12 + 1 padding)

// WAVES per workgroup: the waves never talk to each other (each transposes its own 64 rows), so the
// workgroup size only sets the LDS granule of the scheduler: 13 KB per wave.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void sh16_fwd_kernel(
    const unsigned n, const unsigned deg_use, const float *__restrict__ viewdirs,
    const float4 *__restrict__ coeffs, float *__restrict__ colors) {
  __shared__ float4 lds[WAVES][64 * kShRow];
  const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned g0 = (blockIdx.x * WAVES + w) * 64;  // first Gaussian of this wave
  const unsigned navail = g0 < n ? (n - g0 < 64 ? n - g0 : 64) * 12 : 0;
  const float4 *src = coeffs + (size_t)g0 * 12;
  // the view direction goes out with the coefficient rows (after the barrier it is a second round trip per wave)
  const unsigned gd = g0 + lane < n ? g0 + lane : n - 1;
  const float dx = viewdirs[3 * gd], dy = viewdirs[3 * gd + 1], dz = viewdirs[3 * gd + 2];
  float4 q[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const unsigned j = i * 64 + lane;
    q[i] = j < navail ? gsr_load_stream(src + j) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const unsigned j = i * 64 + lane;
    lds[w][(j / 12) * kShRow + j % 12] = q[i];
  }
  __syncthreads();
  const unsigned g = g0 + lane;
  if (g >= n) return;
#pragma unroll
  for (int i = 0; i < 12; ++i) q[i] = lds[w][lane * kShRow + i];
  float B[16];
  sh_basis<16>(deg_use, dx, dy, dz, B);
  if (deg_use == 0) {
#pragma unroll
    for (int k = 1; k < 16; ++k) B[k] = 0.f;
  }
  const float f[48] = {
      q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w, q[2].x, q[2].y, q[2].z, q[2].w,
      q[3].x, q[3].y, q[3].z, q[3].w, q[4].x, q[4].y, q[4].z, q[4].w, q[5].x, q[5].y, q[5].z, q[5].w,
      q[6].x, q[6].y, q[6].z, q[6].w, q[7].x, q[7].y, q[7].z, q[7].w, q[8].x, q[8].y, q[8].z, q[8].w,
      q[9].x, q[9].y, q[9].z, q[9].w, q[10].x, q[10].y, q[10].z, q[10].w, q[11].x, q[11].y, q[11].z, q[11].w};
  float r = 0.f, gr = 0.f, b = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    r += B[k] * f[3 * k];
    gr += B[k] * f[3 * k + 1];
    b += B[k] * f[3 * k + 2];
  }
  colors[3 * g] = r;
  colors[3 * g + 1] = gr;
  colors[3 * g + 2] = b;
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void sh16_bwd_kernel(
    const unsigned n, const unsigned deg_use, const float *__restrict__ viewdirs,
    const float *__restrict__ v_colors, float4 *__restrict__ v_coeffs) {
  __shared__ float4 lds[WAVES][64 * kShRow];
  const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned g0 = (blockIdx.x * WAVES + w) * 64;
  const unsigned g = g0 + lane;
  if (g < n) {
    float B[16];
    sh_basis<16>(deg_use, viewdirs[3 * g], viewdirs[3 * g + 1], viewdirs[3 * g + 2], B);
    if (deg_use == 0) {
#pragma unroll
      for (int k = 1; k < 16; ++k) B[k] = 0.f;
    }
    const float vr = v_colors[3 * g], vg = v_colors[3 * g + 1], vb = v_colors[3 * g + 2];
    float f[48];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      f[3 * k] = B[k] * vr;
      f[3 * k + 1] = B[k] * vg;
      f[3 * k + 2] = B[k] * vb;
    }
#pragma unroll
    for (int i = 0; i < 12; ++i)
      lds[w][lane * kShRow + i] = make_float4(f[4 * i], f[4 * i + 1], f[4 * i + 2], f[4 * i + 3]);
  }
  __syncthreads();
  const unsigned navail = g0 < n ? (n - g0 < 64 ? n - g0 : 64) * 12 : 0;
  float4 *dst = v_coeffs + (size_t)g0 * 12;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const unsigned j = i * 64 + lane;
    if (j < navail) gsr_store_stream(dst + j, lds[w][(j / 12) * kShRow + j % 12]);
  }
}

// ---- any degree / any alignment: one lane per Gaussian, dword accesses -----
template <int K>
__global__ __launch_bounds__(256) void sh_fwd_kernel(
    const unsigned n, const unsigned deg_use, const float *__restrict__ viewdirs,
    const float *__restrict__ coeffs, float *__restrict__ colors) {
  const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  float B[K];
  sh_basis<K>(deg_use, viewdirs[3 * g], viewdirs[3 * g + 1], viewdirs[3 * g + 2], B);
  if (deg_use == 0) {
#pragma unroll
    for (int k = 1; k < K; ++k) B[k] = 0.f;
  }
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
  if (deg_use == 0) {
#pragma unroll
    for (int k = 1; k < K; ++k) B[k] = 0.f;
  }
  const float vr = v_colors[3 * g], vg = v_colors[3 * g + 1], vb = v_colors[3 * g + 2];
  float *o = v_coeffs + (size_t)g * K * 3;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    o[3 * k] = B[k] * vr;
    o[3 * k + 1] = B[k] * vg;
    o[3 * k + 2] = B[k] * vb;
  }
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

// ---- split coefficients: features_dc [N,3] + features_rest [N,K-1,3] ---------
// The models keep the DC band and the higher bands as separate parameters and
// torch.cat them for every render (vanilla_gs.py:809): at K = 16 that is a 192-MB
// copy per forward and two strided copies back per backward, more than the SH
// kernels themselves.  These kernels read / write the two tensors in place.  Same
// scheme as the K = 16 kernels above: the wave's 64 x (K-1) x 3 contiguous floats of
// `rest` move as coalesced dwordx4 and are transposed through LDS (row stride padded
// to an odd number of floats: conflict-free dword reads).
template <int K>
struct SplitCfg {
  static constexpr int R = 3 * (K - 1);  // floats of `rest` per Gaussian
  static constexpr int STRIDE = R | 1;
};

template <int K, bool VEC>
__device__ __forceinline__ void split_rows_to_lds(const float *__restrict__ src, const unsigned cnt,
                                                  const unsigned lane, float *lds) {
  constexpr int R = SplitCfg<K>::R, STRIDE = SplitCfg<K>::STRIDE;
  if (VEC && cnt == 64) {
    const float4 *s4 = reinterpret_cast<const float4 *>(src);
#pragma unroll
    for (int i = 0; i < (16 * R + 63) / 64; ++i) {
      const unsigned j = i * 64 + lane;
      if (j < 16u * R) {
        const float4 q = gsr_load_stream(s4 + j);
        const float v[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const unsigned e = 4 * j + c;
          lds[(e / R) * STRIDE + e % R] = v[c];
        }
      }
    }
  } else {
    for (unsigned e = lane; e < cnt * R; e += 64) lds[(e / R) * STRIDE + e % R] = src[e];
  }
}

template <int K, bool VEC>
__device__ __forceinline__ void split_rows_from_lds(float *__restrict__ dst, const unsigned cnt,
                                                    const unsigned lane, const float *lds) {
  constexpr int R = SplitCfg<K>::R, STRIDE = SplitCfg<K>::STRIDE;
  if (VEC && cnt == 64) {
    float4 *d4 = reinterpret_cast<float4 *>(dst);
#pragma unroll
    for (int i = 0; i < (16 * R + 63) / 64; ++i) {
      const unsigned j = i * 64 + lane;
      if (j < 16u * R) {
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const unsigned e = 4 * j + c;
          v[c] = lds[(e / R) * STRIDE + e % R];
        }
        gsr_store_stream(d4 + j, make_float4(v[0], v[1], v[2], v[3]));
      }
    }
  } else {
    for (unsigned e = lane; e < cnt * R; e += 64) dst[e] = lds[(e / R) * STRIDE + e % R];
  }
}

template <int K, bool VEC>
__global__ __launch_bounds__(256) void sh_split_fwd_kernel(
    const unsigned n, const unsigned deg_use, const float *__restrict__ viewdirs,
    const float *__restrict__ dc, const float *__restrict__ rest, float *__restrict__ colors,
    const float shift, const int clamp_zero) {
  constexpr int R = SplitCfg<K>::R, STRIDE = SplitCfg<K>::STRIDE;
  __shared__ float lds[4][64 * STRIDE];
  const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned g0 = (blockIdx.x * 4 + w) * 64;
  const unsigned cnt = g0 < n ? (n - g0 < 64 ? n - g0 : 64) : 0;
  // view direction and DC term go out before the rows (after the barrier they are a second round trip per wave)
  const unsigned gd = g0 + lane < n ? g0 + lane : n - 1;
  const float dx = viewdirs[3 * gd], dy = viewdirs[3 * gd + 1], dz = viewdirs[3 * gd + 2];
  const float dc0 = dc[3 * gd], dc1 = dc[3 * gd + 1], dc2 = dc[3 * gd + 2];
  split_rows_to_lds<K, VEC>(rest + (size_t)g0 * R, cnt, lane, lds[w]);
  __syncthreads();
  const unsigned g = g0 + lane;
  if (g >= n) return;
  float B[K];
  sh_basis<K>(deg_use, dx, dy, dz, B);
  if (deg_use == 0) {
#pragma unroll
    for (int k = 1; k < K; ++k) B[k] = 0.f;
  }
  float r = B[0] * dc0, gr = B[0] * dc1, b = B[0] * dc2;
  const float *row = lds[w] + lane * STRIDE;
#pragma unroll
  for (int k = 1; k < K; ++k) {
    r += B[k] * row[3 * (k - 1)];
    gr += B[k] * row[3 * (k - 1) + 1];
    b += B[k] * row[3 * (k - 1) + 2];
  }
  // optional epilogue of the models: rgb = clamp(sh + 0.5, min=0) (vanilla_gs.py:826)
  r += shift;
  gr += shift;
  b += shift;
  if (clamp_zero) {
    // a channel the clamp cut is stored as -0.0 (== 0 for every consumer), a channel at exactly
    // zero as +0.0: the backward tells them apart, because torch.clamp(x, min=0) passes the
    // gradient where x >= 0 and blocks it where x < 0
    r = r < 0.f ? -0.f : r + 0.f;
    gr = gr < 0.f ? -0.f : gr + 0.f;
    b = b < 0.f ? -0.f : b + 0.f;
  }
  colors[3 * g] = r;
  colors[3 * g + 1] = gr;
  colors[3 * g + 2] = b;
}

template <int K, bool VEC>
__global__ __launch_bounds__(256) void sh_split_bwd_kernel(
    const unsigned n, const unsigned deg_use, const float *__restrict__ viewdirs,
    const float *__restrict__ v_colors, float *__restrict__ v_dc, float *__restrict__ v_rest,
    const float *__restrict__ clamped_colors) {
  constexpr int R = SplitCfg<K>::R, STRIDE = SplitCfg<K>::STRIDE;
  __shared__ float lds[4][64 * STRIDE];
  const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned g0 = (blockIdx.x * 4 + w) * 64;
  const unsigned cnt = g0 < n ? (n - g0 < 64 ? n - g0 : 64) : 0;
  const unsigned g = g0 + lane;
  if (g < n) {
    float B[K];
    sh_basis<K>(deg_use, viewdirs[3 * g], viewdirs[3 * g + 1], viewdirs[3 * g + 2], B);
    if (deg_use == 0) {
#pragma unroll
      for (int k = 1; k < K; ++k) B[k] = 0.f;
    }
    float vr = v_colors[3 * g], vg = v_colors[3 * g + 1], vb = v_colors[3 * g + 2];
    if (clamped_colors) {  // forward output of the clamped epilogue: no gradient where it cut (-0.0)
      vr = __float_as_uint(clamped_colors[3 * g]) == 0x80000000u ? 0.f : vr;
      vg = __float_as_uint(clamped_colors[3 * g + 1]) == 0x80000000u ? 0.f : vg;
      vb = __float_as_uint(clamped_colors[3 * g + 2]) == 0x80000000u ? 0.f : vb;
    }
    v_dc[3 * g] = B[0] * vr;
    v_dc[3 * g + 1] = B[0] * vg;
    v_dc[3 * g + 2] = B[0] * vb;
    float *row = lds[w] + lane * STRIDE;
#pragma unroll
    for (int k = 1; k < K; ++k) {
      row[3 * (k - 1)] = B[k] * vr;
      row[3 * (k - 1) + 1] = B[k] * vg;
      row[3 * (k - 1) + 2] = B[k] * vb;
    }
  }
  __syncthreads();
  split_rows_from_lds<K, VEC>(v_rest + (size_t)g0 * R, cnt, lane, lds[w]);
}

// degree 0 (K = 1, `rest` is [n, 0, 3]): colours = C0 dc + shift, same -0.0 marking of clamped channels
__global__ __launch_bounds__(256) void sh_dc_fwd_kernel(const unsigned n3, const float *__restrict__ dc,
                                                        float *__restrict__ colors, const float shift,
                                                        const int clamp_zero) {
  // product and sum rounded separately, like the degree-0 kernel followed by the models' `+ 0.5`
  // (HIP's __fmul_rn / __fadd_rn are plain operators the compiler may still contract: the pragma is what holds)
#pragma clang fp contract(off)
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n3) return;
  float v = C0 * dc[i];
  v = v + shift;
  if (clamp_zero) v = v < 0.f ? -0.f : v + 0.f;
  colors[i] = v;
}
__global__ __launch_bounds__(256) void sh_dc_bwd_kernel(const unsigned n3, const float *__restrict__ v_colors,
                                                        const float *__restrict__ clamped_colors,
                                                        float *__restrict__ v_dc) {
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n3) return;
  const bool cut = clamped_colors && __float_as_uint(clamped_colors[i]) == 0x80000000u;
  v_dc[i] = cut ? 0.f : C0 * v_colors[i];
}

GSR_EXPORT int gsr_sh_forward_split(unsigned num_points, unsigned degree, unsigned degrees_to_use,
                                    const float *viewdirs, const float *dc, const float *rest,
                                    float *colors, float shift, int clamp_zero, gsr_stream_t stream) {
  GSR_REQUIRE(degree <= 3, "sh_forward_split: degree must be in [0,3]");
  GSR_REQUIRE(degrees_to_use <= degree, "sh_forward_split: degrees_to_use > degree");
  if (num_points == 0) return GSR_OK;
  hipStream_t s = (hipStream_t)stream;
  if (degree == 0) {  // the DC band only: no direction, no `rest`
    GSR_REQUIRE(dc && colors, "sh_forward_split: null pointer");
    GSR_REQUIRE(num_points < (1u << 30), "sh_forward_split: too many points");
    hipLaunchKernelGGL(sh_dc_fwd_kernel, dim3(gsr_cdiv(3 * num_points, 256)), dim3(256), 0, s, 3 * num_points, dc,
                       colors, shift, clamp_zero);
    GSR_CHECK_LAUNCH("sh_forward_split");
    return GSR_OK;
  }
  GSR_REQUIRE(viewdirs && dc && rest && colors, "sh_forward_split: null pointer");
  const dim3 blk(256), grd(gsr_cdiv(num_points, 256));
  const bool vec = aligned16(rest);
#define GSR_SPLIT_FWD(KK)                                                                                  \
  if (vec)                                                                                                 \
    hipLaunchKernelGGL((sh_split_fwd_kernel<KK, true>), grd, blk, 0, s, num_points, degrees_to_use,        \
                       viewdirs, dc, rest, colors, shift, clamp_zero);                                     \
  else                                                                                                     \
    hipLaunchKernelGGL((sh_split_fwd_kernel<KK, false>), grd, blk, 0, s, num_points, degrees_to_use,       \
                       viewdirs, dc, rest, colors, shift, clamp_zero)
  switch (degree) {
    case 1: GSR_SPLIT_FWD(4); break;
    case 2: GSR_SPLIT_FWD(9); break;
    default: GSR_SPLIT_FWD(16); break;
  }
#undef GSR_SPLIT_FWD
  GSR_CHECK_LAUNCH("sh_forward_split");
  return GSR_OK;
}

GSR_EXPORT int gsr_sh_backward_split(unsigned num_points, unsigned degree, unsigned degrees_to_use,
                                     const float *viewdirs, const float *v_colors,
                                     const float *clamped_colors, float *v_dc, float *v_rest,
                                     gsr_stream_t stream) {
  GSR_REQUIRE(degree <= 3, "sh_backward_split: degree must be in [0,3]");
  GSR_REQUIRE(degrees_to_use <= degree, "sh_backward_split: degrees_to_use > degree");
  if (num_points == 0) return GSR_OK;
  hipStream_t s = (hipStream_t)stream;
  if (degree == 0) {
    GSR_REQUIRE(v_colors && v_dc, "sh_backward_split: null pointer");
    GSR_REQUIRE(num_points < (1u << 30), "sh_backward_split: too many points");
    hipLaunchKernelGGL(sh_dc_bwd_kernel, dim3(gsr_cdiv(3 * num_points, 256)), dim3(256), 0, s, 3 * num_points,
                       v_colors, clamped_colors, v_dc);
    GSR_CHECK_LAUNCH("sh_backward_split");
    return GSR_OK;
  }
  GSR_REQUIRE(viewdirs && v_colors && v_dc && v_rest, "sh_backward_split: null pointer");
  const dim3 blk(256), grd(gsr_cdiv(num_points, 256));
  const bool vec = aligned16(v_rest);
#define GSR_SPLIT_BWD(KK)                                                                                  \
  if (vec)                                                                                                 \
    hipLaunchKernelGGL((sh_split_bwd_kernel<KK, true>), grd, blk, 0, s, num_points, degrees_to_use,        \
                       viewdirs, v_colors, v_dc, v_rest, clamped_colors);                                  \
  else                                                                                                     \
    hipLaunchKernelGGL((sh_split_bwd_kernel<KK, false>), grd, blk, 0, s, num_points, degrees_to_use,       \
                       viewdirs, v_colors, v_dc, v_rest, clamped_colors)
  switch (degree) {
    case 1: GSR_SPLIT_BWD(4); break;
    case 2: GSR_SPLIT_BWD(9); break;
    default: GSR_SPLIT_BWD(16); break;
  }
#undef GSR_SPLIT_BWD
  GSR_CHECK_LAUNCH("sh_backward_split");
  return GSR_OK;
}

// ---- SH backward over several views at once (the data-parallel exchange of harness/parallel.py) ---------
// The SH gradient of one view is rank one per Gaussian: v_coeffs[g, k, :] = B_k(dir_g) v_colors[g, :].  Ranks that
// render different views therefore need not all-reduce 12 K bytes per Gaussian: they all-gather their 12-byte
// v_colors (and camera positions) and every rank forms  scale * sum_r B_k(normalize(mean_g - campos_r)) v_colors_r[g]
// itself, views in rank order: the same sum on every rank, bit for bit.
// Output rows go through LDS so that a wave writes 64 Gaussians' rows as one contiguous run.
// kJoint: v_coeffs [n, K, 3]; else v_dc [n, 3] + v_rest [n, K - 1, 3] (K >= 2).
template <int K, bool kJoint>
__global__ __launch_bounds__(256) void sh_bwd_views_kernel(const unsigned n, const unsigned deg_use,
                                                           const unsigned num_views,
                                                           const float *__restrict__ means,
                                                           const float *__restrict__ campos, const size_t cstride,
                                                           const float *__restrict__ v_colors, const size_t vstride,
                                                           const float scale, float *__restrict__ v_dc,
                                                           float *__restrict__ v_rest) {
  constexpr int F = 3 * K, STRIDE = F | 1;
  __shared__ float lds[4][64 * STRIDE];
  const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned g0 = (blockIdx.x * 4 + w) * 64;
  const unsigned cnt = g0 < n ? (n - g0 < 64 ? n - g0 : 64) : 0;
  const unsigned g = g0 + lane < n ? g0 + lane : n - 1;
  const float mx = means[3 * g], my = means[3 * g + 1], mz = means[3 * g + 2];
  float f[F];
#pragma unroll
  for (int j = 0; j < F; ++j) f[j] = 0.f;
  for (unsigned r = 0; r < num_views; ++r) {
    const float *vc = v_colors + (size_t)r * vstride + (size_t)g * 3, *cp = campos + (size_t)r * cstride;
    const float vr = vc[0], vg = vc[1], vb = vc[2];
    // the direction as gsr_activate_forward forms it (activations.hip)
    const float dx = mx - cp[0], dy = my - cp[1], dz = mz - cp[2];
    const float dinv = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
    float B[K];
    sh_basis<K>(deg_use, dx * dinv, dy * dinv, dz * dinv, B);
    if (deg_use == 0) {
#pragma unroll
      for (int k = 1; k < K; ++k) B[k] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      f[3 * k] += B[k] * vr;
      f[3 * k + 1] += B[k] * vg;
      f[3 * k + 2] += B[k] * vb;
    }
  }
  float *row = lds[w] + lane * STRIDE;
#pragma unroll
  for (int j = 0; j < F; ++j) row[j] = f[j] * scale;
  __syncthreads();
  if (kJoint) {
    float *dst = v_rest + (size_t)g0 * F;
    for (unsigned e = lane; e < cnt * F; e += 64) dst[e] = lds[w][(e / F) * STRIDE + e % F];
  } else {
    constexpr int R = F - 3;
    float *d0 = v_dc + (size_t)g0 * 3, *d1 = v_rest + (size_t)g0 * R;
    for (unsigned e = lane; e < cnt * 3; e += 64) d0[e] = lds[w][(e / 3) * STRIDE + e % 3];
    for (unsigned e = lane; e < cnt * R; e += 64) d1[e] = lds[w][(e / R) * STRIDE + 3 + e % R];
  }
}

GSR_EXPORT int gsr_sh_backward_views(unsigned num_points, unsigned degree, unsigned degrees_to_use, unsigned num_views,
                                     const float *means3d, const float *camera_positions, size_t camera_stride,
                                     const float *v_colors, size_t v_colors_stride, float scale, float *v_dc,
                                     float *v_rest, float *v_coeffs, gsr_stream_t stream) {
  GSR_REQUIRE(degree <= 3, "sh_backward_views: degree must be in [0,3]");
  GSR_REQUIRE(degrees_to_use <= degree, "sh_backward_views: degrees_to_use > degree");
  GSR_REQUIRE(num_views >= 1, "sh_backward_views: num_views < 1");
  GSR_REQUIRE((v_coeffs != nullptr) != (v_dc != nullptr), "sh_backward_views: give v_coeffs, or v_dc (+ v_rest)");
  if (num_points == 0) return GSR_OK;
  GSR_REQUIRE(means3d && camera_positions && v_colors, "sh_backward_views: null pointer");
  GSR_REQUIRE(v_coeffs || degree == 0 || v_rest, "sh_backward_views: v_rest is NULL");
  hipStream_t s = (hipStream_t)stream;
  const dim3 blk(256), grd(gsr_cdiv(num_points, 256));
#define GSR_VIEWS(KK)                                                                                          \
  if (v_coeffs)                                                                                                \
    hipLaunchKernelGGL((sh_bwd_views_kernel<KK, true>), grd, blk, 0, s, num_points, degrees_to_use, num_views, \
                       means3d, camera_positions, camera_stride, v_colors, v_colors_stride, scale,             \
                       (float *)nullptr, v_coeffs);                                                            \
  else                                                                                                         \
    hipLaunchKernelGGL((sh_bwd_views_kernel<KK, false>), grd, blk, 0, s, num_points, degrees_to_use, num_views, \
                       means3d, camera_positions, camera_stride, v_colors, v_colors_stride, scale, v_dc, v_rest)
  switch (degree) {
    case 0:  // (K = 1: the joint layout [n, 1, 3] and v_dc [n, 3] are the same bytes)
      hipLaunchKernelGGL((sh_bwd_views_kernel<1, true>), grd, blk, 0, s, num_points, degrees_to_use, num_views, means3d,
                         camera_positions, camera_stride, v_colors, v_colors_stride, scale, (float *)nullptr,
                         v_coeffs ? v_coeffs : v_dc);
      break;
    case 1: GSR_VIEWS(4); break;
    case 2: GSR_VIEWS(9); break;
    default: GSR_VIEWS(16); break;
  }
#undef GSR_VIEWS
  GSR_CHECK_LAUNCH("sh_backward_views");
  return GSR_OK;
}

// waves per workgroup of the K = 16 kernels (1 / 2 / 4 measured in round 3: 4)
static int sh16_waves() { return 4; }

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
      if (aligned16(coeffs)) {
        const int wv = sh16_waves();
        if (wv == 1)
          hipLaunchKernelGGL(sh16_fwd_kernel<1>, dim3(gsr_cdiv(num_points, 64)), dim3(64), 0, s, num_points,
                             degrees_to_use, viewdirs, reinterpret_cast<const float4 *>(coeffs), colors);
        else if (wv == 2)
          hipLaunchKernelGGL(sh16_fwd_kernel<2>, dim3(gsr_cdiv(num_points, 128)), dim3(128), 0, s, num_points,
                             degrees_to_use, viewdirs, reinterpret_cast<const float4 *>(coeffs), colors);
        else
          hipLaunchKernelGGL(sh16_fwd_kernel<4>, grd, blk, 0, s, num_points, degrees_to_use, viewdirs,
                             reinterpret_cast<const float4 *>(coeffs), colors);
      } else
        hipLaunchKernelGGL(sh_fwd_kernel<16>, grd, blk, 0, s, num_points, degrees_to_use, viewdirs, coeffs, colors);
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
      if (aligned16(v_coeffs)) {
        const int wv = sh16_waves();
        if (wv == 1)
          hipLaunchKernelGGL(sh16_bwd_kernel<1>, dim3(gsr_cdiv(num_points, 64)), dim3(64), 0, s, num_points,
                             degrees_to_use, viewdirs, v_colors, reinterpret_cast<float4 *>(v_coeffs));
        else if (wv == 2)
          hipLaunchKernelGGL(sh16_bwd_kernel<2>, dim3(gsr_cdiv(num_points, 128)), dim3(128), 0, s, num_points,
                             degrees_to_use, viewdirs, v_colors, reinterpret_cast<float4 *>(v_coeffs));
        else
          hipLaunchKernelGGL(sh16_bwd_kernel<4>, grd, blk, 0, s, num_points, degrees_to_use, viewdirs, v_colors,
                             reinterpret_cast<float4 *>(v_coeffs));
      } else
        hipLaunchKernelGGL(sh_bwd_kernel<16>, grd, blk, 0, s, num_points, degrees_to_use, viewdirs, v_colors, v_coeffs);
      break;
    default: hipLaunchKernelGGL(sh_bwd_kernel<25>, grd, blk, 0, s, num_points, degrees_to_use, viewdirs, v_colors, v_coeffs); break;
  }
  GSR_CHECK_LAUNCH("sh_backward");
  return GSR_OK;
}
