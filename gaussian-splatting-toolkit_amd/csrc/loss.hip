// loss.hip -- fused photometric loss head  (1-lambda)*L1 + lambda*(1-SSIM)  and
// its gradient w.r.t. the rendered image (gfx950).   [SURVEY.md 8(f) row f2]
//
// What it replaces: GaussianSplattingModel.get_loss_dict
// (gs_toolkit/models/vanilla_gs.py:926-944): `torch.abs(gt - pred).mean()` and
// `1 - SSIM(data_range=1.0, size_average=True, channel=3)(gt, pred)` from the
// third-party package pytorch_msssim (pinned "1.0.0" in the reference's
// pyproject.toml:27; not vendored).  Its published algorithm, restated: 11-tap
// Gaussian window (sigma 1.5, normalised), separable, VALID padding (maps are
// (H-10) x (W-10)); mu1, mu2, E[x^2], E[y^2], E[xy]; sigma = E[..] - mu*mu;
// C1 = 0.01^2, C2 = 0.03^2;
//   S = (2 mu1 mu2 + C1)(2 s12 + C2) / ((mu1^2 + mu2^2 + C1)(s11 + s22 + C2)),
// mean over channels and positions.
//
// As torch ops this is 5 depthwise blurs (10 conv launches, MIOpen picks slow
// grouped-conv kernels: 2.3 ms forward + ~1.5 ms backward at 1080p, more than the
// whole rasterizer) plus ~20 elementwise kernels.  Here: ONE forward kernel
// (both images staged once per tile in LDS, the five blurs done separably from
// LDS, S and the three partial derivatives dS/dmu1, dS/dE[x^2], dS/dE[xy] written
// as planar maps, the two sums reduced to double atomics) and ONE backward kernel
// (transposed blur of the three maps + the L1 sign term -> d loss / d pred, which
// is exactly the `v_out_img` the compositing backward consumes).
#include <algorithm>

#include "gsr_common.h"

namespace {

constexpr int kWin = 11;
constexpr int kHalo = kWin - 1;
constexpr int kTW = 32, kTH = 16;                  // tile of outputs per workgroup
constexpr int kIW = kTW + kHalo, kIH = kTH + kHalo;  // staged inputs

// normalised 11-tap Gaussian, sigma = 1.5 (pytorch_msssim._fspecial_gauss_1d)
__constant__ float kW[kWin] = {0.00102838f, 0.00759876f, 0.03600077f, 0.10936069f, 0.21300553f,
                               0.26601172f, 0.21300553f, 0.10936069f, 0.03600077f, 0.00759876f,
                               0.00102838f};
constexpr float kC1 = 0.01f * 0.01f, kC2 = 0.03f * 0.03f;

// img: [H,W,3] interleaved.  maps: [3 derivative kinds][3 channels][Hv][Wv] planar.
// sums[0] += sum |x-y| over the tile's own pixels, sums[1] += sum S over its valid outputs.
// clamp_pred: the prediction is min(pred, 1) (the models clamp the rendered image at 1
// before the loss, vanilla_gs.py:857 `torch.clamp(rgb, max=1.0)`; folding it in here
// saves that op and its three-kernel backward).
__global__ __launch_bounds__(256) void l1_ssim_fwd_kernel(
    const int H, const int W, const float lambda, const int clamp_pred, const float *__restrict__ pred,
    const float *__restrict__ gt, float *__restrict__ maps, double *__restrict__ sums) {
  __shared__ float sx[kIH][kIW + 1], sy[kIH][kIW + 1];
  __shared__ float hb[5][kIH][kTW + 1];
  __shared__ float red[2][4];

  const int Hv = H - kHalo, Wv = W - kHalo;
  const int c = blockIdx.z;
  const int ox = blockIdx.x * kTW, oy = blockIdx.y * kTH;
  const int tid = threadIdx.x;

  float l1 = 0.f;
  for (int i = tid; i < kIH * kIW; i += 256) {
    const int r = i / kIW, q = i % kIW;
    const int gy = oy + r, gx = ox + q;
    float x = 0.f, y = 0.f;
    if (gy < H && gx < W) {
      const size_t o = ((size_t)gy * W + gx) * 3 + c;
      x = pred[o];
      if (clamp_pred) x = fminf(x, 1.f);
      y = gt[o];
      if (r < kTH && q < kTW) l1 += fabsf(x - y);  // every pixel belongs to exactly one tile
    }
    sx[r][q] = x;
    sy[r][q] = y;
  }
  __syncthreads();

  // horizontal pass of the five quantities
  for (int i = tid; i < kIH * kTW; i += 256) {
    const int r = i / kTW, q = i % kTW;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
    for (int k = 0; k < kWin; ++k) {
      const float x = sx[r][q + k], y = sy[r][q + k], w = kW[k];
      a0 += w * x;
      a1 += w * y;
      a2 += w * (x * x);
      a3 += w * (y * y);
      a4 += w * (x * y);
    }
    hb[0][r][q] = a0;
    hb[1][r][q] = a1;
    hb[2][r][q] = a2;
    hb[3][r][q] = a3;
    hb[4][r][q] = a4;
  }
  __syncthreads();

  // vertical pass + SSIM and its partials: 512 outputs, 2 per lane
  float ssum = 0.f;
  const size_t plane = (size_t)Hv * Wv;
  for (int i = tid; i < kTH * kTW; i += 256) {
    const int r = i / kTW, q = i % kTW;
    const int gy = oy + r, gx = ox + q;
    if (gy >= Hv || gx >= Wv) continue;
    float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int k = 0; k < kWin; ++k) {
      const float w = kW[k];
      mu1 += w * hb[0][r + k][q];
      mu2 += w * hb[1][r + k][q];
      e11 += w * hb[2][r + k][q];
      e22 += w * hb[3][r + k][q];
      e12 += w * hb[4][r + k][q];
    }
    const float s11 = e11 - mu1 * mu1, s22 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
    const float A1 = 2.f * mu1 * mu2 + kC1, A2 = 2.f * s12 + kC2;
    const float B1 = mu1 * mu1 + mu2 * mu2 + kC1, B2 = s11 + s22 + kC2;
    const float inv = 1.f / (B1 * B2);
    const float S = A1 * A2 * inv;
    ssum += S;
    // d S / d (mu1, E[x^2], E[xy]) with x = pred
    const float dA1 = A2 * inv, dA2 = A1 * inv, dB1 = -S / B1, dB2 = -S / B2;
    const float d_mu = dA1 * (2.f * mu2) + dB1 * (2.f * mu1) + dA2 * (-2.f * mu2) + dB2 * (-2.f * mu1);
    const size_t o = (size_t)c * plane + (size_t)gy * Wv + gx;
    maps[o] = d_mu;
    maps[3 * plane + o] = dB2;        // d S / d E[x^2]
    maps[6 * plane + o] = 2.f * dA2;  // d S / d E[xy]
  }

  // block reduction -> two double atomics
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    l1 += __shfl_xor(l1, o);
    ssum += __shfl_xor(ssum, o);
  }
  if ((tid & 63) == 0) {
    red[0][tid >> 6] = l1;
    red[1][tid >> 6] = ssum;
  }
  __syncthreads();
  if (tid == 0) {
    // GSR_LOSS_SUM_SLOTS partial sums per quantity: 12 k workgroups adding doubles to the
    // same two addresses serialised in the L2 (that alone was ~250 us of this kernel)
    const unsigned linear = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned slot = linear % GSR_LOSS_SUM_SLOTS;
    atomicAdd(&sums[slot], (double)red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(&sums[GSR_LOSS_SUM_SLOTS + slot], (double)red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

// The partial sums -> the three scalars.  A separate one-wave launch: doing it in the
// workgroup that finishes last needs an agent-scope release fence in every workgroup,
// and on gfx950 that fence writes the XCD's whole L2 back (measured: 80 -> 520 us).
__global__ __launch_bounds__(64) void l1_ssim_finalize_kernel(const int H, const int W, const float lambda,
                                                              const double *__restrict__ sums,
                                                              float *__restrict__ loss_out,
                                                              float *__restrict__ terms_out) {
  const int tid = threadIdx.x;
  double a = 0.0, b = 0.0;
  for (int k = tid; k < GSR_LOSS_SUM_SLOTS; k += 64) {
    a += sums[k];
    b += sums[GSR_LOSS_SUM_SLOTS + k];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o);
    b += __shfl_xor(b, o);
  }
  if (tid == 0) {
    const double l1m = a / (3.0 * (double)H * (double)W);
    const double ssm = b / (3.0 * (double)(H - kHalo) * (double)(W - kHalo));
    *loss_out = (float)((1.0 - (double)lambda) * l1m + (double)lambda * (1.0 - ssm));
    if (terms_out) {
      terms_out[0] = (float)l1m;
      terms_out[1] = (float)ssm;
    }
  }
}

// v_pred = up * [ (1-lambda) sign(x-y)/(3HW) - lambda/(3 Hv Wv) * ( blurT(Dmu) + 2x blurT(D11) + y blurT(D12) ) ]
__global__ __launch_bounds__(256) void l1_ssim_bwd_kernel(
    const int H, const int W, const float lambda, const int clamp_pred, const float *__restrict__ upstream,
    const float *__restrict__ pred, const float *__restrict__ gt, const float *__restrict__ maps,
    float *__restrict__ v_pred) {
  __shared__ float sm[3][kIH][kIW + 1];
  __shared__ float hb[3][kIH][kTW + 1];

  const int Hv = H - kHalo, Wv = W - kHalo;
  const int c = blockIdx.z;
  const int ox = blockIdx.x * kTW, oy = blockIdx.y * kTH;  // tile of image pixels
  const int tid = threadIdx.x;
  const size_t plane = (size_t)Hv * Wv;

  // map entries (i-k, j-l), k,l in [0,10]  ->  rows oy-10 .. oy+15, cols ox-10 .. ox+31
  for (int i = tid; i < kIH * kIW; i += 256) {
    const int r = i / kIW, q = i % kIW;
    const int my = oy - kHalo + r, mx = ox - kHalo + q;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f;
    if (my >= 0 && my < Hv && mx >= 0 && mx < Wv) {
      const size_t o = (size_t)c * plane + (size_t)my * Wv + mx;
      d0 = maps[o];
      d1 = maps[3 * plane + o];
      d2 = maps[6 * plane + o];
    }
    sm[0][r][q] = d0;
    sm[1][r][q] = d1;
    sm[2][r][q] = d2;
  }
  __syncthreads();
  // pixel column j gathers map columns j-l with weight w[l]: staged index q + kHalo - l
  for (int i = tid; i < kIH * kTW; i += 256) {
    const int r = i / kTW, q = i % kTW;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int l = 0; l < kWin; ++l) {
      const float w = kW[l];
      a0 += w * sm[0][r][q + kHalo - l];
      a1 += w * sm[1][r][q + kHalo - l];
      a2 += w * sm[2][r][q + kHalo - l];
    }
    hb[0][r][q] = a0;
    hb[1][r][q] = a1;
    hb[2][r][q] = a2;
  }
  __syncthreads();
  const float up = upstream[0];
  const float k_l1 = up * (1.f - lambda) / (3.f * (float)H * (float)W);
  const float k_ss = -up * lambda / (3.f * (float)Hv * (float)Wv);
  for (int i = tid; i < kTH * kTW; i += 256) {
    const int r = i / kTW, q = i % kTW;
    const int gy = oy + r, gx = ox + q;
    if (gy >= H || gx >= W) continue;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
#pragma unroll
    for (int k = 0; k < kWin; ++k) {
      const float w = kW[k];
      g0 += w * hb[0][r + kHalo - k][q];
      g1 += w * hb[1][r + kHalo - k][q];
      g2 += w * hb[2][r + kHalo - k][q];
    }
    const size_t o = ((size_t)gy * W + gx) * 3 + c;
    const float xr = pred[o], y = gt[o];
    const float x = clamp_pred ? fminf(xr, 1.f) : xr;
    const float d = x - y;
    const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    const float g = k_l1 * sgn + k_ss * (g0 + 2.f * x * g1 + y * g2);
    v_pred[o] = (clamp_pred && xr > 1.f) ? 0.f : g;  // clamp(max=1) passes the gradient where pred <= 1
  }
}

}  // namespace

GSR_EXPORT int gsr_l1_ssim_forward(unsigned img_height, unsigned img_width, float ssim_lambda,
                                   int clamp_pred, const float *pred, const float *gt, float *maps,
                                   double *sums, float *loss_out, float *terms_out, gsr_stream_t stream) {
  GSR_REQUIRE(img_height > kHalo && img_width > kHalo, "l1_ssim_forward: image must be larger than 10x10");
  GSR_REQUIRE(pred && gt && maps && sums && loss_out, "l1_ssim_forward: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (int zrc = gsr_zero_async(sums, GSR_LOSS_WORKSPACE_DOUBLES * sizeof(double), s)) return zrc;
  const dim3 grd(gsr_cdiv(img_width, kTW), gsr_cdiv(img_height, kTH), 3);
  hipLaunchKernelGGL(l1_ssim_fwd_kernel, grd, dim3(256), 0, s, (int)img_height, (int)img_width, ssim_lambda,
                     clamp_pred, pred, gt, maps, sums);
  hipLaunchKernelGGL(l1_ssim_finalize_kernel, dim3(1), dim3(64), 0, s, (int)img_height, (int)img_width,
                     ssim_lambda, (const double *)sums, loss_out, terms_out);
  GSR_CHECK_LAUNCH("l1_ssim_forward");
  return GSR_OK;
}

GSR_EXPORT int gsr_l1_ssim_backward(unsigned img_height, unsigned img_width, float ssim_lambda,
                                    int clamp_pred, const float *upstream, const float *pred, const float *gt,
                                    const float *maps, float *v_pred, gsr_stream_t stream) {
  GSR_REQUIRE(img_height > kHalo && img_width > kHalo, "l1_ssim_backward: image must be larger than 10x10");
  GSR_REQUIRE(upstream && pred && gt && maps && v_pred, "l1_ssim_backward: null pointer");
  const dim3 grd(gsr_cdiv(img_width, kTW), gsr_cdiv(img_height, kTH), 3);
  hipLaunchKernelGGL(l1_ssim_bwd_kernel, grd, dim3(256), 0, (hipStream_t)stream, (int)img_height,
                     (int)img_width, ssim_lambda, clamp_pred, upstream, pred, gt, maps, v_pred);
  GSR_CHECK_LAUNCH("l1_ssim_backward");
  return GSR_OK;
}

// ---- L1-only photometric head: what DepthGSModel's `main_loss` really is --------
// depth_gs.py:445-448 reads
//     loss_dict["main_loss"] = (1 - self.config.ssim_lambda) * Ll1
//     +self.config.ssim_lambda * simloss
// -- the second line is a stand-alone expression statement, so the SSIM term is computed
// and thrown away: the co-gs photometric loss is  weight * mean |gt - pred|  with
// weight = 1 - ssim_lambda.  One streaming kernel each way, no maps, no blur.
namespace {

__global__ __launch_bounds__(256) void l1_fwd_kernel(const long long n4, const long long n, const int clamp_pred,
                                                     const float *__restrict__ pred, const float *__restrict__ gt,
                                                     double *__restrict__ sums) {
  __shared__ float red[4];
  float acc = 0.f;
  const float4 *p4 = reinterpret_cast<const float4 *>(pred);
  const float4 *g4 = reinterpret_cast<const float4 *>(gt);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 x = p4[i];
    const float4 y = g4[i];
    if (clamp_pred) {
      x.x = fminf(x.x, 1.f);
      x.y = fminf(x.y, 1.f);
      x.z = fminf(x.z, 1.f);
      x.w = fminf(x.w, 1.f);
    }
    acc += (fabsf(x.x - y.x) + fabsf(x.y - y.y)) + (fabsf(x.z - y.z) + fabsf(x.w - y.w));
  }
  if (blockIdx.x == 0) {  // the (at most three) elements behind the last float4
    const long long i = 4 * n4 + threadIdx.x;
    if (i < n) {
      const float x = clamp_pred ? fminf(pred[i], 1.f) : pred[i];
      acc += fabsf(x - gt[i]);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0)
    atomicAdd(&sums[blockIdx.x % GSR_LOSS_SUM_SLOTS], (double)red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(64) void l1_finalize_kernel(const long long n, const float weight,
                                                         const double *__restrict__ sums,
                                                         float *__restrict__ loss_out) {
  double a = 0.0;
  for (int k = threadIdx.x; k < GSR_LOSS_SUM_SLOTS; k += 64) a += sums[k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
  if (threadIdx.x == 0) *loss_out = (float)((double)weight * (a / (double)n));
}

__global__ __launch_bounds__(256) void l1_bwd_kernel(const long long n, const float weight, const int clamp_pred,
                                                     const float *__restrict__ upstream,
                                                     const float *__restrict__ pred, const float *__restrict__ gt,
                                                     float *__restrict__ v_pred) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float xr = pred[i];
  const float x = clamp_pred ? fminf(xr, 1.f) : xr;
  const float d = x - gt[i];
  const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
  const float k = upstream[0] * weight / (float)n;
  v_pred[i] = (clamp_pred && xr > 1.f) ? 0.f : k * sgn;
}

}  // namespace

GSR_EXPORT int gsr_l1_forward(long long num_values, float weight, int clamp_pred, const float *pred,
                              const float *gt, double *sums, float *loss_out, gsr_stream_t stream) {
  GSR_REQUIRE(num_values > 0, "l1_forward: empty image");
  GSR_REQUIRE(pred && gt && sums && loss_out, "l1_forward: null pointer");
  GSR_REQUIRE(((uintptr_t)pred & 15) == 0 && ((uintptr_t)gt & 15) == 0, "l1_forward: images must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  if (int zrc = gsr_zero_async(sums, GSR_LOSS_SUM_SLOTS * sizeof(double), s)) return zrc;
  const long long n4 = num_values / 4;
  const unsigned blocks = (unsigned)std::max<long long>(1, std::min<long long>((n4 + 1023) / 1024, 4096));
  hipLaunchKernelGGL(l1_fwd_kernel, dim3(blocks), dim3(256), 0, s, n4, num_values, clamp_pred, pred, gt, sums);
  hipLaunchKernelGGL(l1_finalize_kernel, dim3(1), dim3(64), 0, s, num_values, weight, (const double *)sums,
                     loss_out);
  GSR_CHECK_LAUNCH("l1_forward");
  return GSR_OK;
}

GSR_EXPORT int gsr_l1_backward(long long num_values, float weight, int clamp_pred, const float *upstream,
                               const float *pred, const float *gt, float *v_pred, gsr_stream_t stream) {
  GSR_REQUIRE(num_values > 0, "l1_backward: empty image");
  GSR_REQUIRE(upstream && pred && gt && v_pred, "l1_backward: null pointer");
  hipLaunchKernelGGL(l1_bwd_kernel, dim3((unsigned)((num_values + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     num_values, weight, clamp_pred, upstream, pred, gt, v_pred);
  GSR_CHECK_LAUNCH("l1_backward");
  return GSR_OK;
}

// ---- depth head of the co-gs model (DepthGSModel, gs_toolkit/models/depth_gs.py) --
//   pred  = where(alpha > 0, depth / alpha, depth.detach().max())      (:356-363)
//   loss  = | gt * (gt > 0) - pred * (gt > 0) |.mean()                   (:531-538)
// as one kernel forward (+ the one-wave sum) and one backward that writes the
// cotangents of the two compositing passes' outputs (accumulated depth, alpha)
// directly -- ~8 elementwise / indexing launches forward and ~10 backward otherwise.
namespace {

__global__ __launch_bounds__(256) void depth_l1_fwd_kernel(const long long n, const float *__restrict__ depth,
                                                           const float *__restrict__ alpha,
                                                           const float *__restrict__ gt,
                                                           const float *__restrict__ depth_max,
                                                           double *__restrict__ sums) {
  __shared__ float red[4];
  const float far = *depth_max;
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float a = alpha[i], g = gt[i];
    const float pred = a > 0.f ? depth[i] / a : far;
    acc += g > 0.f ? fabsf(g - pred) : 0.f;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0)
    atomicAdd(&sums[blockIdx.x % GSR_LOSS_SUM_SLOTS], (double)red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(64) void depth_l1_finalize_kernel(const long long n, const double *__restrict__ sums,
                                                               float *__restrict__ loss_out) {
  double a = 0.0;
  for (int k = threadIdx.x; k < GSR_LOSS_SUM_SLOTS; k += 64) a += sums[k];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
  if (threadIdx.x == 0) *loss_out = (float)(a / (double)n);
}

__global__ __launch_bounds__(256) void depth_l1_bwd_kernel(const long long n, const float *__restrict__ upstream,
                                                           const float *__restrict__ depth,
                                                           const float *__restrict__ alpha,
                                                           const float *__restrict__ gt,
                                                           const float *__restrict__ depth_max,
                                                           float *__restrict__ v_depth, float *__restrict__ v_alpha) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float a = alpha[i], g = gt[i], d = depth[i];
  float vd = 0.f, va = 0.f;
  if (a > 0.f && g > 0.f) {  // the far value of uncovered pixels is detached
    const float inv = 1.f / a;
    const float pred = d * inv;
    const float s = pred > g ? 1.f : (pred < g ? -1.f : 0.f);
    const float vp = upstream[0] * s / (float)n;
    vd = vp * inv;
    va = -vp * pred * inv;
  }
  v_depth[i] = vd;
  v_alpha[i] = va;
}

}  // namespace

GSR_EXPORT int gsr_depth_l1_forward(long long num_pixels, const float *depth, const float *alpha, const float *gt,
                                    const float *depth_max, double *sums, float *loss_out, gsr_stream_t stream) {
  GSR_REQUIRE(num_pixels > 0, "depth_l1_forward: empty image");
  GSR_REQUIRE(depth && alpha && gt && depth_max && sums && loss_out, "depth_l1_forward: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (int zrc = gsr_zero_async(sums, GSR_LOSS_SUM_SLOTS * sizeof(double), s)) return zrc;
  const unsigned blocks = (unsigned)std::min<long long>((num_pixels + 1023) / 1024, 4096);
  hipLaunchKernelGGL(depth_l1_fwd_kernel, dim3(blocks), dim3(256), 0, s, num_pixels, depth, alpha, gt, depth_max,
                     sums);
  hipLaunchKernelGGL(depth_l1_finalize_kernel, dim3(1), dim3(64), 0, s, num_pixels, (const double *)sums, loss_out);
  GSR_CHECK_LAUNCH("depth_l1_forward");
  return GSR_OK;
}

GSR_EXPORT int gsr_depth_l1_backward(long long num_pixels, const float *upstream, const float *depth,
                                     const float *alpha, const float *gt, const float *depth_max, float *v_depth,
                                     float *v_alpha, gsr_stream_t stream) {
  GSR_REQUIRE(num_pixels > 0, "depth_l1_backward: empty image");
  GSR_REQUIRE(upstream && depth && alpha && gt && depth_max && v_depth && v_alpha, "depth_l1_backward: null pointer");
  hipLaunchKernelGGL(depth_l1_bwd_kernel, dim3((unsigned)((num_pixels + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, num_pixels, upstream, depth, alpha, gt, depth_max, v_depth, v_alpha);
  GSR_CHECK_LAUNCH("depth_l1_backward");
  return GSR_OK;
}
