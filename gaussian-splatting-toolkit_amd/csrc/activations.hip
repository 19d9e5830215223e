// activations.hip -- the per-Gaussian activations in front of the rasterizer, one
// kernel forward and one backward (caller-side glue), gfx950.
//
// What it replaces, in GaussianSplattingModel.get_outputs
// (gs_toolkit/models/vanilla_gs.py:765-826):
//     scales    = torch.exp(scales_crop)
//     quats     = quats_crop / quats_crop.norm(dim=-1, keepdim=True)
//     opacities = torch.sigmoid(opacities_crop)
//     viewdirs  = means_crop.detach() - camera_position;  viewdirs /= viewdirs.norm(...)
// i.e. ~9 elementwise / reduction launches forward and ~12 backward over N-sized
// arrays, every iteration.  HBM-bound streaming: 44 B read + 44 B written per
// Gaussian forward.  No gradient flows to the view directions (the reference
// detaches the means there).
#include "gsr_common.h"

namespace {

__global__ __launch_bounds__(256) void activate_fwd_kernel(
    const int n, const float *__restrict__ means, const float *__restrict__ log_scales,
    const float *__restrict__ raw_quats, const float *__restrict__ opacity_logits,
    const float *__restrict__ campos, float *__restrict__ scales, float *__restrict__ quats,
    float *__restrict__ opacities, float *__restrict__ viewdirs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
#pragma unroll
  for (int k = 0; k < 3; ++k) scales[3 * i + k] = expf(log_scales[3 * i + k]);
  const float4 q = reinterpret_cast<const float4 *>(raw_quats)[i];
  const float inv = 1.f / sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  reinterpret_cast<float4 *>(quats)[i] = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
  opacities[i] = 1.f / (1.f + expf(-opacity_logits[i]));
  if (viewdirs) {
    const float dx = means[3 * i] - campos[0], dy = means[3 * i + 1] - campos[1], dz = means[3 * i + 2] - campos[2];
    const float dinv = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
    viewdirs[3 * i] = dx * dinv;
    viewdirs[3 * i + 1] = dy * dinv;
    viewdirs[3 * i + 2] = dz * dinv;
  }
}

// v_log_scale = v_scale * scale;  v_raw_quat = (v_q - q (q . v_q)) / |raw|;
// v_logit = v_opacity * o (1 - o).  A missing cotangent (nullptr) is zero.
__global__ __launch_bounds__(256) void activate_bwd_kernel(
    const int n, const float *__restrict__ raw_quats, const float *__restrict__ scales,
    const float *__restrict__ quats, const float *__restrict__ opacities,
    const float *__restrict__ v_scales, const float *__restrict__ v_quats,
    const float *__restrict__ v_opacities, float *__restrict__ v_log_scales,
    float *__restrict__ v_raw_quats, float *__restrict__ v_logits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
#pragma unroll
  for (int k = 0; k < 3; ++k) v_log_scales[3 * i + k] = v_scales ? v_scales[3 * i + k] * scales[3 * i + k] : 0.f;
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  if (v_quats) {
    const float4 r = reinterpret_cast<const float4 *>(raw_quats)[i];
    const float4 q = reinterpret_cast<const float4 *>(quats)[i];
    const float4 v = reinterpret_cast<const float4 *>(v_quats)[i];
    const float inv = 1.f / sqrtf(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w);
    const float d = q.x * v.x + q.y * v.y + q.z * v.z + q.w * v.w;
    g = make_float4((v.x - q.x * d) * inv, (v.y - q.y * d) * inv, (v.z - q.z * d) * inv, (v.w - q.w * d) * inv);
  }
  reinterpret_cast<float4 *>(v_raw_quats)[i] = g;
  const float o = opacities[i];
  v_logits[i] = v_opacities ? v_opacities[i] * o * (1.f - o) : 0.f;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

GSR_EXPORT int gsr_activate_forward(int num_points, const float *means, const float *log_scales,
                                    const float *raw_quats, const float *opacity_logits,
                                    const float *camera_position, float *scales, float *quats,
                                    float *opacities, float *viewdirs, gsr_stream_t stream) {
  GSR_REQUIRE(num_points >= 0, "activate_forward: num_points < 0");
  if (num_points == 0) return GSR_OK;
  GSR_REQUIRE(log_scales && raw_quats && opacity_logits && scales && quats && opacities,
              "activate_forward: null pointer");
  GSR_REQUIRE((viewdirs == nullptr) || (means && camera_position), "activate_forward: viewdirs need means and the camera position");
  GSR_REQUIRE(aligned16(raw_quats) && aligned16(quats), "activate_forward: quaternions must be 16-byte aligned");
  hipLaunchKernelGGL(activate_fwd_kernel, dim3(gsr_cdiv(num_points, 256)), dim3(256), 0, (hipStream_t)stream,
                     num_points, means, log_scales, raw_quats, opacity_logits, camera_position, scales, quats,
                     opacities, viewdirs);
  GSR_CHECK_LAUNCH("activate_forward");
  return GSR_OK;
}

GSR_EXPORT int gsr_activate_backward(int num_points, const float *raw_quats, const float *scales,
                                     const float *quats, const float *opacities, const float *v_scales,
                                     const float *v_quats, const float *v_opacities, float *v_log_scales,
                                     float *v_raw_quats, float *v_logits, gsr_stream_t stream) {
  GSR_REQUIRE(num_points >= 0, "activate_backward: num_points < 0");
  if (num_points == 0) return GSR_OK;
  GSR_REQUIRE(raw_quats && scales && quats && opacities && v_log_scales && v_raw_quats && v_logits,
              "activate_backward: null pointer");
  GSR_REQUIRE(aligned16(raw_quats) && aligned16(quats) && aligned16(v_raw_quats) &&
                  (v_quats == nullptr || aligned16(v_quats)),
              "activate_backward: quaternions must be 16-byte aligned");
  hipLaunchKernelGGL(activate_bwd_kernel, dim3(gsr_cdiv(num_points, 256)), dim3(256), 0, (hipStream_t)stream,
                     num_points, raw_quats, scales, quats, opacities, v_scales, v_quats, v_opacities,
                     v_log_scales, v_raw_quats, v_logits);
  GSR_CHECK_LAUNCH("activate_backward");
  return GSR_OK;
}

// ---- densification statistics (GaussianSplattingModel.after_train,
// gs_toolkit/models/vanilla_gs.py:344-372): for the Gaussians visible in this view
//   xys_grad_norm += |xys.grad|;  vis_counts += 1;  max_2dsize = max(max_2dsize, radius / max(W, H))
// -- ~10 masked-indexing launches per iteration there, one here.  `first` is the
// reference's first call after a refinement (:354-356): every Gaussian starts with
// count 1 and its own gradient norm, whether visible or not.
namespace {
__global__ __launch_bounds__(256) void densify_stats_kernel(const int n, const float2 *__restrict__ v_xys,
                                                            const int *__restrict__ radii, const float inv_size,
                                                            const int first_host, const int *__restrict__ first_dev,
                                                            float *__restrict__ xys_grad_norm,
                                                            int *__restrict__ vis_counts,
                                                            float *__restrict__ max_2dsize) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int first = first_dev ? *first_dev : first_host;
  const int r = radii[i];
  float norm = 0.f;
  if (v_xys && (first || r > 0)) {
    const float2 g = v_xys[i];
    norm = sqrtf(g.x * g.x + g.y * g.y);
  }
  if (first) {
    xys_grad_norm[i] = norm;
    vis_counts[i] = 1;
    max_2dsize[i] = r > 0 ? fmaxf(0.f, (float)r * inv_size) : 0.f;
    return;
  }
  if (r <= 0) return;
  if (v_xys) xys_grad_norm[i] += norm;
  vis_counts[i] += 1;
  max_2dsize[i] = fmaxf(max_2dsize[i], (float)r * inv_size);
}
}  // namespace

GSR_EXPORT int gsr_densify_stats(int num_points, const float *v_xys, const int32_t *radii, float inv_size, int first,
                                 float *xys_grad_norm, int32_t *vis_counts, float *max_2dsize,
                                 gsr_stream_t stream) {
  GSR_REQUIRE(num_points >= 0, "densify_stats: num_points < 0");
  if (num_points == 0) return GSR_OK;
  GSR_REQUIRE(radii && xys_grad_norm && vis_counts && max_2dsize, "densify_stats: null pointer");
  GSR_REQUIRE(v_xys == nullptr || (reinterpret_cast<uintptr_t>(v_xys) & 7u) == 0, "densify_stats: v_xys must be 8-byte aligned");
  hipLaunchKernelGGL(densify_stats_kernel, dim3(gsr_cdiv(num_points, 256)), dim3(256), 0, (hipStream_t)stream,
                     num_points, reinterpret_cast<const float2 *>(v_xys), radii, inv_size, first, (const int *)nullptr,
                     xys_grad_norm, vis_counts, max_2dsize);
  GSR_CHECK_LAUNCH("densify_stats");
  return GSR_OK;
}

GSR_EXPORT int gsr_densify_stats_dev(int num_points, const float *v_xys, const int32_t *radii, float inv_size,
                                     const int32_t *first, float *xys_grad_norm, int32_t *vis_counts,
                                     float *max_2dsize, gsr_stream_t stream) {
  GSR_REQUIRE(num_points >= 0, "densify_stats_dev: num_points < 0");
  if (num_points == 0) return GSR_OK;
  GSR_REQUIRE(radii && first && xys_grad_norm && vis_counts && max_2dsize, "densify_stats_dev: null pointer");
  GSR_REQUIRE(v_xys == nullptr || (reinterpret_cast<uintptr_t>(v_xys) & 7u) == 0, "densify_stats_dev: v_xys must be 8-byte aligned");
  hipLaunchKernelGGL(densify_stats_kernel, dim3(gsr_cdiv(num_points, 256)), dim3(256), 0, (hipStream_t)stream,
                     num_points, reinterpret_cast<const float2 *>(v_xys), radii, inv_size, 0, first, xys_grad_norm,
                     vis_counts, max_2dsize);
  GSR_CHECK_LAUNCH("densify_stats_dev");
  return GSR_OK;
}
