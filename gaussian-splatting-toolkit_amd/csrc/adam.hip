// adam.hip -- one launch of Adam over all parameter tensors of a Gaussian model
// ("fused per-Gaussian Adam"), gfx950.
//
// The toolkit builds one torch.optim.Adam per parameter group
// (gs_toolkit/engine/optimizers.py:59-196 with the learning rates of
// configs/method_configs.py:47-80; eps = 1e-15): six optimisers, i.e. six
// sequences of ~10 elementwise launches (or six multi-tensor launches) per
// iteration.  The update itself is pure streaming: 16 B read + 12 B written per
// element, 59 elements per Gaussian at SH degree 3 = 1.65 GB per step for 1 M
// Gaussians.  This kernel walks up to 8 tensors in one launch, dwordx4 wide.
//
// Update rule of torch.optim.Adam (amsgrad=False, maximize=False, weight_decay=0),
// torch/optim/adam.py `_single_tensor_adam`:
//   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include <cmath>

#include "gsr_common.h"

namespace {

constexpr int kMaxTensors = GSR_ADAM_MAX_TENSORS;
constexpr int kVecPerBlock = 256 * 4;  // float4 per workgroup (4 per thread)

struct AdamArgs {
  gsr_adam_tensor t[kMaxTensors];
  int first_block[kMaxTensors + 1];  // workgroups [first_block[i], first_block[i+1]) serve tensor i
  int num;
};

// c1 = 1 - b1 and c2 = 1 - b2 are formed in double on the host, as torch forms them
// in Python floats (1.f - 0.999f is 4.7e-5 off 0.001f)
struct Hyper {
  float b1, c1, b2, c2, eps, bc1, inv_bc2_sqrt;
};

__device__ __forceinline__ void adam_one(float &p, const float g, float &m, float &v, const Hyper &h,
                                         const float step_size) {
  m = h.b1 * m + h.c1 * g;
  v = h.b2 * v + h.c2 * (g * g);
  const float inv_bc2_sqrt = h.inv_bc2_sqrt, eps = h.eps;
  const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
  p -= step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adam_kernel(const AdamArgs a, const Hyper h) {
  int ti = 0;
#pragma unroll
  for (int i = 1; i < kMaxTensors; ++i)
    if (i < a.num && (int)blockIdx.x >= a.first_block[i]) ti = i;
  const gsr_adam_tensor T = a.t[ti];
  const long long blk = (long long)blockIdx.x - a.first_block[ti];
  const float step_size = T.lr / h.bc1;
  const long long n4 = T.n >> 2;
  const bool vec = ((reinterpret_cast<uintptr_t>(T.param) | reinterpret_cast<uintptr_t>(T.grad) |
                     reinterpret_cast<uintptr_t>(T.exp_avg) | reinterpret_cast<uintptr_t>(T.exp_avg_sq)) & 15) == 0;
  if (vec) {
    float4 *P = reinterpret_cast<float4 *>(T.param), *M = reinterpret_cast<float4 *>(T.exp_avg),
           *V = reinterpret_cast<float4 *>(T.exp_avg_sq);
    const float4 *G = reinterpret_cast<const float4 *>(T.grad);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long i = blk * kVecPerBlock + u * 256 + threadIdx.x;
      if (i < n4) {
        float4 p = gsr_load_stream(P + i), m = gsr_load_stream(M + i), v = gsr_load_stream(V + i);
        const float4 g = gsr_load_stream(G + i);
        adam_one(p.x, g.x, m.x, v.x, h, step_size);
        adam_one(p.y, g.y, m.y, v.y, h, step_size);
        adam_one(p.z, g.z, m.z, v.z, h, step_size);
        adam_one(p.w, g.w, m.w, v.w, h, step_size);
        gsr_store_stream(P + i, p);
        gsr_store_stream(M + i, m);
        gsr_store_stream(V + i, v);
      }
    }
    // the < 4 trailing elements: first workgroup of the tensor
    if (blk == 0 && threadIdx.x < (T.n & 3)) {
      const long long i = (n4 << 2) + threadIdx.x;
      adam_one(T.param[i], T.grad[i], T.exp_avg[i], T.exp_avg_sq[i], h, step_size);
    }
  } else {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const long long i = blk * (4 * kVecPerBlock) + u * 256 + threadIdx.x;
      if (i < T.n)
        adam_one(T.param[i], T.grad[i], T.exp_avg[i], T.exp_avg_sq[i], h, step_size);
    }
  }
}

}  // namespace

GSR_EXPORT int gsr_adam_step(int num_tensors, const gsr_adam_tensor *tensors, double beta1, double beta2,
                             double eps, long long step, gsr_stream_t stream) {
  GSR_REQUIRE(num_tensors >= 0 && num_tensors <= kMaxTensors, "adam_step: at most 8 tensors per call");
  GSR_REQUIRE(step >= 1, "adam_step: step counts from 1");
  GSR_REQUIRE(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0, "adam_step: betas must be in [0,1)");
  if (num_tensors == 0) return GSR_OK;
  GSR_REQUIRE(tensors, "adam_step: null pointer");
  AdamArgs a{};
  a.num = num_tensors;
  long long blocks = 0;
  for (int i = 0; i < num_tensors; ++i) {
    const gsr_adam_tensor &t = tensors[i];
    GSR_REQUIRE(t.n >= 0, "adam_step: negative size");
    GSR_REQUIRE(t.n == 0 || (t.param && t.grad && t.exp_avg && t.exp_avg_sq), "adam_step: null pointer");
    a.t[i] = t;
    a.first_block[i] = (int)blocks;
    blocks += (t.n + 4 * kVecPerBlock - 1) / (4 * kVecPerBlock);
    GSR_REQUIRE(blocks < (1ll << 31), "adam_step: too many elements for one launch");
  }
  a.first_block[num_tensors] = (int)blocks;
  if (blocks == 0) return GSR_OK;
  // bias corrections in double, like torch (python floats), then rounded once
  const double bc1 = 1.0 - std::pow(beta1, (double)step);
  const double bc2 = 1.0 - std::pow(beta2, (double)step);
  const Hyper h{(float)beta1, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)bc1,
                (float)(1.0 / std::sqrt(bc2))};
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, h);
  GSR_CHECK_LAUNCH("adam_step");
  return GSR_OK;
}
