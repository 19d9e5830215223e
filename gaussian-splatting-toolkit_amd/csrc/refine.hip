// refine.hip -- Gaussian refinement (densify / split / duplicate / cull) as ONE
// compaction over the parameter tensors and both Adam moments (SURVEY.md 8f row f1,
// second half), gfx950.
//
// What the toolkit does every `refine_every` iterations
// (GaussianSplattingModel.refinement_after, gs_toolkit/models/vanilla_gs.py:381-497,
// with split_gaussians :540-592, dup_gaussians :594-603, cull_gaussians :499-538,
// dup_in_optim :303-337, remove_from_optim :282-301): ~25 boolean-mask indexing ops,
// six `torch.cat`s of three pieces each, twelve `torch.cat`s of optimizer state with
// fresh zero tensors, then a second round of mask indexing over all eighteen tensors
// to drop the culled rows -- every one a separate launch with its own nonzero() +
// host sync, and every tensor is copied twice (concatenate, then cull).
//
// Here the *decisions* are separated from the *data movement*:
//   gsr_refine_plan   one pass over the per-Gaussian scalars (log-scales, opacity
//                     logit, the three densification statistics): which originals
//                     survive, whose split children survive, whose duplicate
//                     survives; exclusive scans of those flags give every surviving
//                     row its final position.  Three small kernels, 16 B / Gaussian.
//   gsr_refine_apply  one launch that streams every input tensor once and writes
//                     each surviving row straight to its final position in the
//                     output tensor (originals, then the split children sample by
//                     sample, then the duplicates: the order the reference's
//                     cat + cull produces), computing the split children's means
//                     and shrunk scales on the way and zero-filling the Adam moments
//                     of new rows.  Bytes moved: read N rows, write N' rows, once.
//
// The rule followed (including its quirks):
//   high   = (xys_grad_norm / vis_counts) * 0.5 * max(W,H) > densify_grad_thresh
//   split  = (max exp(scale) > densify_size_thresh  [| max_2dsize > split_screen_size]) & high
//   the split Gaussian's own log-scale is shrunk IN PLACE to log(exp(s) / 1.6) (:568)
//   BEFORE the duplicates are chosen, so
//   dup    = (max exp(scale') <= densify_size_thresh) & high        (scale' = after the shrink)
//   a split original is always culled; the others by sigmoid(opacity) < cull_alpha_thresh
//   and, when cull_big, max exp(scale') > cull_scale_thresh [| max_2dsize > cull_screen_size];
//   new rows carry max_2dsize = 0 and zero Adam moments.
// Random offsets of the split children: N(0,1) samples, either handed in (the
// reference's `torch.randn((samps * n_splits, 3))`, row j * n_splits + rank(i)) or,
// with samples == NULL, generated in the kernel by Philox4x32-10 keyed on
// (seed; Gaussian index, sample index) + Box-Muller: no dependence on how many
// Gaussians split, identical on every data-parallel replica given the same seed.
#include "gsr_common.h"

namespace {

constexpr int kBlock = 256;
constexpr float kSizeFac = 1.6f;  // vanilla_gs.py:564

// flags byte
constexpr unsigned kKeepOrig = 1u, kKeepSplit = 2u, kKeepDup = 4u, kIsSplit = 8u;

struct Cfg {
  float grad_thresh, size_thresh, split_screen, alpha_thresh, scale_thresh, cull_screen, half_max_dim;
  int densify, split_by_screen, cull_big, cull_by_screen;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ unsigned classify(const int i, const float *__restrict__ log_scales,
                                             const float *__restrict__ logits, const float *__restrict__ gn,
                                             const int *__restrict__ vc, const float *__restrict__ m2, const Cfg c) {
  const float s0 = log_scales[3 * i], s1 = log_scales[3 * i + 1], s2 = log_scales[3 * i + 2];
  const float e0 = expf(s0), e1 = expf(s1), e2 = expf(s2);
  const float emax = fmaxf(e0, fmaxf(e1, e2));
  const bool alpha_cull = sigmoidf_(logits[i]) < c.alpha_thresh;
  const float screen = m2 ? m2[i] : 0.f;
  if (!c.densify) {
    const bool big = c.cull_big && (emax > c.scale_thresh || (c.cull_by_screen && screen > c.cull_screen));
    return (alpha_cull || big) ? 0u : kKeepOrig;
  }
  const float avg = (gn[i] / (float)vc[i]) * c.half_max_dim;
  const bool high = avg > c.grad_thresh;  // NaN (0/0) compares false
  bool split = emax > c.size_thresh;
  if (c.split_by_screen) split = split || screen > c.split_screen;
  split = split && high;
  float emax_after = emax;
  if (split) {
    emax_after = fmaxf(expf(logf(e0 / kSizeFac)), fmaxf(expf(logf(e1 / kSizeFac)), expf(logf(e2 / kSizeFac))));
  }
  const bool dup = (emax_after <= c.size_thresh) && high;
  const bool child_big = c.cull_big && emax_after > c.scale_thresh;
  const bool orig_big = c.cull_big && (emax_after > c.scale_thresh || (c.cull_by_screen && screen > c.cull_screen));
  unsigned f = 0;
  if (!(alpha_cull || split || orig_big)) f |= kKeepOrig;
  if (split) f |= kIsSplit;
  if (split && !alpha_cull && !child_big) f |= kKeepSplit;
  if (dup && !alpha_cull && !child_big) f |= kKeepDup;
  return f;
}

__device__ __forceinline__ int4 flag_counts(unsigned f) {
  return make_int4((f & kKeepOrig) ? 1 : 0, (f & kKeepSplit) ? 1 : 0, (f & kKeepDup) ? 1 : 0, (f & kIsSplit) ? 1 : 0);
}
__device__ __forceinline__ int4 add4(int4 a, int4 b) { return make_int4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

__device__ __forceinline__ int4 wave_inclusive(int4 v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int4 o;
    o.x = __shfl_up(v.x, d, 64);
    o.y = __shfl_up(v.y, d, 64);
    o.z = __shfl_up(v.z, d, 64);
    o.w = __shfl_up(v.w, d, 64);
    if ((int)(threadIdx.x & 63) >= d) v = add4(v, o);
  }
  return v;
}

// inclusive scan over the 256 threads of a workgroup; `total` = sum over the workgroup
__device__ __forceinline__ int4 block_inclusive(int4 v, int4 &total) {
  __shared__ int4 wsum[kBlock / 64];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  v = wave_inclusive(v);
  __syncthreads();  // wsum may still be read from a previous call
  if (l == 63) wsum[w] = v;
  __syncthreads();
  int4 base = make_int4(0, 0, 0, 0), t = make_int4(0, 0, 0, 0);
#pragma unroll
  for (int k = 0; k < kBlock / 64; ++k) {
    if (k < w) base = add4(base, wsum[k]);
    t = add4(t, wsum[k]);
  }
  total = t;
  return add4(v, base);
}

// pass 1: flags + per-workgroup totals
__global__ __launch_bounds__(kBlock) void refine_classify_kernel(const int n, const float *__restrict__ log_scales,
                                                                 const float *__restrict__ logits,
                                                                 const float *__restrict__ gn, const int *__restrict__ vc,
                                                                 const float *__restrict__ m2, const Cfg c,
                                                                 uint8_t *__restrict__ flags, int4 *__restrict__ block_sums) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  unsigned f = 0;
  if (i < n) {
    f = classify(i, log_scales, logits, gn, vc, m2, c);
    flags[i] = (uint8_t)f;
  }
  int4 total;
  block_inclusive(flag_counts(f), total);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// pass 2: one workgroup scans the per-workgroup totals in place (exclusive) and writes
// counts = {kept originals, kept split sources, kept duplicates, all split sources}
__global__ __launch_bounds__(kBlock) void refine_scan_blocks_kernel(const int num_blocks, int4 *__restrict__ block_sums,
                                                                    int *__restrict__ counts) {
  int4 carry = make_int4(0, 0, 0, 0);
  for (int base = 0; base < num_blocks; base += kBlock) {
    const int b = base + threadIdx.x;
    const int4 v = b < num_blocks ? block_sums[b] : make_int4(0, 0, 0, 0);
    int4 total;
    const int4 inc = block_inclusive(v, total);
    if (b < num_blocks) {
      int4 ex = add4(carry, inc);
      ex.x -= v.x, ex.y -= v.y, ex.z -= v.z, ex.w -= v.w;
      block_sums[b] = ex;
    }
    carry = add4(carry, total);
  }
  if (threadIdx.x == 0) {
    counts[0] = carry.x;
    counts[1] = carry.y;
    counts[2] = carry.z;
    counts[3] = carry.w;
  }
}

// pass 3: exclusive offsets of every Gaussian
__global__ __launch_bounds__(kBlock) void refine_offsets_kernel(const int n, const uint8_t *__restrict__ flags,
                                                                const int4 *__restrict__ block_sums,
                                                                int4 *__restrict__ offsets) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  const int4 v = flag_counts(i < n ? flags[i] : 0u);
  int4 total;
  int4 inc = block_inclusive(v, total);
  if (i < n) {
    const int4 b = block_sums[blockIdx.x];
    offsets[i] = make_int4(b.x + inc.x - v.x, b.y + inc.y - v.y, b.z + inc.z - v.z, b.w + inc.w - v.w);
  }
}

// ---- Philox4x32-10 (Salmon et al., SC'11) + Box-Muller; mirrored in oracle/refine.py ----
__device__ __forceinline__ void philox_round(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3, uint32_t k0,
                                             uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1,
                 n3 = (uint32_t)p0;
  c0 = n0, c1 = n1, c2 = n2, c3 = n3;
}

__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 9) + 0.5f) * 1.1920928955078125e-07f; }

__device__ __forceinline__ void split_normals(const uint64_t seed, const uint32_t i, const uint32_t j, float z[3]) {
  uint32_t c0 = i, c1 = j, c2 = 0, c3 = 0, k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const float r0 = sqrtf(-2.f * logf(u01(c0))), r1 = sqrtf(-2.f * logf(u01(c2)));
  const float a0 = 6.283185307179586f * u01(c1), a1 = 6.283185307179586f * u01(c3);
  z[0] = r0 * cosf(a0);
  z[1] = r0 * sinf(a0);
  z[2] = r1 * cosf(a1);
}

// ---- data movement -----------------------------------------------------------
constexpr int kMaxTensors = GSR_REFINE_MAX_TENSORS;
struct ApplyArgs {
  gsr_refine_tensor t[kMaxTensors];
  int first_block[kMaxTensors + 1];
  int num;
};

constexpr int kElemsPerBlock = kBlock * 4;

__global__ __launch_bounds__(kBlock) void refine_apply_kernel(const ApplyArgs a, const int n_in, const int n_samples,
                                                              const uint8_t *__restrict__ flags,
                                                              const int4 *__restrict__ offsets,
                                                              const int *__restrict__ counts,
                                                              const float *__restrict__ log_scales,
                                                              const float *__restrict__ raw_quats,
                                                              const float *__restrict__ samples, const uint64_t seed) {
  int ti = 0;
#pragma unroll
  for (int k = 1; k < kMaxTensors; ++k)
    if (k < a.num && (int)blockIdx.x >= a.first_block[k]) ti = k;
  const gsr_refine_tensor T = a.t[ti];
  const long long blk = (long long)blockIdx.x - a.first_block[ti];
  const int w = T.width;
  const long long total = (long long)n_in * w;
  const int kept_orig = counts[0], kept_split = counts[1], n_split = counts[3];
  const long long dup_base = (long long)kept_orig + (long long)n_samples * kept_split;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long long e = blk * kElemsPerBlock + u * kBlock + threadIdx.x;
    if (e >= total) continue;
    const int i = (int)(e / w), c = (int)(e - (long long)i * w);
    const unsigned f = flags[i];
    if ((f & (kKeepOrig | kKeepSplit | kKeepDup)) == 0) continue;
    const int4 off = offsets[i];
    const float v = T.in[e];
    if (f & kKeepOrig) T.out[(long long)off.x * w + c] = v;  // never a split Gaussian: value unchanged
    float child = v;  // value the new rows inherit
    if (T.kind == GSR_REFINE_MOMENT) child = 0.f;
    if (T.kind == GSR_REFINE_LOG_SCALES && (f & kIsSplit)) child = logf(expf(v) / kSizeFac);
    if (f & kKeepDup) T.out[(dup_base + off.z) * w + c] = child;
    if (f & kKeepSplit) {
      if (T.kind == GSR_REFINE_MEANS) {
        // mean + R(q / |q|) (exp(s) * z), row c of R (rasterizer/_torch_impl.py:116-138)
        float qw = raw_quats[4 * i], qx = raw_quats[4 * i + 1], qy = raw_quats[4 * i + 2], qz = raw_quats[4 * i + 3];
        const float inv = 1.f / fmaxf(sqrtf(qw * qw + qx * qx + qy * qy + qz * qz), 1e-12f);
        qw *= inv, qx *= inv, qy *= inv, qz *= inv;
        float r0, r1, r2;
        if (c == 0) {
          r0 = 1.f - 2.f * (qy * qy + qz * qz), r1 = 2.f * (qx * qy - qw * qz), r2 = 2.f * (qx * qz + qw * qy);
        } else if (c == 1) {
          r0 = 2.f * (qx * qy + qw * qz), r1 = 1.f - 2.f * (qx * qx + qz * qz), r2 = 2.f * (qy * qz - qw * qx);
        } else {
          r0 = 2.f * (qx * qz - qw * qy), r1 = 2.f * (qy * qz + qw * qx), r2 = 1.f - 2.f * (qx * qx + qy * qy);
        }
        const float e0 = expf(log_scales[3 * i]), e1 = expf(log_scales[3 * i + 1]), e2 = expf(log_scales[3 * i + 2]);
        for (int j = 0; j < n_samples; ++j) {
          float z[3];
          if (samples) {
            const float *zp = samples + 3ll * ((long long)j * n_split + off.w);
            z[0] = zp[0], z[1] = zp[1], z[2] = zp[2];
          } else {
            split_normals(seed, (uint32_t)i, (uint32_t)j, z);
          }
          const float rot = r0 * (e0 * z[0]) + r1 * (e1 * z[1]) + r2 * (e2 * z[2]);
          T.out[((long long)kept_orig + (long long)j * kept_split + off.y) * w + c] = rot + v;
        }
      } else {
        for (int j = 0; j < n_samples; ++j)
          T.out[((long long)kept_orig + (long long)j * kept_split + off.y) * w + c] = child;
      }
    }
  }
}

}  // namespace

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

GSR_EXPORT size_t gsr_refine_workspace_bytes(int num_points) {
  const size_t blocks = gsr_cdiv((unsigned)(num_points > 0 ? num_points : 1), kBlock);
  return align256(blocks * sizeof(int4)) + 256;
}

GSR_EXPORT int gsr_refine_plan(int num_points, const float *log_scales, const float *opacity_logits,
                               const float *xys_grad_norm, const int32_t *vis_counts, const float *max_2dsize,
                               const gsr_refine_config *cfg, uint8_t *flags, int32_t *offsets, int32_t *counts,
                               void *workspace, size_t workspace_bytes, gsr_stream_t stream) {
  GSR_REQUIRE(num_points >= 1, "refine_plan: num_points < 1");
  GSR_REQUIRE(cfg && log_scales && opacity_logits && flags && offsets && counts && workspace,
              "refine_plan: null pointer");
  GSR_REQUIRE(!cfg->densify || (xys_grad_norm && vis_counts), "refine_plan: densification needs the statistics");
  GSR_REQUIRE(!((cfg->densify && cfg->split_by_screen_size) || (cfg->cull_big && cfg->cull_by_screen_size)) ||
                  max_2dsize,
              "refine_plan: the screen-size rules need max_2dsize");
  GSR_REQUIRE(workspace_bytes >= gsr_refine_workspace_bytes(num_points), "refine_plan: workspace too small");
  GSR_REQUIRE((reinterpret_cast<uintptr_t>(offsets) & 15) == 0 && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
              "refine_plan: offsets / workspace must be 16-byte aligned");
  GSR_REQUIRE(cfg->n_split_samples >= 1, "refine_plan: n_split_samples < 1");
  const Cfg c{cfg->densify_grad_thresh, cfg->densify_size_thresh, cfg->split_screen_size, cfg->cull_alpha_thresh,
              cfg->cull_scale_thresh, cfg->cull_screen_size, cfg->half_max_dim,
              cfg->densify, cfg->split_by_screen_size, cfg->cull_big, cfg->cull_big && cfg->cull_by_screen_size};
  const unsigned blocks = gsr_cdiv((unsigned)num_points, kBlock);
  int4 *block_sums = reinterpret_cast<int4 *>(workspace);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(refine_classify_kernel, dim3(blocks), dim3(kBlock), 0, s, num_points, log_scales, opacity_logits,
                     xys_grad_norm, vis_counts, max_2dsize, c, flags, block_sums);
  hipLaunchKernelGGL(refine_scan_blocks_kernel, dim3(1), dim3(kBlock), 0, s, (int)blocks, block_sums, counts);
  hipLaunchKernelGGL(refine_offsets_kernel, dim3(blocks), dim3(kBlock), 0, s, num_points, flags, block_sums,
                     reinterpret_cast<int4 *>(offsets));
  GSR_CHECK_LAUNCH("refine_plan");
  return GSR_OK;
}

GSR_EXPORT int gsr_refine_apply(int num_points, int n_split_samples, const uint8_t *flags, const int32_t *offsets,
                                const int32_t *counts, const float *log_scales, const float *raw_quats,
                                const float *samples, unsigned long long seed, int num_tensors,
                                const gsr_refine_tensor *tensors, gsr_stream_t stream) {
  GSR_REQUIRE(num_points >= 1 && n_split_samples >= 1, "refine_apply: num_points / n_split_samples < 1");
  GSR_REQUIRE(num_tensors >= 0 && num_tensors <= kMaxTensors, "refine_apply: too many tensors for one call");
  if (num_tensors == 0) return GSR_OK;
  GSR_REQUIRE(flags && offsets && counts && tensors, "refine_apply: null pointer");
  ApplyArgs a{};
  a.num = num_tensors;
  long long blocks = 0;
  for (int k = 0; k < num_tensors; ++k) {
    const gsr_refine_tensor &t = tensors[k];
    GSR_REQUIRE(t.in && t.width >= 1, "refine_apply: bad tensor");
    GSR_REQUIRE(t.kind >= GSR_REFINE_COPY && t.kind <= GSR_REFINE_MOMENT, "refine_apply: unknown tensor kind");
    GSR_REQUIRE(t.kind != GSR_REFINE_MEANS || (t.width == 3 && log_scales && raw_quats),
                "refine_apply: the means need log_scales and raw_quats (width 3)");
    GSR_REQUIRE(t.kind != GSR_REFINE_LOG_SCALES || t.width == 3, "refine_apply: log-scales have width 3");
    a.t[k] = t;
    a.first_block[k] = (int)blocks;
    blocks += ((long long)num_points * t.width + kElemsPerBlock - 1) / kElemsPerBlock;
    GSR_REQUIRE(blocks < (1ll << 31), "refine_apply: too many elements for one launch");
  }
  a.first_block[num_tensors] = (int)blocks;
  hipLaunchKernelGGL(refine_apply_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, a, num_points,
                     n_split_samples, flags, reinterpret_cast<const int4 *>(offsets), counts, log_scales, raw_quats,
                     samples, (uint64_t)seed);
  GSR_CHECK_LAUNCH("refine_apply");
  return GSR_OK;
}
