// raster_bwd.hip -- VJP of the per-tile alpha compositing (gfx950).
//
// Rule restated from rasterizer/cuda/csrc/backward.cu:133-303 (3 channels) and
// :23-131 (N channels): every pixel re-walks its tile's sorted splat list
// back-to-front starting at its own final_idx, recomputes
//   alpha = min(0.99, opac*exp(-sigma))      (0.99 here, 0.999 in the forward)
//   ra = 1/(1-alpha); T *= ra; fac = alpha*T
//   v_rgb   += fac * v_out
//   v_alpha  = sum_c (rgb_c*T - buffer_c*ra) v_out_c
//              + T_final*ra*v_out_alpha - T_final*ra*sum_c bg_c v_out_c
//   buffer  += rgb*fac
//   v_sigma  = -opac*vis*v_alpha,  v_opacity += vis*v_alpha
//   v_conic += (.5 v_sigma dx^2, v_sigma dx dy, .5 v_sigma dy^2)
//   v_xy    += v_sigma*(a dx + b dy, b dx + c dy)
// and the per-pixel contributions are summed per Gaussian.
//
// tile16 mapping (block_width 16, 3 channels): one wave64 per tile, 4 pixels per
// lane (one in each 8x8 sub-tile), splats staged 64 at a time in LDS after an
// exact reach test (raster_common.h).  The reference reduces every splat across
// a 32-lane warp (9 values x 5 shuffle steps) and issues 9 atomics per warp per
// splat (72 per tile-splat).  Here a lane first folds its pixels into six
// moments of w = vis*v_alpha (sum w, w dx, w dy, w dx^2, w dx dy, w dy^2) + the
// rgb sums, turns them into the 9 gradient components, and then G splats x 9
// components lane-partials (G = 4) are reduced together
// with a halving butterfly: v_permlane32_swap / v_permlane16_swap (new on
// gfx950) and DPP row ops, every step halving the number of live values, ~20
// instructions per splat instead of 54.  The butterfly ends with one fully
// reduced (splat, component) per lane, so one wave-wide global_atomic_add_f32
// retires 32 components: 9 atomics per tile-splat instead of 72.
#include <stdlib.h>

#include <algorithm>

#include "raster_common.h"

// (the A/B variants of this kernel that lost -- the cross-row sum on the matrix pipe / through LDS, more resident waves,
//  three selects instead of the validity fold -- live in tools/exp/bwd_variants.patch with their records)

namespace {
// measurement hook, see raster_fwd.hip
__device__ unsigned long long *g_bwd_staged = nullptr;
__device__ gsr::WaveTrace g_bwd_trace = {nullptr, 0u};
}  // namespace

namespace {

using namespace gsr;

#define DPP_QUAD_XOR1 0xB1         // quad_perm:[1,0,3,2]
#define DPP_QUAD_XOR2 0x4E         // quad_perm:[2,3,0,1]
#define DPP_ROW_HALF_MIRROR 0x141  // lane l <-> 7-l inside each 8 lanes
#define DPP_ROW_ROR8 0x128         // lane l <-> l^8 inside each row of 16

template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), CTRL, 0xf, 0xf, true));
}

// After the call, lanes 0-31 hold (lo+hi) of `a`, lanes 32-63 hold (lo+hi) of `b`.
__device__ __forceinline__ float fold32(float a, float b) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// Rows (16 lanes) 0,2 end with `a` summed over the row pair, rows 1,3 with `b`.
__device__ __forceinline__ float fold16(float a, float b) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// halving step inside a row: the lane keeps x if !bit else y, and receives the
// same quantity from its DPP partner (whose `bit` is the opposite)
template <int CTRL>
__device__ __forceinline__ float halve(float x, float y, bool bit) {
  const float keep = bit ? y : x;
  const float send = bit ? x : y;
  return keep + dpp<CTRL>(send);
}

// Wave-wide sums of G*9 lane-partials P[9*j + c] (splat j < G, component c < 9).
//  (G == 8, one butterfly per eight splats, was selectable until round 6: tools/exp/bwd_variants.patch)
//  G == 4: lane l ends with component comp_of_lane(l) of splat l>>4 in `main_v`
//          (duplicated in lanes l and l^2) and component 8 of splat l>>4 in
//          `extra_v` (all 16 lanes of the row).
template <int G>
struct Butterfly;

template <>
struct Butterfly<4> {
  static __device__ __forceinline__ int splat_of_lane(int lane) { return lane >> 4; }
  static __device__ __forceinline__ int comp_of_lane(int lane) {
    return ((lane & 8) ? 4 : 0) + ((lane & 4) ? 2 : 0) + ((lane & 1) ? 1 : 0);
  }
  static __device__ __forceinline__ bool owns_main(int lane) { return (lane & 2) == 0; }
  static __device__ __forceinline__ bool owns_extra(int lane) { return (lane & 15) == 0; }
  static __device__ __forceinline__ float swap01(float v) { return dpp<DPP_QUAD_XOR1>(v); }
  static __device__ __forceinline__ void run(float (&P)[36], int lane, float &main_v, float &extra_v) {
    float Q[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) Q[i] = fold32(P[i], P[i + 18]);  // lane bit 5 <-> splat bit 1
    float R[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = fold16(Q[i], Q[i + 9]);    // lane bit 4 <-> splat bit 0
    finish(R, lane, main_v, extra_v);
  }
  // R[c]: component c of splat lane>>4, already summed over the four rows -> the sums over the row's 16 lanes
  static __device__ __forceinline__ void finish(const float (&R)[9], int lane, float &main_v, float &extra_v) {
    const bool b3 = lane & 8, b2 = lane & 4, b0 = lane & 1;
    float S[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) S[i] = halve<DPP_ROW_ROR8>(R[i], R[i + 4], b3);
    float U[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) U[i] = halve<DPP_ROW_HALF_MIRROR>(S[i], S[i + 2], b2);
    const float v = halve<DPP_QUAD_XOR1>(U[0], U[1], b0);
    main_v = v + dpp<DPP_QUAD_XOR2>(v);
    float e = R[8];
    e += dpp<DPP_ROW_ROR8>(e);
    e += dpp<DPP_ROW_HALF_MIRROR>(e);
    e += dpp<DPP_QUAD_XOR1>(e);
    e += dpp<DPP_QUAD_XOR2>(e);
    extra_v = e;
  }
};

// G == 4 with a tenth component (RGBD): same lane roles for components 0-7, and
// components 8, 9 of splat l>>4 in `extra_v`, `extra2_v` (all 16 lanes of the row).
struct Butterfly4x10 {
  static __device__ __forceinline__ void run(float (&P)[40], int lane, float &main_v, float &extra_v,
                                             float &extra2_v) {
    float Q[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) Q[i] = fold32(P[i], P[i + 20]);  // lane bit 5 <-> splat bit 1
    float R[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) R[i] = fold16(Q[i], Q[i + 10]);  // lane bit 4 <-> splat bit 0
    const bool b3 = lane & 8, b2 = lane & 4, b0 = lane & 1;
    float S[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) S[i] = halve<DPP_ROW_ROR8>(R[i], R[i + 4], b3);
    float U[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) U[i] = halve<DPP_ROW_HALF_MIRROR>(S[i], S[i + 2], b2);
    const float v = halve<DPP_QUAD_XOR1>(U[0], U[1], b0);
    main_v = v + dpp<DPP_QUAD_XOR2>(v);
    float e = R[8], f = R[9];
    e += dpp<DPP_ROW_ROR8>(e);
    f += dpp<DPP_ROW_ROR8>(f);
    e += dpp<DPP_ROW_HALF_MIRROR>(e);
    f += dpp<DPP_ROW_HALF_MIRROR>(f);
    e += dpp<DPP_QUAD_XOR1>(e);
    f += dpp<DPP_QUAD_XOR1>(f);
    e += dpp<DPP_QUAD_XOR2>(e);
    f += dpp<DPP_QUAD_XOR2>(f);
    extra_v = e;
    extra2_v = f;
  }
};

__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

// RGBD (G == 4 only): a fourth colour channel (one scalar per Gaussian, composited into its
// own image over background bg_extra by the forward) with cotangent v_out_extra [H,W];
// its gradient goes to v_extra [N].
constexpr int kPartialStride = 12;  // floats per list entry in deterministic mode (10 used)

// ---- deterministic mode, second pass -------------------------------------------------
// One lane per Gaussian (depth position i, Gaussian order[i]): sums the partial rows of its
// list entries in stream order -- entry e of band b, e in [cum[b n + i - 1], cum[b n + i]),
// lives in row slot_of[e] -- so the result does not depend on which tile's wave finished
// first.  Rows a tile never reached (it stopped early / the splat was culled at staging)
// carry flag 0.
template <bool RGBD>
__global__ __launch_bounds__(256) void reduce_partials_kernel(
    const int n, const int num_bands, const int capacity, const int *__restrict__ order,
    const int *__restrict__ cum, const int *__restrict__ slot_of, const float *__restrict__ partials,
    const unsigned char *__restrict__ pflags, float *__restrict__ v_xy, float *__restrict__ v_conic,
    float *__restrict__ v_colors, float *__restrict__ v_opacity, float *__restrict__ v_extra) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float acc[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int b = 0; b < num_bands; ++b) {
    const long long ci = (long long)b * n + i;
    int e0 = ci > 0 ? cum[ci - 1] : 0, e1 = cum[ci];
    e0 = min(e0, capacity);
    e1 = min(e1, capacity);
    for (int e = e0; e < e1; ++e) {
      const int s = slot_of[e];
      if (!pflags[s]) continue;
      const float4 *row = reinterpret_cast<const float4 *>(partials + (size_t)s * kPartialStride);
      const float4 r0 = row[0], r1 = row[1], r2 = row[2];
      acc[0] += r0.x, acc[1] += r0.y, acc[2] += r0.z, acc[3] += r0.w;
      acc[4] += r1.x, acc[5] += r1.y, acc[6] += r1.z, acc[7] += r1.w;
      acc[8] += r2.x;
      if (RGBD) acc[9] += r2.y;
    }
  }
  const int g = order[i];
  v_xy[2 * g] = acc[0], v_xy[2 * g + 1] = acc[1];
  v_conic[3 * g] = acc[2], v_conic[3 * g + 1] = acc[3], v_conic[3 * g + 2] = acc[4];
  v_colors[3 * g] = acc[5], v_colors[3 * g + 1] = acc[6], v_colors[3 * g + 2] = acc[7];
  v_opacity[g] = acc[8];
  if (RGBD) v_extra[g] = acc[9];
}

// ---- depth segments -------------------------------------------------------------------
// A deep tile's walk is a serial chain; on a grid that cannot fill the chip (480 x 270 is 510 tiles) the kernel lasts as
// long as its deepest tile.  The backward's per-pixel state is two scalars (T, K) and a run of list entries maps it
// affinely:  T_out = T_in R,  K_out = K_in - T_in S  with R the product of the run's `ra` and S the sum of
// alpha rho (rgb . v_out), rho the running product inside the run.  So a deep tile's list is cut into `seg_count`
// segments: a PRE-PASS (raster_bwd_segstate_kernel) computes (R, S) per pixel for segments 1 .. seg_count - 1 in
// parallel, and the main kernel runs one wave per (tile, sub-tile, segment): segment k applies the maps of the
// segments behind it to (T_final, K_0) and walks only its own entries.  Same per-entry arithmetic; T reaches a
// segment as a product of segment products instead of one chain: equal to the single walk to rounding, not bitwise.

template <int G, bool RGBD, bool SEG = false>
__global__ __launch_bounds__(64) void raster_bwd_tile16_kernel(
    const int tiles_x, const int num_tiles, const int img_w, const int img_h,
    const int *__restrict__ ids_sorted, const int2 *__restrict__ tile_bins,
    const float2 *__restrict__ xys, const float *__restrict__ conics,
    const float *__restrict__ colors, const float *__restrict__ opacities,
    const float *__restrict__ background, const float *__restrict__ final_Ts,
    const int *__restrict__ final_idx, const float *__restrict__ v_output,
    const float *__restrict__ v_output_alpha, float *__restrict__ v_xy,
    float *__restrict__ v_conic, float *__restrict__ v_colors, float *__restrict__ v_opacity,
    const float *__restrict__ extra, const float bg_extra, const float *__restrict__ v_out_extra,
    float *__restrict__ v_extra, const int deep_threshold, const unsigned base_grid,
    float *__restrict__ partials, unsigned char *__restrict__ pflags, const int2 *__restrict__ tile_bins2,
    const int idx_base2, const int seg_count = 1, const int seg_min = 0,
    const float2 *__restrict__ seg_state = nullptr) {
  // tile_bins2 (two-round lists, gsr_rasterize_forward_round): a tile's list is its range in tile_bins followed by
  // its range in tile_bins2 (relative to idx_base2 in ids_sorted): walked back to front, second segment first.
  static_assert(!RGBD || G == 4, "the 10-component butterfly exists for groups of 4");
  constexpr int NC = RGBD ? 10 : 9;
  __shared__ SplatA sA[kChunk];
  __shared__ SplatB sB[kChunk];
  __shared__ SplatC sC[kChunk];
  __shared__ int sId[kChunk];
  using BF = Butterfly<G>;

  int2 range = make_int2(0, 0);
  unsigned blk = blockIdx.x;
  int seg_k = 0;
  if constexpr (SEG) {  // block = segment * (4 base_grid) + the block of the unsegmented launch
    seg_k = (int)(blk / (4u * base_grid));
    blk -= (unsigned)seg_k * (4u * base_grid);
  }
  const WaveTrace trace = g_bwd_trace;
  const unsigned long long trace_t0 = trace_begin(trace);
  const unsigned long long stats_t0 = gsr_deep_ordered(deep_threshold) ? wall_clock64() : 0ull;
  const TileJob job = tile_job(blk, base_grid, tiles_x, num_tiles / tiles_x, tile_bins, deep_threshold, range);
  const int tile = job.tile, allowed = job.allowed;  // allowed: the sub-tiles this wave owns (raster_common.h)
  if (tile < 0) return;
  const int trace_len = range.y - range.x;
  int seg_behind = 0;  // segments behind this one whose maps are applied first
  if constexpr (SEG) {
    const int len = range.y - range.x;
    if (allowed == 15 || len <= seg_min) {  // not a deep tile: one walk, by segment 0's block
      if (seg_k > 0) return;
    } else {
      const int sl = seg_len_of(len, seg_count), first = range.x;
      seg_behind = min(seg_count, (len + sl - 1) / sl) - 1 - seg_k;
      if (seg_behind < 0) return;
      range.x = first + seg_k * sl;
      range.y = min(range.x + sl, range.y);
    }
  }
  int2 range2 = make_int2(0, 0);
  if (tile_bins2) {
    range2 = tile_bins2[tile];
    range2.x += idx_base2;
    range2.y += idx_base2;
  }
  if (range.y <= range.x && range2.y <= range2.x) return;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int lane = threadIdx.x;
  const int qx = tx * 16 + (lane & 7), qy = ty * 16 + (lane >> 3);
  const float fx0 = (float)qx, fx1 = (float)(qx + 8);
  const float fy0 = (float)qy, fy1 = (float)(qy + 8);
  const float tx0 = (float)(tx * 16), ty0 = (float)(ty * 16);
  const float bg0 = background[0], bg1 = background[1], bg2 = background[2];

  float T[4], K[4], vr[4], vg[4], vb[4], ve[4];
  int binf[4];
  // The per-pixel inputs of all four pixels are requested together, from a clamped address and selected afterwards
  // (a load under `inside` / `drawn` is a branch with its own wait: eight dependent round trips per wave before the
  // first splat).  What is selected away is never used: see `drawn` below.
  float in_T[4], in_r[4], in_g[4], in_b[4], in_a[4], in_e[4];
  int in_idx[4];
  bool in_img[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int col = qx + 8 * (p & 1), row = qy + 8 * (p >> 1);
    in_img[p] = col < img_w && row < img_h && ((allowed >> p) & 1);
    const size_t pid = in_img[p] ? (size_t)row * img_w + col : 0;
    in_T[p] = final_Ts[pid];
    in_r[p] = v_output[3 * pid];
    in_g[p] = v_output[3 * pid + 1];
    in_b[p] = v_output[3 * pid + 2];
    in_a[p] = v_output_alpha ? v_output_alpha[pid] : 0.f;
    in_e[p] = 0.f;
    if constexpr (RGBD) in_e[p] = v_out_extra[pid];
    in_idx[p] = final_idx[pid];
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const bool inside = in_img[p];
    T[p] = 1.f;
    K[p] = vr[p] = vg[p] = vb[p] = ve[p] = 0.f;
    binf[p] = -1;  // `inside && idx <= bin_final` folds into one compare
    {
      const float Tf = inside ? in_T[p] : 1.f;
      T[p] = Tf;
      // T_final == 1 exactly <=> nothing was composited at this pixel (a drawn splat has alpha >= 1/255):
      // no splat is `valid` there, and its cotangent must not be USED either -- the models' depth image is
      // `where(alpha > 0, depth / alpha, max)` (vanilla_gs.py:855, depth_gs.py:356), whose backward hands
      // 0/0 = NaN to exactly these pixels.  The reference's kernel branches on `valid` and never touches
      // them (backward.cu:133-303); the flat selects below would turn 0 * NaN into NaN sums.
      const bool drawn = Tf < 1.f;  // (false outside the image)
      vr[p] = drawn ? in_r[p] : 0.f;
      vg[p] = drawn ? in_g[p] : 0.f;
      vb[p] = drawn ? in_b[p] : 0.f;
      // T_final*ra*v_out_alpha - T_final*ra*(bg . v_out) = ra * K
      if constexpr (RGBD) ve[p] = drawn ? in_e[p] : 0.f;
      K[p] = !drawn ? 0.f
                    : Tf * ((v_output_alpha ? in_a[p] : 0.f) -
                            (bg0 * vr[p] + bg1 * vg[p] + bg2 * vb[p] + (RGBD ? bg_extra * ve[p] : 0.f)));
      binf[p] = drawn ? in_idx[p] : -1;
    }
  }
  if constexpr (SEG) {  // the state behind this run: ONE composed map per pixel (raster_bwd_segprefix_kernel)
    if (seg_behind > 0) {
      const size_t pixels = (size_t)img_w * img_h;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int col = qx + 8 * (p & 1), row = qy + 8 * (p >> 1);
        if (in_img[p]) {
          const float2 st = seg_state[(size_t)seg_k * pixels + (size_t)row * img_w + col];
          K[p] -= T[p] * st.y;
          T[p] *= st.x;
        }
      }
    }
  }
  // last sorted index any pixel of sub-tile p still needs (wave-uniform)
  int topp[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) topp[p] = __builtin_amdgcn_readfirstlane(wave_max(binf[p]));
  const int max_top = max(max(topp[0], topp[1]), max(topp[2], topp[3]));

  // lane-constant destination of the `main` value
  const int comp = BF::comp_of_lane(lane);
  float *const dst_base = comp < 2 ? v_xy : (comp < 5 ? v_conic : (comp < 8 ? v_colors : v_opacity));
  const int dst_stride = comp < 2 ? 2 : (comp < 8 ? 3 : 1);
  const int dst_off = comp < 2 ? comp : (comp < 5 ? comp - 2 : (comp < 8 ? comp - 5 : 0));
  const bool owns_main = BF::owns_main(lane);
  const bool owns_extra = BF::owns_extra(lane);
  // lane-constant selectors of the moment -> gradient map for this lane's component
  const float sel_ha = comp == 0 ? 1.f : 0.f, sel_hc = comp == 1 ? 1.f : 0.f, sel_b = comp < 2 ? 1.f : 0.f;
  const float sel_no = (comp == 2 || comp == 4) ? 0.5f : (comp == 3 ? 1.f : 0.f);
  const float sel_one = comp >= 5 ? 1.f : 0.f;

  unsigned long long *const staged = g_bwd_staged;
  for (int seg = tile_bins2 ? 1 : 0; seg >= 0; --seg) {
  if (seg) range = range2; else if (tile_bins2) range = tile_bins[tile];  // (tile_job left the first segment in `range`)
  const int top = min(range.y - 1, max_top);
#if GSR_STAGE_AHEAD
  // (raster_fwd.hip: the next chunk's loads in flight during the walk over the current one)
  // (nothing to walk loads nothing: lanes outside a NON-empty range read Gaussian 0, which then exists)
  int g_cur = 0, g_next = 0;
  StageRegs regs = {};
  if (top >= range.x) {
    g_cur = stage_load_id(top - lane >= range.x, top - lane, ids_sorted);
    g_next = stage_load_id(top - kChunk - lane >= range.x, top - kChunk - lane, ids_sorted);
    regs = stage_load_attrs(g_cur, xys, conics, colors, opacities, RGBD ? extra : nullptr);
  }
#endif
  for (int hi = top; hi >= range.x; hi -= kChunk) {
    // back to front: lane l fetches sorted index hi - l; kept splats stay in that order
    const int sidx_l = hi - lane;
#if GSR_STAGE_AHEAD
    const int count = stage_commit(lane, sidx_l >= range.x, sidx_l, g_cur, regs, tx0, ty0, sA, sB, sC, sId, staged, allowed);
    __syncthreads();
    g_cur = g_next;
    regs = stage_load_attrs(g_cur, xys, conics, colors, opacities, RGBD ? extra : nullptr);
    g_next = stage_load_id(sidx_l - 2 * kChunk >= range.x, sidx_l - 2 * kChunk, ids_sorted);
#else
    const int count = stage_chunk(lane, sidx_l >= range.x, sidx_l, tx0, ty0, ids_sorted, xys, conics,
                                  colors, opacities, sA, sB, sC, sId, RGBD ? extra : nullptr, staged, allowed);
    __syncthreads();
#endif

    for (int t0 = 0; t0 < count; t0 += G) {
      float P[NC * G];
      bool lane_any = false;
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int t = t0 + j;
        float m0 = 0.f, mx = 0.f, my = 0.f, mxx = 0.f, mxy = 0.f, myy = 0.f;
        float sr = 0.f, sg = 0.f, sb = 0.f, se = 0.f;
        if (t < count) {  // wave-uniform
          const SplatA A = sA[t];
          const SplatB B = sB[t];
          const SplatC C = sC[t];
          const float dx0 = A.x - fx0, dx1 = A.x - fx1;
          const float dy0 = A.y - fy0, dy1 = A.y - fy1;
          const float ax0 = A.ha * dx0 * dx0, ax1 = A.ha * dx1 * dx1;
          const float cy0 = B.hc * dy0 * dy0, cy1 = B.hc * dy1 * dy1;
          const float bx0 = A.b * dx0, bx1 = A.b * dx1;
          const float sig[4] = {(ax0 + cy0) + bx0 * dy0, (ax1 + cy0) + bx1 * dy0,
                                (ax0 + cy1) + bx0 * dy1, (ax1 + cy1) + bx1 * dy1};
          const float dxs[4] = {dx0, dx1, dx0, dx1};
          const float dys[4] = {dy0, dy0, dy1, dy1};
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            // wave-uniform: sub-tile out of the splat's reach, or every pixel of
            // it finished in front of this splat
            if (!((C.mask >> p) & 1) || C.sidx > topp[p]) continue;
            const float sigma = sig[p];
            const float vis0 = __expf(-sigma);
            const float alpha0 = fminf(GSR_ALPHA_MAX_BWD, B.opac * vis0);
            const bool valid = (C.sidx <= binf[p]) && !(sigma < 0.f || alpha0 < GSR_ALPHA_MIN);
            // An invalid (pixel, splat) pair runs the same arithmetic on alpha = vis = 0: ra = 1 / (1 - 0) = 1, T * 1 = T
            // and alpha * T = 0 EXACTLY, so T, K and every sum come out as with the three selects this replaces (two
            // v_cndmask, 4.2 issue cycles each on gfx950, instead of three: DESIGN.md section 4.17).  v_alpha stays finite
            // (T, K and the cotangents are: NaN cotangents of undrawn pixels were zeroed at the load), so 0 * v_alpha = 0.
            const float alpha = valid ? alpha0 : 0.f;
            const float vis = valid ? vis0 : 0.f;
            const float ra = __builtin_amdgcn_rcpf(1.f - alpha);
            const float Tn = T[p] * ra;
            // sum_c (rgb_c*T - buffer_c*ra) v_out_c + T_final*ra*(v_out_alpha - bg.v_out)
            //   = T * (rgb . v_out) + ra * (K - buffer . v_out);
            // K[p] carries  T_final*(v_out_alpha - bg.v_out) - buffer.v_out  (a scalar per
            // pixel instead of the reference's 3-channel running buffer)
            float d = B.r * vr[p] + B.g * vg[p] + C.blue * vb[p];
            if constexpr (RGBD) d += C.extra * ve[p];  // (a literal "+ 0.f" in the 3-channel case is a real v_add: -0 semantics)
            const float v_alpha = Tn * d + ra * K[p];
            const float w = vis * v_alpha;
            const float fac = alpha * Tn;
            T[p] = Tn;
            K[p] -= fac * d;
            sr += fac * vr[p];
            sg += fac * vg[p];
            sb += fac * vb[p];
            if constexpr (RGBD) se += fac * ve[p];
            const float wx = w * dxs[p], wy = w * dys[p];
            m0 += w;
            mx += wx;
            my += wy;
            mxx += wx * dxs[p];
            mxy += wx * dys[p];
            myy += wy * dys[p];
            lane_any = lane_any || valid;
          }
        }
        // raw moments; they are turned into gradient components AFTER the wave-wide
        // reduction (linear map: once per group by the owning lanes instead of once
        // per splat by all 64)
        P[NC * j + 0] = mx;
        P[NC * j + 1] = my;
        P[NC * j + 2] = mxx;
        P[NC * j + 3] = mxy;
        P[NC * j + 4] = myy;
        P[NC * j + 5] = sr;
        P[NC * j + 6] = sg;
        P[NC * j + 7] = sb;
        P[NC * j + 8] = m0;
        if constexpr (RGBD) P[NC * j + 9] = se;
      }
      if (!__any(lane_any)) continue;  // nothing in this group touched a live pixel

      float main_v, extra_v, extra2_v = 0.f;
      if constexpr (RGBD) {
        Butterfly4x10::run(P, lane, main_v, extra_v, extra2_v);
      } else {
        BF::run(P, lane, main_v, extra_v);
      }
      // v_sigma = -opac * w:  v_xy = -opac (a Sx + b Sy, b Sx + c Sy),
      // v_conic = -opac (Sxx/2, Sxy, Syy/2),  v_rgb = the colour sums,  v_opacity = S0
      const float other = BF::swap01(main_v);
      const int t = t0 + BF::splat_of_lane(lane);
      if (t < count) {
        const SplatA A = sA[t];
        const SplatB B = sB[t];
        const float no = -B.opac;
        const float k1 = no * (2.f * (sel_ha * A.ha + sel_hc * B.hc) + sel_no) + sel_one;
        const float k2 = no * sel_b * A.b;
        const float grad = main_v * k1 + other * k2;
        const int g = sId[t];
        if (partials) {
          // deterministic mode: the (tile, splat) partial goes to the row of its list entry;
          // rows are summed per Gaussian in a fixed order by reduce_partials_kernel
          const int sx = sC[t].sidx;
          float *row = partials + (size_t)sx * kPartialStride;
          if (owns_main) row[comp] = grad;
          if (owns_extra) {
            row[8] = extra_v;
            pflags[sx] = 1;
          }
          if constexpr (RGBD) {
            if ((lane & 15) == 1) row[9] = extra2_v;
          }
        } else {
        if (owns_main && grad != 0.f)
          unsafeAtomicAdd(dst_base + (size_t)g * dst_stride + dst_off, grad);
        if (owns_extra && extra_v != 0.f) unsafeAtomicAdd(v_opacity + g, extra_v);
        if constexpr (RGBD) {
          if ((lane & 15) == 1 && extra2_v != 0.f) unsafeAtomicAdd(v_extra + g, extra2_v);
        }
        }
      }
    }
    __syncthreads();
  }
  }
  trace_end(trace, trace_t0, tile, allowed, trace_len);
  job_stats_end(job, 1, stats_t0, trace_len);
}

// The pre-pass of the depth segments: (R, S) of segment seg_k = 1 + block / (4 base_grid) for every pixel of a deep
// tile's sub-tile -> seg_state[(seg_k - 1) pixels + pixel].  Same staging, same sigma / alpha / validity expressions as
// the walk above; no gradients.
template <bool RGBD>
__global__ __launch_bounds__(64) void raster_bwd_segstate_kernel(
    const int tiles_x, const int num_tiles, const int img_w, const int img_h,
    const int *__restrict__ ids_sorted, const int2 *__restrict__ tile_bins,
    const float2 *__restrict__ xys, const float *__restrict__ conics,
    const float *__restrict__ colors, const float *__restrict__ opacities,
    const float *__restrict__ final_Ts, const int *__restrict__ final_idx, const float *__restrict__ v_output,
    const float *__restrict__ extra, const float *__restrict__ v_out_extra,
    const int deep_threshold, const unsigned base_grid, const int seg_count, const int seg_min,
    float2 *__restrict__ seg_state) {
  __shared__ SplatA sA[kChunk];
  __shared__ SplatB sB[kChunk];
  __shared__ SplatC sC[kChunk];
  int2 range = make_int2(0, 0);
  unsigned blk = blockIdx.x;
  const int seg_k = 1 + (int)(blk / (4u * base_grid));
  blk -= (unsigned)(seg_k - 1) * (4u * base_grid);
  const TileJob job = tile_job(blk, base_grid, tiles_x, num_tiles / tiles_x, tile_bins, deep_threshold, range);
  const int tile = job.tile, allowed = job.allowed;
  if (tile < 0) return;
  const int len = range.y - range.x;
  if (allowed == 15 || len <= seg_min) return;
  const int sl = seg_len_of(len, seg_count);
  if (seg_k >= min(seg_count, (len + sl - 1) / sl)) return;
  range.x += seg_k * sl;
  range.y = min(range.x + sl, range.y);
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int lane = threadIdx.x;
  const int qx = tx * 16 + (lane & 7), qy = ty * 16 + (lane >> 3);
  const float fx0 = (float)qx, fx1 = (float)(qx + 8);
  const float fy0 = (float)qy, fy1 = (float)(qy + 8);
  const float tx0 = (float)(tx * 16), ty0 = (float)(ty * 16);

  float rho[4], S[4], vr[4], vg[4], vb[4], ve[4];
  int binf[4];
  bool in_img[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int col = qx + 8 * (p & 1), row = qy + 8 * (p >> 1);
    in_img[p] = col < img_w && row < img_h && ((allowed >> p) & 1);
    const size_t pid = in_img[p] ? (size_t)row * img_w + col : 0;
    const float Tf = in_img[p] ? final_Ts[pid] : 1.f;
    const bool drawn = Tf < 1.f;
    const float r = v_output[3 * pid], g = v_output[3 * pid + 1], b = v_output[3 * pid + 2];
    const int fi = final_idx[pid];
    vr[p] = drawn ? r : 0.f;
    vg[p] = drawn ? g : 0.f;
    vb[p] = drawn ? b : 0.f;
    ve[p] = 0.f;
    if constexpr (RGBD) {
      const float e = v_out_extra[pid];
      ve[p] = drawn ? e : 0.f;
    }
    binf[p] = drawn ? fi : -1;
    rho[p] = 1.f;
    S[p] = 0.f;
  }
  int topp[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) topp[p] = __builtin_amdgcn_readfirstlane(wave_max(binf[p]));
  const int max_top = max(max(topp[0], topp[1]), max(topp[2], topp[3]));
  const int top = min(range.y - 1, max_top);
  for (int hi = top; hi >= range.x; hi -= kChunk) {
    const int sidx_l = hi - lane;
#if GSR_STAGE_AHEAD
    const int count = stage_chunk_flat(lane, sidx_l >= range.x, sidx_l, tx0, ty0, ids_sorted, xys, conics, colors,
                                       opacities, sA, sB, sC, RGBD ? extra : nullptr, allowed);
#else
    const int count = stage_chunk(lane, sidx_l >= range.x, sidx_l, tx0, ty0, ids_sorted, xys, conics, colors, opacities,
                                  sA, sB, sC, nullptr, RGBD ? extra : nullptr, nullptr, allowed);
#endif
    __syncthreads();
    for (int t = 0; t < count; ++t) {
      const SplatA A = sA[t];
      const SplatB B = sB[t];
      const SplatC C = sC[t];
      const float dx0 = A.x - fx0, dx1 = A.x - fx1;
      const float dy0 = A.y - fy0, dy1 = A.y - fy1;
      const float ax0 = A.ha * dx0 * dx0, ax1 = A.ha * dx1 * dx1;
      const float cy0 = B.hc * dy0 * dy0, cy1 = B.hc * dy1 * dy1;
      const float bx0 = A.b * dx0, bx1 = A.b * dx1;
      const float sig[4] = {(ax0 + cy0) + bx0 * dy0, (ax1 + cy0) + bx1 * dy0,
                            (ax0 + cy1) + bx0 * dy1, (ax1 + cy1) + bx1 * dy1};
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if (!((C.mask >> p) & 1) || C.sidx > topp[p]) continue;
        const float sigma = sig[p];
        const float vis = __expf(-sigma);
        const float alpha = fminf(GSR_ALPHA_MAX_BWD, B.opac * vis);
        const bool valid = (C.sidx <= binf[p]) && !(sigma < 0.f || alpha < GSR_ALPHA_MIN);
        const float ra = __builtin_amdgcn_rcpf(1.f - alpha);
        const float rn = rho[p] * ra;
        float d = B.r * vr[p] + B.g * vg[p] + C.blue * vb[p];
        if constexpr (RGBD) d += C.extra * ve[p];
        const float fac = valid ? alpha * rn : 0.f;
        rho[p] = valid ? rn : rho[p];
        S[p] += fac * d;
      }
    }
    __syncthreads();
  }
  const size_t pixels = (size_t)img_w * img_h;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int col = qx + 8 * (p & 1), row = qy + 8 * (p >> 1);
    if (in_img[p]) seg_state[(size_t)(seg_k - 1) * pixels + (size_t)row * img_w + col] = make_float2(rho[p], S[p]);
  }
}

// Between the pre-pass and the runs: per pixel of a split tile, the runs' maps (R_j, S_j), j = 1 .. n - 1 (stored at
// j - 1) become, in place, the composed maps run k starts from, k = 0 .. n - 2 (stored at k): everything behind run k
// applied in walk order, farthest run first -- (A, B) with T = T_final A, K = K_0 - T_final B; composing run j onto
// (A, B) gives (A R_j, B + A S_j).  A wave of run k then loads one map per pixel instead of n - 1 - k.
__global__ __launch_bounds__(256) void raster_bwd_segprefix_kernel(
    const int tiles_x, const int img_w, const int img_h, const int2 *__restrict__ tile_bins, const int deep_threshold,
    const int seg_count, const int seg_min, float2 *__restrict__ seg_state) {
  const size_t pixels = (size_t)img_w * img_h;
  const size_t pid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pid >= pixels) return;
  const int row = (int)(pid / img_w), col = (int)(pid - (size_t)row * img_w);
  const int2 range = tile_bins[(row >> 4) * tiles_x + (col >> 4)];
  const int len = range.y - range.x;
  if (!(deep_threshold > 0 && len > deep_threshold) || len <= seg_min) return;
  const int sl = seg_len_of(len, seg_count);
  const int nseg = min(seg_count, (len + sl - 1) / sl);
  float A = 1.f, B = 0.f;
  for (int k = nseg - 2; k >= 0; --k) {  // run k + 1's map sits at k
    float2 *q = seg_state + (size_t)k * pixels + pid;
    const float2 rs = *q;
    B += A * rs.y;
    A *= rs.x;
    *q = make_float2(A, B);
  }
}

// ----------------------------------------------------------------- generic
// One lane per pixel; per-batch gradient accumulators live in LDS (ds_add_f32),
// one global atomic per (tile, splat, component) at the end of the batch.
template <int CMAX>
__global__ __launch_bounds__(256) void raster_bwd_generic_kernel(
    const int tiles_x, const int img_w, const int img_h, const int channels, const int cstride,
    const int with_alpha,
    const int *__restrict__ ids_sorted, const int2 *__restrict__ tile_bins,
    const float2 *__restrict__ xys, const float *__restrict__ conics,
    const float *__restrict__ colors, const float *__restrict__ opacities,
    const float *__restrict__ background, const float *__restrict__ final_Ts,
    const int *__restrict__ final_idx, const float *__restrict__ v_output,
    const float *__restrict__ v_output_alpha, float *__restrict__ v_xy,
    float *__restrict__ v_conic, float *__restrict__ v_colors, float *__restrict__ v_opacity) {
  extern __shared__ float s_dyn[];  // [bsize][6 + channels] accumulators
  __shared__ int s_id[256];
  __shared__ float s_x[256], s_y[256], s_o[256], s_a[256], s_b[256], s_c[256];
  __shared__ int s_top;

  const int bsize = blockDim.x * blockDim.y;
  const int tr = threadIdx.y * blockDim.x + threadIdx.x;
  const int tile = blockIdx.y * tiles_x + blockIdx.x;
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = blockIdx.y * blockDim.y + threadIdx.y;
  const float px = (float)col, py = (float)row;
  const bool inside = col < img_w && row < img_h;
  const int stride = 6 + channels;

  const int2 range = tile_bins[tile];
  float vout[CMAX], buf[CMAX];
#pragma unroll
  for (int c = 0; c < CMAX; ++c) vout[c] = buf[c] = 0.f;
  float T_final = 1.f, vout_alpha = 0.f, bgdot = 0.f;
  int bin_final = -1;
  if (inside) {
    const size_t pid = (size_t)row * img_w + col;
    T_final = final_Ts[pid];
    vout_alpha = (with_alpha && v_output_alpha) ? v_output_alpha[pid] : 0.f;
    bin_final = final_idx[pid];
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < channels) {
        vout[c] = v_output[pid * cstride + c];
        bgdot += background[c] * vout[c];
      }
  }
  float T = T_final;

  if (tr == 0) s_top = -1;
  __syncthreads();
  atomicMax(&s_top, bin_final);
  __syncthreads();
  const int top = min(range.y - 1, s_top);

  for (int hi = top; hi >= range.x; hi -= bsize) {
    __syncthreads();  // previous batch fully flushed
    const int idx = hi - tr;
    if (idx >= range.x) {
      const int g = ids_sorted[idx];
      s_id[tr] = g;
      const float2 xy = xys[g];
      s_x[tr] = xy.x;
      s_y[tr] = xy.y;
      s_o[tr] = opacities[g];
      s_a[tr] = conics[3 * g];
      s_b[tr] = conics[3 * g + 1];
      s_c[tr] = conics[3 * g + 2];
    }
    for (int k = 0; k < stride; ++k) s_dyn[tr * stride + k] = 0.f;
    __syncthreads();
    const int count = min(bsize, hi - range.x + 1);
    for (int t = 0; t < count; ++t) {
      if (hi - t > bin_final) continue;
      const float dx = s_x[t] - px, dy = s_y[t] - py;
      const float a = s_a[t], b = s_b[t], c = s_c[t], opac = s_o[t];
      const float sigma = 0.5f * (a * dx * dx + c * dy * dy) + b * dx * dy;
      const float vis = __expf(-sigma);
      const float alpha = fminf(GSR_ALPHA_MAX_BWD, opac * vis);
      if (sigma < 0.f || alpha < GSR_ALPHA_MIN) continue;
      const float ra = 1.f / (1.f - alpha);
      T *= ra;
      const float fac = alpha * T;
      float v_alpha = 0.f;
      float *acc = s_dyn + t * stride;
      const float *rgb = colors + (size_t)s_id[t] * cstride;
#pragma unroll
      for (int ch = 0; ch < CMAX; ++ch)
        if (ch < channels) {
          const float cval = rgb[ch];
          atomicAdd(acc + 6 + ch, fac * vout[ch]);
          v_alpha += (cval * T - buf[ch] * ra) * vout[ch];
          buf[ch] += cval * fac;
        }
      v_alpha += T_final * ra * vout_alpha;
      v_alpha += -T_final * ra * bgdot;
      const float v_sigma = -opac * vis * v_alpha;
      atomicAdd(acc + 0, v_sigma * (a * dx + b * dy));
      atomicAdd(acc + 1, v_sigma * (b * dx + c * dy));
      atomicAdd(acc + 2, 0.5f * v_sigma * dx * dx);
      atomicAdd(acc + 3, v_sigma * dx * dy);
      atomicAdd(acc + 4, 0.5f * v_sigma * dy * dy);
      atomicAdd(acc + 5, vis * v_alpha);
    }
    __syncthreads();
    if (tr < count) {
      const int g = s_id[tr];
      const float *acc = s_dyn + tr * stride;
      if (acc[0] != 0.f) unsafeAtomicAdd(v_xy + 2 * (size_t)g, acc[0]);
      if (acc[1] != 0.f) unsafeAtomicAdd(v_xy + 2 * (size_t)g + 1, acc[1]);
      if (acc[2] != 0.f) unsafeAtomicAdd(v_conic + 3 * (size_t)g, acc[2]);
      if (acc[3] != 0.f) unsafeAtomicAdd(v_conic + 3 * (size_t)g + 1, acc[3]);
      if (acc[4] != 0.f) unsafeAtomicAdd(v_conic + 3 * (size_t)g + 2, acc[4]);
      if (acc[5] != 0.f) unsafeAtomicAdd(v_opacity + g, acc[5]);
      for (int ch = 0; ch < channels; ++ch)
        if (acc[6 + ch] != 0.f) unsafeAtomicAdd(v_colors + (size_t)g * cstride + ch, acc[6 + ch]);
    }
  }
}

int launch_generic(unsigned img_h, unsigned img_w, unsigned bw, unsigned channels, const int32_t *ids,
                   const int32_t *bins, const float *xys, const float *conics, const float *colors,
                   const float *opac, const float *background, const float *final_Ts,
                   const int32_t *final_idx, const float *v_output, const float *v_output_alpha,
                   float *v_xy, float *v_conic, float *v_colors, float *v_opacity, hipStream_t s) {
  const int tiles_x = (int)gsr_cdiv(img_w, bw), tiles_y = (int)gsr_cdiv(img_h, bw);
  const dim3 grd(tiles_x, tiles_y), blk(bw, bw);
  // More than 32 channels: 32 per pass.  v_alpha is linear in the channels (sum over c of
  // (rgb_c T - buffer_c ra) v_out_c, minus T_final ra sum_c bg_c v_out_c), so every pass adds its
  // share of v_xy / v_conic / v_opacity; the v_output_alpha term goes with the first pass.
#define GSR_LAUNCH_BWD(CM, C0, CN)                                                                \
  hipLaunchKernelGGL(raster_bwd_generic_kernel<CM>, grd, blk,                                     \
                     (size_t)bw * bw * (6 + (CN)) * sizeof(float), s, tiles_x, (int)img_w,        \
                     (int)img_h, (int)(CN), (int)channels, (C0) == 0 ? 1 : 0, ids,                \
                     reinterpret_cast<const int2 *>(bins), reinterpret_cast<const float2 *>(xys), \
                     conics, colors + (C0), opac, background + (C0), final_Ts, final_idx,         \
                     v_output + (C0), v_output_alpha, v_xy, v_conic, v_colors + (C0), v_opacity)
  if (channels <= 4) GSR_LAUNCH_BWD(4, 0, channels);
  else if (channels <= 8) GSR_LAUNCH_BWD(8, 0, channels);
  else if (channels <= 16) GSR_LAUNCH_BWD(16, 0, channels);
  else
    for (unsigned c0 = 0; c0 < channels; c0 += 32) GSR_LAUNCH_BWD(32, c0, std::min(32u, channels - c0));
#undef GSR_LAUNCH_BWD
  GSR_CHECK_LAUNCH("rasterize_backward(generic)");
  return GSR_OK;
}

int zero_grads(int n, unsigned channels, float *v_xy, float *v_conic, float *v_colors,
               float *v_opacity, hipStream_t s) {
  // one fill when the four accumulators are slices of one allocation (the
  // Python binding lays them out back to back), four otherwise
  if (v_conic == v_xy + 2 * (size_t)n && v_colors == v_conic + 3 * (size_t)n &&
      v_opacity == v_colors + (size_t)channels * n) {
    if (int zrc = gsr_zero_async(v_xy, sizeof(float) * (6 + channels) * (size_t)n, s)) return zrc;
    return GSR_OK;
  }
  if (int zrc = gsr_zero_async(v_xy, sizeof(float) * 2 * (size_t)n, s)) return zrc;
  if (int zrc = gsr_zero_async(v_conic, sizeof(float) * 3 * (size_t)n, s)) return zrc;
  if (int zrc = gsr_zero_async(v_colors, sizeof(float) * channels * (size_t)n, s)) return zrc;
  if (int zrc = gsr_zero_async(v_opacity, sizeof(float) * (size_t)n, s)) return zrc;
  return GSR_OK;
}

}  // namespace

GSR_EXPORT int gsr_rasterize_backward_nd(
    unsigned img_height, unsigned img_width, unsigned block_width, unsigned channels,
    int num_points, const int32_t *gaussian_ids_sorted, const int32_t *tile_bins, const float *xys,
    const float *conics, const float *colors, const float *opacities, const float *background,
    const float *final_Ts, const int32_t *final_idx, const float *v_output,
    const float *v_output_alpha, float *v_xy, float *v_conic, float *v_colors, float *v_opacity,
    gsr_stream_t stream) {
  GSR_REQUIRE(block_width >= 2 && block_width <= 16, "rasterize_backward: block_width must be in [2,16]");
  GSR_REQUIRE(img_height > 0 && img_width > 0, "rasterize_backward: empty image");
  GSR_REQUIRE(channels >= 1 && channels <= GSR_MAX_CHANNELS, "rasterize_backward: channels must be in [1,%d]", GSR_MAX_CHANNELS);
  GSR_REQUIRE(num_points >= 0, "rasterize_backward: num_points < 0");
  if (num_points == 0) return GSR_OK;
  GSR_REQUIRE(gaussian_ids_sorted && tile_bins && xys && conics && colors && opacities &&
                  background && final_Ts && final_idx && v_output && v_xy &&
                  v_conic && v_colors && v_opacity,
              "rasterize_backward: null pointer");
  int rc = zero_grads(num_points, channels, v_xy, v_conic, v_colors, v_opacity, (hipStream_t)stream);
  if (rc != GSR_OK) return rc;
  return launch_generic(img_height, img_width, block_width, channels, gaussian_ids_sorted, tile_bins,
                        xys, conics, colors, opacities, background, final_Ts, final_idx, v_output,
                        v_output_alpha, v_xy, v_conic, v_colors, v_opacity, (hipStream_t)stream);
}

GSR_EXPORT int gsr_rasterize_backward(
    unsigned img_height, unsigned img_width, unsigned block_width, int num_points,
    const int32_t *gaussian_ids_sorted, const int32_t *tile_bins, const float *xys,
    const float *conics, const float *colors, const float *opacities, const float *background,
    const float *final_Ts, const int32_t *final_idx, const float *v_output,
    const float *v_output_alpha, float *v_xy, float *v_conic, float *v_colors, float *v_opacity,
    int deep_tile_threshold, gsr_stream_t stream) {
  return gsr_rasterize_backward_ex(img_height, img_width, block_width, num_points, gaussian_ids_sorted, tile_bins,
                                   xys, conics, colors, opacities, background, final_Ts, final_idx, v_output,
                                   v_output_alpha, v_xy, v_conic, v_colors, v_opacity, deep_tile_threshold, 0, stream);
}

GSR_EXPORT int gsr_rasterize_backward_ex(
    unsigned img_height, unsigned img_width, unsigned block_width, int num_points,
    const int32_t *gaussian_ids_sorted, const int32_t *tile_bins, const float *xys,
    const float *conics, const float *colors, const float *opacities, const float *background,
    const float *final_Ts, const int32_t *final_idx, const float *v_output,
    const float *v_output_alpha, float *v_xy, float *v_conic, float *v_colors, float *v_opacity,
    int deep_tile_threshold, int accumulators_zeroed, gsr_stream_t stream) {
  if (block_width != 16)
    return gsr_rasterize_backward_nd(img_height, img_width, block_width, 3, num_points,
                                     gaussian_ids_sorted, tile_bins, xys, conics, colors, opacities,
                                     background, final_Ts, final_idx, v_output, v_output_alpha, v_xy,
                                     v_conic, v_colors, v_opacity, stream);
  GSR_REQUIRE(img_height > 0 && img_width > 0, "rasterize_backward: empty image");
  GSR_REQUIRE(num_points >= 0, "rasterize_backward: num_points < 0");
  if (num_points == 0) return GSR_OK;
  GSR_REQUIRE(gaussian_ids_sorted && tile_bins && xys && conics && colors && opacities &&
                  background && final_Ts && final_idx && v_output && v_xy &&
                  v_conic && v_colors && v_opacity,
              "rasterize_backward: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (!accumulators_zeroed) {  // (else: cleared by gsr_rasterize_forward_ex's zero_ptr, untouched since)
    int rc = zero_grads(num_points, 3, v_xy, v_conic, v_colors, v_opacity, s);
    if (rc != GSR_OK) return rc;
  }
  const int tiles_x = (int)gsr_cdiv(img_width, 16), tiles_y = (int)gsr_cdiv(img_height, 16);
  const int num_tiles = tiles_x * tiles_y;
  const unsigned base = gsr_xcd_grid(tiles_x, num_tiles / tiles_x);
  const int deep = gsr_prepare_jobs(deep_tile_threshold, tiles_x, tiles_y, tile_bins, s);
  hipLaunchKernelGGL((raster_bwd_tile16_kernel<4, false>), dim3(deep ? 4 * base : base), dim3(64), 0, s, tiles_x,
                     num_tiles, (int)img_width, (int)img_height, gaussian_ids_sorted,
                     reinterpret_cast<const int2 *>(tile_bins), reinterpret_cast<const float2 *>(xys), conics, colors,
                     opacities, background, final_Ts, final_idx, v_output, v_output_alpha, v_xy, v_conic, v_colors,
                     v_opacity, (const float *)nullptr, 0.f, (const float *)nullptr, (float *)nullptr, deep, base,
                     (float *)nullptr, (unsigned char *)nullptr, (const int2 *)nullptr, 0);
  GSR_CHECK_LAUNCH("rasterize_backward(tile16)");
  return GSR_OK;
}

GSR_EXPORT int gsr_rasterize_backward_rgbd(
    unsigned img_height, unsigned img_width, int num_points, const int32_t *gaussian_ids_sorted,
    const int32_t *tile_bins, const float *xys, const float *conics, const float *colors, const float *extra,
    const float *opacities, const float *background, float extra_background, const float *final_Ts,
    const int32_t *final_idx, const float *v_output, const float *v_output_extra, const float *v_output_alpha,
    float *v_xy, float *v_conic, float *v_colors, float *v_extra, float *v_opacity, int deep_tile_threshold,
    int accumulators_zeroed, gsr_stream_t stream) {
  GSR_REQUIRE(img_height > 0 && img_width > 0, "rasterize_backward_rgbd: empty image");
  GSR_REQUIRE(num_points >= 0, "rasterize_backward_rgbd: num_points < 0");
  if (num_points == 0) return GSR_OK;
  GSR_REQUIRE(gaussian_ids_sorted && tile_bins && xys && conics && colors && extra && opacities && background &&
                  final_Ts && final_idx && v_output && v_output_extra && v_xy && v_conic && v_colors && v_extra &&
                  v_opacity,
              "rasterize_backward_rgbd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (!accumulators_zeroed) {  // (else: cleared by gsr_rasterize_forward_rgbd's zero_ptr)
    int rc = zero_grads(num_points, 3, v_xy, v_conic, v_colors, v_opacity, s);
    if (rc != GSR_OK) return rc;
    if (int zrc = gsr_zero_async(v_extra, sizeof(float) * (size_t)num_points, s)) return zrc;
  }
  const int tiles_x = (int)gsr_cdiv(img_width, 16), tiles_y = (int)gsr_cdiv(img_height, 16);
  const int num_tiles = tiles_x * tiles_y;
  const unsigned base = gsr_xcd_grid(tiles_x, num_tiles / tiles_x);
  const int deep = gsr_prepare_jobs(deep_tile_threshold, tiles_x, tiles_y, tile_bins, s);
  hipLaunchKernelGGL((raster_bwd_tile16_kernel<4, true>), dim3(deep ? 4 * base : base), dim3(64),
                     0, s, tiles_x, num_tiles, (int)img_width, (int)img_height, gaussian_ids_sorted,
                     reinterpret_cast<const int2 *>(tile_bins), reinterpret_cast<const float2 *>(xys), conics,
                     colors, opacities, background, final_Ts, final_idx, v_output, v_output_alpha, v_xy, v_conic,
                     v_colors, v_opacity, extra, extra_background, v_output_extra, v_extra, deep, base, (float *)nullptr,
                     (unsigned char *)nullptr, (const int2 *)nullptr, 0);
  GSR_CHECK_LAUNCH("rasterize_backward_rgbd");
  return GSR_OK;
}

// ---- depth segments: deep tiles' lists cut into `segments` pieces walked by their own waves -------------------
GSR_EXPORT size_t gsr_rasterize_backward_seg_workspace_bytes(unsigned img_height, unsigned img_width, int segments) {
  if (segments < 2) return 0;
  return (size_t)(segments - 1) * img_height * img_width * sizeof(float2);
}

GSR_EXPORT int gsr_rasterize_backward_seg(
    unsigned img_height, unsigned img_width, int num_points, const int32_t *gaussian_ids_sorted,
    const int32_t *tile_bins, const float *xys, const float *conics, const float *colors, const float *extra,
    const float *opacities, const float *background, float extra_background, const float *final_Ts,
    const int32_t *final_idx, const float *v_output, const float *v_output_extra, const float *v_output_alpha,
    float *v_xy, float *v_conic, float *v_colors, float *v_extra, float *v_opacity, int deep_tile_threshold,
    int accumulators_zeroed, int segments, int segment_min_entries, void *workspace, size_t workspace_bytes,
    gsr_stream_t stream) {
  const bool rgbd = extra != nullptr;
  if (segments < 2 || deep_tile_threshold <= 0) {
    if (rgbd)
      return gsr_rasterize_backward_rgbd(img_height, img_width, num_points, gaussian_ids_sorted, tile_bins, xys, conics,
                                         colors, extra, opacities, background, extra_background, final_Ts, final_idx,
                                         v_output, v_output_extra, v_output_alpha, v_xy, v_conic, v_colors, v_extra,
                                         v_opacity, deep_tile_threshold, accumulators_zeroed, stream);
    return gsr_rasterize_backward_ex(img_height, img_width, 16, num_points, gaussian_ids_sorted, tile_bins, xys, conics,
                                     colors, opacities, background, final_Ts, final_idx, v_output, v_output_alpha, v_xy,
                                     v_conic, v_colors, v_opacity, deep_tile_threshold, accumulators_zeroed, stream);
  }
  GSR_REQUIRE(img_height > 0 && img_width > 0, "rasterize_backward_seg: empty image");
  GSR_REQUIRE(num_points >= 0, "rasterize_backward_seg: num_points < 0");
  GSR_REQUIRE(segments <= 16, "rasterize_backward_seg: at most 16 segments");
  if (num_points == 0) return GSR_OK;
  GSR_REQUIRE(gaussian_ids_sorted && tile_bins && xys && conics && colors && opacities && background && final_Ts &&
                  final_idx && v_output && v_xy && v_conic && v_colors && v_opacity &&
                  (!rgbd || (v_output_extra && v_extra)),
              "rasterize_backward_seg: null pointer");
  GSR_REQUIRE(workspace && workspace_bytes >= gsr_rasterize_backward_seg_workspace_bytes(img_height, img_width, segments) &&
                  (reinterpret_cast<uintptr_t>(workspace) & 7) == 0,
              "rasterize_backward_seg: workspace too small or not 8-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  if (!accumulators_zeroed) {
    int rc = zero_grads(num_points, 3, v_xy, v_conic, v_colors, v_opacity, s);
    if (rc != GSR_OK) return rc;
    if (rgbd)
      if (int zrc = gsr_zero_async(v_extra, sizeof(float) * (size_t)num_points, s)) return zrc;
  }
  const int tiles_x = (int)gsr_cdiv(img_width, 16), tiles_y = (int)gsr_cdiv(img_height, 16);
  const int num_tiles = tiles_x * tiles_y;
  const unsigned base = gsr_xcd_grid(tiles_x, tiles_y);
  const int deep_arg = gsr_prepare_jobs(deep_tile_threshold, tiles_x, tiles_y, tile_bins, s);  // (threshold | order flag)
  deep_tile_threshold = gsr_deep_threshold(deep_tile_threshold);
  const int seg_min = segment_min_entries > deep_tile_threshold ? segment_min_entries : deep_tile_threshold;
  float2 *state = static_cast<float2 *>(workspace);
#define GSR_LAUNCH_BWD_SEG(RGBD_)                                                                                      \
  hipLaunchKernelGGL(raster_bwd_segstate_kernel<RGBD_>, dim3((unsigned)(segments - 1) * 4u * base), dim3(64), 0, s,    \
                     tiles_x, num_tiles, (int)img_width, (int)img_height, gaussian_ids_sorted,                         \
                     reinterpret_cast<const int2 *>(tile_bins), reinterpret_cast<const float2 *>(xys), conics, colors,  \
                     opacities, final_Ts, final_idx, v_output, extra, v_output_extra, deep_arg, base,                  \
                     segments, seg_min, state);                                                                        \
  hipLaunchKernelGGL(raster_bwd_segprefix_kernel, dim3((unsigned)(((size_t)img_height * img_width + 255) / 256)),     \
                     dim3(256), 0, s, tiles_x, (int)img_width, (int)img_height,                                        \
                     reinterpret_cast<const int2 *>(tile_bins), deep_tile_threshold, segments, seg_min, state);        \
  hipLaunchKernelGGL((raster_bwd_tile16_kernel<4, RGBD_, true>), dim3((unsigned)segments * 4u * base), dim3(64), 0, s, \
                     tiles_x, num_tiles, (int)img_width, (int)img_height, gaussian_ids_sorted,                         \
                     reinterpret_cast<const int2 *>(tile_bins), reinterpret_cast<const float2 *>(xys), conics, colors,  \
                     opacities, background, final_Ts, final_idx, v_output, v_output_alpha, v_xy, v_conic, v_colors,    \
                     v_opacity, extra, extra_background, v_output_extra, v_extra, deep_arg, base,                      \
                     (float *)nullptr, (unsigned char *)nullptr, (const int2 *)nullptr, 0, segments, seg_min,          \
                     (const float2 *)state)
  if (rgbd) {
    GSR_LAUNCH_BWD_SEG(true);
  } else {
    GSR_LAUNCH_BWD_SEG(false);
  }
#undef GSR_LAUNCH_BWD_SEG
  GSR_CHECK_LAUNCH("rasterize_backward_seg");
  return GSR_OK;
}

// ---- two-round lists: a tile's list = its range in tile_bins, then its range in tile_bins2 (+ idx_base2) ----
GSR_EXPORT int gsr_rasterize_backward_two(
    unsigned img_height, unsigned img_width, int num_points, const int32_t *gaussian_ids_sorted,
    const int32_t *tile_bins, const int32_t *tile_bins2, int idx_base2, const float *xys, const float *conics,
    const float *colors, const float *extra, const float *opacities, const float *background, float extra_background,
    const float *final_Ts, const int32_t *final_idx, const float *v_output, const float *v_output_extra,
    const float *v_output_alpha, float *v_xy, float *v_conic, float *v_colors, float *v_extra, float *v_opacity,
    int deep_tile_threshold, int accumulators_zeroed, gsr_stream_t stream) {
  GSR_REQUIRE(img_height > 0 && img_width > 0, "rasterize_backward_two: empty image");
  GSR_REQUIRE(num_points >= 0 && idx_base2 >= 0, "rasterize_backward_two: negative size");
  if (num_points == 0) return GSR_OK;
  const bool rgbd = extra != nullptr;
  GSR_REQUIRE(gaussian_ids_sorted && tile_bins && tile_bins2 && xys && conics && colors && opacities && background &&
                  final_Ts && final_idx && v_output && v_xy && v_conic && v_colors && v_opacity &&
                  (!rgbd || (v_output_extra && v_extra)),
              "rasterize_backward_two: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (!accumulators_zeroed) {
    int rc = zero_grads(num_points, 3, v_xy, v_conic, v_colors, v_opacity, s);
    if (rc != GSR_OK) return rc;
    if (rgbd)
      if (int zrc = gsr_zero_async(v_extra, sizeof(float) * (size_t)num_points, s)) return zrc;
  }
  const int tiles_x = (int)gsr_cdiv(img_width, 16), tiles_y = (int)gsr_cdiv(img_height, 16);
  const int num_tiles = tiles_x * tiles_y;
  const unsigned base = gsr_xcd_grid(tiles_x, tiles_y);
  const int deep = gsr_deep_threshold(deep_tile_threshold);  // (two-round lists: the static block order)
  const dim3 grd(deep ? 4 * base : base), blk(64);
  if (rgbd)
    hipLaunchKernelGGL((raster_bwd_tile16_kernel<4, true>), grd, blk, 0, s, tiles_x, num_tiles, (int)img_width,
                       (int)img_height, gaussian_ids_sorted, reinterpret_cast<const int2 *>(tile_bins),
                       reinterpret_cast<const float2 *>(xys), conics, colors, opacities, background, final_Ts, final_idx,
                       v_output, v_output_alpha, v_xy, v_conic, v_colors, v_opacity, extra, extra_background,
                       v_output_extra, v_extra, deep, base, (float *)nullptr, (unsigned char *)nullptr,
                       reinterpret_cast<const int2 *>(tile_bins2), idx_base2);
  else
    hipLaunchKernelGGL((raster_bwd_tile16_kernel<4, false>), grd, blk, 0, s, tiles_x, num_tiles, (int)img_width,
                       (int)img_height, gaussian_ids_sorted, reinterpret_cast<const int2 *>(tile_bins),
                       reinterpret_cast<const float2 *>(xys), conics, colors, opacities, background, final_Ts, final_idx,
                       v_output, v_output_alpha, v_xy, v_conic, v_colors, v_opacity, (const float *)nullptr, 0.f,
                       (const float *)nullptr, (float *)nullptr, deep, base, (float *)nullptr, (unsigned char *)nullptr,
                       reinterpret_cast<const int2 *>(tile_bins2), idx_base2);
  GSR_CHECK_LAUNCH("rasterize_backward_two");
  return GSR_OK;
}

// ---- deterministic backward ----------------------------------------------------------
GSR_EXPORT size_t gsr_rasterize_backward_det_workspace_bytes(int list_capacity) {
  if (list_capacity <= 0) return 0;
  return (((size_t)list_capacity * kPartialStride * sizeof(float) + 255) & ~(size_t)255) +
         (((size_t)list_capacity + 255) & ~(size_t)255);
}

GSR_EXPORT int gsr_rasterize_backward_det(
    unsigned img_height, unsigned img_width, int num_points, int list_capacity, const int32_t *gaussian_ids_sorted,
    const int32_t *tile_bins, const float *xys, const float *conics, const float *colors, const float *extra,
    const float *opacities, const float *background, float extra_background, const float *final_Ts,
    const int32_t *final_idx, const float *v_output, const float *v_output_extra, const float *v_output_alpha,
    const int32_t *order, const int32_t *cum_sorted, int num_bands, const int32_t *slot_of_entry, void *workspace,
    size_t workspace_bytes, float *v_xy, float *v_conic, float *v_colors, float *v_extra, float *v_opacity,
    gsr_stream_t stream) {
  GSR_REQUIRE(img_height > 0 && img_width > 0, "rasterize_backward_det: empty image");
  GSR_REQUIRE(num_points >= 0 && list_capacity >= 0, "rasterize_backward_det: negative size");
  if (num_points == 0) return GSR_OK;
  const bool rgbd = extra != nullptr;
  GSR_REQUIRE(gaussian_ids_sorted && tile_bins && xys && conics && colors && opacities && background && final_Ts &&
                  final_idx && v_output && order && cum_sorted && slot_of_entry && v_xy && v_conic && v_colors &&
                  v_opacity && (!rgbd || (v_output_extra && v_extra)),
              "rasterize_backward_det: null pointer");
  GSR_REQUIRE(num_bands >= 1, "rasterize_backward_det: num_bands < 1");
  GSR_REQUIRE(workspace && workspace_bytes >= gsr_rasterize_backward_det_workspace_bytes(list_capacity) &&
                  (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
              "rasterize_backward_det: workspace too small or not 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  float *partials = static_cast<float *>(workspace);
  unsigned char *pflags = reinterpret_cast<unsigned char *>(workspace) +
                          (((size_t)list_capacity * kPartialStride * sizeof(float) + 255) & ~(size_t)255);
  if (int zrc = gsr_zero_async(pflags, ((size_t)list_capacity + 3) & ~(size_t)3, s)) return zrc;
  const int tiles_x = (int)gsr_cdiv(img_width, 16), tiles_y = (int)gsr_cdiv(img_height, 16);
  const int num_tiles = tiles_x * tiles_y;
  const unsigned base = gsr_xcd_grid(tiles_x, num_tiles / tiles_x);
  // (one wave per tile: a split tile would need one partial row per sub-tile wave)
  if (rgbd) {
    hipLaunchKernelGGL((raster_bwd_tile16_kernel<4, true>), dim3(base), dim3(64), 0, s, tiles_x, num_tiles,
                       (int)img_width, (int)img_height, gaussian_ids_sorted, reinterpret_cast<const int2 *>(tile_bins),
                       reinterpret_cast<const float2 *>(xys), conics, colors, opacities, background, final_Ts, final_idx,
                       v_output, v_output_alpha, v_xy, v_conic, v_colors, v_opacity, extra, extra_background,
                       v_output_extra, v_extra, 0, base, partials, pflags, (const int2 *)nullptr, 0);
    hipLaunchKernelGGL(reduce_partials_kernel<true>, dim3(gsr_cdiv(num_points, 256)), dim3(256), 0, s, num_points,
                       num_bands, list_capacity, order, cum_sorted, slot_of_entry, (const float *)partials,
                       (const unsigned char *)pflags, v_xy, v_conic, v_colors, v_opacity, v_extra);
  } else {
    hipLaunchKernelGGL((raster_bwd_tile16_kernel<4, false>), dim3(base), dim3(64), 0, s, tiles_x, num_tiles,
                       (int)img_width, (int)img_height, gaussian_ids_sorted, reinterpret_cast<const int2 *>(tile_bins),
                       reinterpret_cast<const float2 *>(xys), conics, colors, opacities, background, final_Ts, final_idx,
                       v_output, v_output_alpha, v_xy, v_conic, v_colors, v_opacity, (const float *)nullptr, 0.f,
                       (const float *)nullptr, (float *)nullptr, 0, base, partials, pflags, (const int2 *)nullptr, 0);
    hipLaunchKernelGGL(reduce_partials_kernel<false>, dim3(gsr_cdiv(num_points, 256)), dim3(256), 0, s, num_points,
                       num_bands, list_capacity, order, cum_sorted, slot_of_entry, (const float *)partials,
                       (const unsigned char *)pflags, v_xy, v_conic, v_colors, v_opacity, (float *)nullptr);
  }
  GSR_CHECK_LAUNCH("rasterize_backward_det");
  return GSR_OK;
}

int gsr_set_fwd_staged_counter(unsigned long long *counter);  // raster_fwd.hip

int gsr_set_fwd_wave_trace(unsigned long long *buf, unsigned capacity);  // raster_fwd.hip

GSR_EXPORT int gsr_debug_wave_trace(unsigned long long *records, unsigned capacity_waves) {
  int rc = gsr_set_fwd_wave_trace(records, capacity_waves);
  if (rc != GSR_OK) return rc;
  gsr::WaveTrace t = {records ? records + 4ull * capacity_waves : nullptr, capacity_waves};
  GSR_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_trace), &t, sizeof(t)));
  return GSR_OK;
}

GSR_EXPORT int gsr_debug_count_staged(unsigned long long *counters) {
  int rc = gsr_set_fwd_staged_counter(counters);
  if (rc != GSR_OK) return rc;
  unsigned long long *b = counters ? counters + 1 : nullptr;
  GSR_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_staged), &b, sizeof(b)));
  return GSR_OK;
}
