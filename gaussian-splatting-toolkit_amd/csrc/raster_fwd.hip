// raster_fwd.hip -- front-to-back alpha compositing of depth-sorted splats per
// tile (gfx950).
//
// Compositing rule restated from rasterizer/cuda/csrc/forward.cu:278-395
// (3 channels) and :159-276 (N channels):
//   sigma = .5(a dx^2 + c dy^2) + b dx dy,  d = xy - (col,row)   (no +0.5)
//   alpha = min(0.999, opac * exp(-sigma)); skip if sigma<0 or alpha<1/255
//   if T(1-alpha) <= 1e-4 the pixel is finished (that splat is NOT drawn)
//   C += rgb*alpha*T; T *= 1-alpha; final_idx = index in the sorted list
//   out = C + T*background
//
// Two mappings:
//  * tile16 (block_width 16, 3 channels -- what the models use): ONE wave64 per
//    16x16 tile, each lane owns a 2x2 pixel quad.  Splats are staged 64 at a
//    time through LDS (one gather per lane, coalesced index read) and consumed
//    by wave-uniform broadcast reads, so every LDS word feeds 4 pixels per lane
//    and no workgroup barrier is ever waited on by a second wave.  dx/dy terms
//    are shared across the quad.  Workgroups are remapped so that 8x4-tile blocks go to the
//    XCDs round-robin (shared splats stay in one L2, every XCD sees the whole frame).
//  * generic (any block_width in [2,16], any channel count; 32 channels per pass): one lane
//    per pixel, block_width^2 lanes per tile.
#include <algorithm>

#include "raster_common.h"

namespace {

using namespace gsr;

// measurement hook: when non-null, the tile16 kernels add the number of list entries they
// stage to *g_fwd_staged (gsr_debug_count_staged, used by bench.py to price the roofline on
// the entries actually touched -- tiles stop early once every pixel is saturated)
__device__ unsigned long long *g_fwd_staged = nullptr;
__device__ gsr::WaveTrace g_fwd_trace = {nullptr, 0u};

// ------------------------------------------------------------------ tile16
// The kernel is VALU-issue bound (rocprof: VALU busy ~86 %, LDS and memory
// idle).  Measured dead ends, kept out on purpose: writing the pixel math on
// float2 pairs so that it compiles to v_pk_{mul,add,fma}_f32 cut the VALU
// instruction count by ~30 % and made the kernel 9 % SLOWER (packed fp32 issues
// at half rate on gfx950 and needs s_nop hazards) -- so the build also turns
// the SLP vectorizer off for these files (Makefile).
//
// Per-pixel state is ONE float: T > 0 is the live transmittance, T < 0 means the
// pixel is finished (or outside the image) and |T| is the value to report.  A
// finished pixel gives next_T < 0, so it can neither draw nor finish again.
//
// RGBD: a fourth channel (one scalar per Gaussian, e.g. its depth) is composited in the
// same pass into its own [H,W] image over background `bg_extra` (the
// models run a second full pass for the depth image, vanilla_gs.py:840-855).
// ---- depth segments (one walk since round 6) ------------------------------------------------------
// On a tile grid that cannot fill the chip the kernel lasts as long as its deepest tile's serial walk.  The list of a
// tile that is split over four waves is therefore also cut into up to `seg_count` runs, and compositing is ASSOCIATIVE in
// (C, T): a run walked from T = 1 yields its colour sums C_k and its transmittance product P_k, and the pixel is
// C = sum_k (prod_{j<k} P_j) C_k.  Every run is walked ONCE, from T = 1, by its own wave (raster_fwd_tile16_kernel<.,
// true>), leaving its raw state (colour sums, signed T, last drawn index).  raster_fwd_segresolve_kernel then, per
// pixel, multiplies the prefixes and finds the first run k* in which  prefix x P_k  comes down to the stop rule's 1e-4
// (or which finished on its own): in front of k* no splat can have tripped `T (1 - alpha) <= 1e-4` (T only falls), so
// those runs' sums enter scaled by their prefix; from the start of run k* a wave of its own RE-WALKS the list with the
// true incoming T -- the unchanged rule, the exact stop and final_idx (forward.cu:360-380) -- for the pixels that
// cross there (raster_fwd_segrewalk_kernel: one block per flagged (sub-tile, run), the runs in parallel).  A translucent model (what the small grids of the coarse-to-fine schedule render) almost never crosses: one
// walk.  Rounds 4-5 walked every list twice (a transmittance pre-pass, then the runs from their true T).
// Equal to the single walk to rounding (T and C are sums / products of run-wise partial results), not bitwise.

template <bool RGBD, bool SEG = false>
__global__ __launch_bounds__(64) void raster_fwd_tile16_kernel(
    const int tiles_x, const int num_tiles, const int img_w, const int img_h,
    const int *__restrict__ ids_sorted, const int2 *__restrict__ tile_bins,
    const float2 *__restrict__ xys, const float *__restrict__ conics,
    const float *__restrict__ colors, const float *__restrict__ opacities,
    const float *__restrict__ background, float *__restrict__ out_img,
    float *__restrict__ final_Ts, int *__restrict__ final_idx, const float *__restrict__ extra,
    const float bg_extra, float *__restrict__ out_extra, const int deep_threshold, const unsigned base_grid,
    float *__restrict__ out_alpha, unsigned *__restrict__ zero_ptr, const unsigned zero_words,
    const int round, int *__restrict__ tile_flags, const int idx_base, const int seg_count = 1, const int seg_min = 0,
    float4 *__restrict__ seg_raw = nullptr, int *__restrict__ seg_last = nullptr,
    float *__restrict__ seg_extra = nullptr, float *__restrict__ seg_marks = nullptr) {
  // Two-round compositing (gsr_rasterize_forward_round): the lists of the nearest
  // Gaussians are a PREFIX of every tile's list.  round 1 composites such prefix lists and leaves the per-pixel
  // state of a wave that still has a live pixel RAW -- final_Ts = signed T (< 0: finished), out_img / out_extra = C
  // without background, final_idx -- and sets that wave's sub-tile bits in tile_flags[tile] (zeroed by the caller);
  // a wave whose pixels have all finished writes its final values as the single walk does.  round 2 RESUMES the
  // flagged sub-tiles from that state over the lists of the remaining Gaussians (tile_bins relative to idx_base
  // in ids_sorted) and finalises them; unflagged tiles cost it one load.  The per-pixel instruction sequence is
  // that of one walk over the concatenated list: bit-identical results.  round 0: the single walk.
  // (gsr_rasterize_forward_ex) the launch also clears `zero_words` words at `zero_ptr` -- the gradient
  // accumulators of the coming backward: 36 MB of stores that vanish inside this VALU-bound kernel
  // instead of a bandwidth-bound launch of their own -- every workgroup its slice, before any exit
  if (zero_ptr) {
    const unsigned per = (zero_words + gridDim.x - 1) / gridDim.x;
    const unsigned w0 = blockIdx.x * per, w1 = min(w0 + per, zero_words);
    for (unsigned i = w0 + threadIdx.x; i < w1; i += 64) zero_ptr[i] = 0u;
  }
  __shared__ SplatA sA[kChunk];
  __shared__ SplatB sB[kChunk];
  __shared__ SplatC sC[kChunk];

  int2 range = make_int2(0, 0);
  unsigned blk = blockIdx.x;
  int seg_k = 0;
  if constexpr (SEG) {  // block = run * (4 base_grid) + the block of the unsegmented launch
    seg_k = (int)(blk / (4u * base_grid));
    blk -= (unsigned)seg_k * (4u * base_grid);
  }
  const WaveTrace trace = g_fwd_trace;
  const unsigned long long trace_t0 = trace_begin(trace);
  const unsigned long long stats_t0 = gsr_deep_ordered(deep_threshold) ? wall_clock64() : 0ull;
  const TileJob job = tile_job(blk, base_grid, tiles_x, num_tiles / tiles_x, tile_bins, deep_threshold, range);
  const int tile = job.tile;
  int allowed = job.allowed;
  if (tile < 0) return;
  const int trace_len = range.y - range.x;
  bool split = false;
  if constexpr (SEG) {
    const int len = range.y - range.x;
    if (allowed == 15 || len <= seg_min) {  // not split: one walk, by run 0's block
      if (seg_k > 0) return;
    } else {
      const int sl = seg_len_of(len, seg_count);
      if (seg_k >= min(seg_count, (len + sl - 1) / sl)) return;
      split = true;
      range.x += seg_k * sl;
      range.y = min(range.x + sl, range.y);
      // SATURATION MARKS.  Every run walks from T = 1, so a scene that saturates early (a deep list of opaque splats:
      // the single walk stops after a fraction of it) would be walked in full, run by run.  A run's wave therefore
      // leaves the LARGEST transmittance any of its 64 pixels ends the run with (0 for a finished pixel) in
      // seg_marks[16 (4 tile + sub-tile) + run], and a later run first multiplies the marks of the runs in front of it
      // that have been written so far: a pixel's true incoming T is at most the product of its own runs' T, hence at
      // most the product of the maxima -- at 1e-4 or below EVERY pixel of the sub-tile has crossed the stop rule in
      // front of this run, which then draws nothing and is never read (raster_fwd_segresolve_kernel stops at the
      // crossing run).  Exact: it only skips work whose result is unused.  Blocks start in block order (run-major) and
      // all runs of a sub-tile sit on one XCD (4 base_grid is a multiple of 8), so with a grid beyond the chip's wave
      // slots the later runs find the earlier marks; a mark not yet written (0) counts as 1.
      if (seg_marks && seg_k > 0) {
        const float *mk = seg_marks + 16 * (4 * (size_t)tile + __builtin_ctz(allowed));
        float v = 1.f;
        if ((int)threadIdx.x < seg_k) {
          const float m = __hip_atomic_load(mk + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          v = m > 0.f ? m : 1.f;
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) v *= __shfl_xor(v, o);
        if (__builtin_amdgcn_readfirstlane(__float_as_uint(v)) <= __float_as_uint(0.9f * GSR_T_EPS)) return;  // (v > 0)
      }
    }
  }
  if (round == 2) {  // only the sub-tiles round 1 left raw
    allowed &= tile_flags[tile];
    if (allowed == 0) return;
  }
  range.x += idx_base;
  range.y += idx_base;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int lane = threadIdx.x;
  const int qx = tx * 16 + (lane & 7), qy = ty * 16 + (lane >> 3);
  const float fx0 = (float)qx, fx1 = (float)(qx + 8);
  const float fy0 = (float)qy, fy1 = (float)(qy + 8);
  const float tx0 = (float)(tx * 16), ty0 = (float)(ty * 16);

  // pixel p = (qx + 8*(p&1), qy + 8*(p>>1))
  float T[4], cr[4], cg[4], cb[4], ce[4];
  int last[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int col = qx + 8 * (p & 1), row = qy + 8 * (p >> 1);
    const bool inside = col < img_w && row < img_h && ((allowed >> p) & 1);
    T[p] = inside ? 1.f : -1.f;
    cr[p] = cg[p] = cb[p] = ce[p] = 0.f;
    last[p] = 0;
    if (round == 2 && inside) {  // resume from the raw state round 1 left
      const size_t pid = (size_t)row * img_w + col;
      T[p] = final_Ts[pid];
      cr[p] = out_img[3 * pid];
      cg[p] = out_img[3 * pid + 1];
      cb[p] = out_img[3 * pid + 2];
      if constexpr (RGBD) ce[p] = out_extra[pid];
      last[p] = final_idx[pid];
    }
    // (SEG: every run starts from T = 1 -- raster_fwd_segresolve_kernel scales its sums by the runs in front of it)
  }

  // sub-tiles that still have a live pixel (wave-uniform)
  auto live_subtiles = [&]() {
    int m = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) m |= __any(T[p] > 0.f) ? (1 << p) : 0;
    return m;
  };

  unsigned long long *const staged = g_fwd_staged;
  int live = live_subtiles();
#if GSR_STAGE_AHEAD
  // chunk n's records are committed from registers loaded one iteration ago; behind the commit the loads of chunk
  // n + 1 (geometry + colours, from the list entries fetched one iteration ago) and the list entries of chunk n + 2
  // go out, and the walk over chunk n hides them (pays where a chunk's walk is short -- the
  // sub-tile waves of split tiles, i.e. mid-size grids and deep tiles).  An empty list loads nothing; lanes outside a
  // NON-empty range read Gaussian 0, which then exists.
  // Not in the RGB + depth instantiation: the one workload that trains through it (co-gs, 3 M Gaussians at 4K, tiles
  // that saturate within a few chunks) lost 14 % of its render phase to the extra requests
  // (profiles/r05_stage_ahead_ab.txt); it stages with stage_chunk as before.
  constexpr bool kAhead = !RGBD;
  int g_cur = 0, g_next = 0;
  StageRegs regs = {};
  if (kAhead && range.x < range.y) {
    g_cur = stage_load_id(range.x + lane < range.y, range.x + lane, ids_sorted);
    g_next = stage_load_id(range.x + kChunk + lane < range.y, range.x + kChunk + lane, ids_sorted);
    regs = stage_load_attrs(g_cur, xys, conics, colors, opacities, RGBD ? extra : nullptr);
  }
#else
  constexpr bool kAhead = false;
  [[maybe_unused]] int g_cur = 0, g_next = 0;
  [[maybe_unused]] StageRegs regs = {};
#endif
  for (int base = range.x; base < range.y && live != 0; base += kChunk) {
    const int sidx = base + lane;
    int count;
    if constexpr (kAhead) {
      count = stage_commit(lane, sidx < range.y, sidx, g_cur, regs, tx0, ty0, sA, sB, sC, nullptr, staged, allowed);
      __syncthreads();
      g_cur = g_next;
      regs = stage_load_attrs(g_cur, xys, conics, colors, opacities, RGBD ? extra : nullptr);
      g_next = stage_load_id(sidx + 2 * kChunk < range.y, sidx + 2 * kChunk, ids_sorted);
    } else {
      count = stage_chunk(lane, sidx < range.y, sidx, tx0, ty0, ids_sorted, xys, conics, colors, opacities, sA, sB, sC,
                          nullptr, RGBD ? extra : nullptr, staged, allowed);
      __syncthreads();
    }
    for (int t = 0; t < count; ++t) {
      if ((t & 7) == 7) {
        live = live_subtiles();
        if (live == 0) break;
      }
      const SplatC C = sC[t];
      const int m = C.mask & live;
      if (m == 0) continue;
      const SplatA A = sA[t];
      const SplatB B = sB[t];
      const float dx0 = A.x - fx0, dx1 = A.x - fx1;
      const float dy0 = A.y - fy0, dy1 = A.y - fy1;
      const float ax0 = A.ha * dx0 * dx0, ax1 = A.ha * dx1 * dx1;
      const float cy0 = B.hc * dy0 * dy0, cy1 = B.hc * dy1 * dy1;
      const float bx0 = A.b * dx0, bx1 = A.b * dx1;
      const float sig[4] = {(ax0 + cy0) + bx0 * dy0, (ax1 + cy0) + bx1 * dy0,
                            (ax0 + cy1) + bx0 * dy1, (ax1 + cy1) + bx1 * dy1};
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if (!(m & (1 << p))) continue;  // wave-uniform: whole sub-tile out of reach / finished
        const float sigma = sig[p];
        const float alpha = fminf(GSR_ALPHA_MAX_FWD, B.opac * __expf(-sigma));
        const float Tp = T[p];
        const float next_T = Tp * (1.f - alpha);
        // flat selects (v_cndmask); nested ternaries turn into exec-mask branches
        const bool hit = !(sigma < 0.f || alpha < GSR_ALPHA_MIN);
        const bool go = next_T > GSR_T_EPS;
        const bool draw = hit && go;
        const float dead = __uint_as_float(__float_as_uint(Tp) | 0x80000000u);  // -|T|
        const float upd = go ? next_T : dead;
        const float vis = draw ? alpha * Tp : 0.f;
        cr[p] += B.r * vis;
        cg[p] += B.g * vis;
        cb[p] += C.blue * vis;
        if constexpr (RGBD) ce[p] += C.extra * vis;
        last[p] = draw ? C.sidx : last[p];
        T[p] = hit ? upd : Tp;
      }
    }
    __syncthreads();
    live = live_subtiles();
  }

  if (round == 1 && live_subtiles() != 0) {  // a pixel of this wave is still live: raw state, flag its sub-tiles
    if (lane == 0) atomicOr(&tile_flags[tile], allowed);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int col = qx + 8 * (p & 1), row = qy + 8 * (p >> 1);
      if (col < img_w && row < img_h && ((allowed >> p) & 1)) {
        const size_t pid = (size_t)row * img_w + col;
        final_Ts[pid] = T[p];
        final_idx[pid] = last[p];
        out_img[3 * pid] = cr[p];
        out_img[3 * pid + 1] = cg[p];
        out_img[3 * pid + 2] = cb[p];
        if constexpr (RGBD) out_extra[pid] = ce[p];
      }
    }
    return;
  }
  if constexpr (SEG) {
    if (split) {  // the run's raw state; the combine pass finishes the pixel
      const size_t pixels = (size_t)img_w * img_h;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int col = qx + 8 * (p & 1), row = qy + 8 * (p >> 1);
        if (col < img_w && row < img_h && ((allowed >> p) & 1)) {
          const size_t at = (size_t)seg_k * pixels + (size_t)row * img_w + col;
          seg_raw[at] = make_float4(cr[p], cg[p], cb[p], T[p]);
          seg_last[at] = last[p];
          if constexpr (RGBD) seg_extra[at] = ce[p];
        }
      }
      if (seg_marks) {
        const int sub = __builtin_ctz(allowed);
        float m = fmaxf(sub == 0 ? T[0] : sub == 1 ? T[1] : sub == 2 ? T[2] : T[3], 0.f);  // (finished / outside: 0)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0)
          __hip_atomic_store(seg_marks + 16 * (4 * (size_t)tile + sub) + seg_k, fmaxf(m, 1.17549435e-38f), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
      }
      trace_end(trace, trace_t0, tile, allowed, range.y - range.x);  // (a run's wave: the run's length)
      return;
    }
  }
  // wave-uniform -> scalar loads
  const float bg0 = background[0], bg1 = background[1], bg2 = background[2];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int col = qx + 8 * (p & 1), row = qy + 8 * (p >> 1);
    if (col < img_w && row < img_h && ((allowed >> p) & 1)) {
      const size_t pid = (size_t)row * img_w + col;
      const float Tp = fabsf(T[p]);
      final_Ts[pid] = Tp;
      if (out_alpha) out_alpha[pid] = 1.f - Tp;
      final_idx[pid] = last[p];
      out_img[3 * pid] = cr[p] + Tp * bg0;
      out_img[3 * pid + 1] = cg[p] + Tp * bg1;
      out_img[3 * pid + 2] = cb[p] + Tp * bg2;
      if constexpr (RGBD) out_extra[pid] = ce[p] + Tp * bg_extra;
    }
  }
  trace_end(trace, trace_t0, tile, allowed, trace_len);
  job_stats_end(job, 0, stats_t0, trace_len);
}

__global__ __launch_bounds__(256) void seg_marks_clear_kernel(float *__restrict__ marks, const unsigned n) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) marks[i] = 0.f;
}

// The crossing test's margin: prefix x P_k against 1e-4 (1 + margin).  The single walk's T at the end of run k and the
// product of run products differ by rounding (~1e-7 per factor); a pixel within the margin is re-walked from the start
// of that run with its true incoming T, and the re-walk -- the unchanged rule -- is the arbiter.
#define GSR_SEG_CROSS (GSR_T_EPS * 1.001f)

// Resolve pass of the depth segments: one wave per (split tile, sub-tile), lane = one pixel of the 8x8 sub-tile.  Per
// pixel, over the runs in list order: prefix = product of the runs' T in front of run k; the FIRST run whose own walk
// finished (signed T < 0) or whose prefix x T falls to the stop rule's threshold is the pixel's crossing run k*; the
// colour sums of the runs in front of k* enter scaled by their prefix, `last` is the largest drawn index.
//  * a pixel without a crossing run is FINAL here: C = the scaled sums, T = the full product;
//  * the others leave their state at the start of k* -- (C so far, incoming T) in the pixel's record of run 0, the
//    last drawn index and the extra channel's sum likewise, k* in `seg_kstar` (-1: final) -- and the wave leaves the
//    set of crossing runs of its sub-tile as a bit mask in seg_flags[4 tile + sub-tile]: what the re-walk launch reads.
template <bool RGBD>
__global__ __launch_bounds__(64) void raster_fwd_segresolve_kernel(
    const int tiles_x, const int num_tiles, const int img_w, const int img_h, const int2 *__restrict__ tile_bins,
    const float *__restrict__ background, float *__restrict__ out_img, float *__restrict__ final_Ts,
    int *__restrict__ final_idx, const float bg_extra, float *__restrict__ out_extra, const int deep_threshold,
    const unsigned base_grid, float *__restrict__ out_alpha, const int seg_count, const int seg_min,
    float4 *__restrict__ seg_raw, int *__restrict__ seg_last, float *__restrict__ seg_extra,
    int *__restrict__ seg_kstar, int *__restrict__ seg_flags) {
  int2 range = make_int2(0, 0);
  const TileJob job = tile_job(blockIdx.x, base_grid, tiles_x, num_tiles / tiles_x, tile_bins, deep_threshold, range);
  const int tile = job.tile, allowed = job.allowed;
  if (tile < 0) return;
  const int len = range.y - range.x;
  if (allowed == 15 || len <= seg_min) return;  // not split: the run kernel's block 0 wrote the final values
  const int sl = seg_len_of(len, seg_count);
  const int nseg = min(seg_count, (len + sl - 1) / sl);
  const int sub = __builtin_ctz(allowed);
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int lane = threadIdx.x;
  const int col = tx * 16 + 8 * (sub & 1) + (lane & 7), row = ty * 16 + 8 * (sub >> 1) + (lane >> 3);
  const bool inside = col < img_w && row < img_h;
  const size_t pixels = (size_t)img_w * img_h, pid = inside ? (size_t)row * img_w + col : 0;

  float cr = 0.f, cg = 0.f, cb = 0.f, ce = 0.f, prefix = 1.f;
  int last = 0, kstar = -1;
  for (int k = 0; k < nseg; ++k) {  // (nseg is wave-uniform; the loads do not depend on one another)
    const float4 r = seg_raw[(size_t)k * pixels + pid];
    const int l = seg_last[(size_t)k * pixels + pid];
    float e = 0.f;
    if constexpr (RGBD) e = seg_extra[(size_t)k * pixels + pid];
    const bool open = kstar < 0;
    const bool cross = open && (r.w < 0.f || prefix * r.w <= GSR_SEG_CROSS);
    kstar = cross ? k : kstar;
    const bool take = open && !cross;
    // (selects, not products with 0: the records of runs behind the crossing run may never have been written)
    cr = take ? cr + prefix * r.x : cr;
    cg = take ? cg + prefix * r.y : cg;
    cb = take ? cb + prefix * r.z : cb;
    if constexpr (RGBD) ce = take ? ce + prefix * e : ce;
    last = take ? max(last, l) : last;
    prefix = take ? prefix * r.w : prefix;  // stays at the crossing run's incoming T
  }
  if (!inside) kstar = -1;  // (a lane outside the image read pixel 0's records: nothing of it is used)
  unsigned mask = kstar >= 0 ? (1u << kstar) : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mask |= __shfl_xor(mask, o);
  if (lane == 0) seg_flags[4 * tile + sub] = (int)mask;
  if (!inside) return;
  seg_kstar[pid] = kstar;
  if (kstar >= 0) {
    seg_raw[pid] = make_float4(cr, cg, cb, prefix);
    seg_last[pid] = last;
    if constexpr (RGBD) seg_extra[pid] = ce;
    return;
  }
  final_Ts[pid] = prefix;
  if (out_alpha) out_alpha[pid] = 1.f - prefix;
  final_idx[pid] = last;
  out_img[3 * pid] = cr + prefix * background[0];
  out_img[3 * pid + 1] = cg + prefix * background[1];
  out_img[3 * pid + 2] = cb + prefix * background[2];
  if constexpr (RGBD) out_extra[pid] = ce + prefix * bg_extra;
}

// sigma in the operation order of the tile16 kernel's compiled code, pinned: two products each way, their ROUNDED sum,
// one fused multiply-add for the cross term -- left to the compiler the single-pixel form contracts `ax + cy` into an
// fma (cy is used once here, twice there) and sigma comes out one ulp apart.
__device__ __forceinline__ float sigma_in_tile16_order(const float ha, const float b, const float hc, const float dx,
                                                       const float dy) {
#pragma clang fp contract(off)
  const float ax = (ha * dx) * dx;
  const float cy = (hc * dy) * dy;
  const float s = ax + cy;
  const float bx = b * dx;
  return __builtin_fmaf(bx, dy, s);
}

// Re-walk of the depth segments: block = run * (4 base_grid) + the (tile, sub-tile) block of the unsegmented launch; it
// leaves at once unless the resolve pass flagged its run for its sub-tile.  Its pixels are those whose crossing run is
// THIS run: they start from the state the resolve pass left (the sums of the runs in front, the true incoming T, > 1e-4
// by construction) and are composited with the unchanged rule from the start of the run until they finish -- normally
// inside the run; a pixel the margin flagged for nothing simply walks on with its true T, to the end of the list if
// need be.  Every crossing pixel is finished by exactly one block.  The per-splat expressions are those of
// raster_fwd_tile16_kernel, operand for operand.
template <bool RGBD>
__global__ __launch_bounds__(64) void raster_fwd_segrewalk_kernel(
    const int tiles_x, const int num_tiles, const int img_w, const int img_h, const int *__restrict__ ids_sorted,
    const int2 *__restrict__ tile_bins, const float2 *__restrict__ xys, const float *__restrict__ conics,
    const float *__restrict__ colors, const float *__restrict__ opacities, const float *__restrict__ background,
    float *__restrict__ out_img, float *__restrict__ final_Ts, int *__restrict__ final_idx,
    const float *__restrict__ extra, const float bg_extra, float *__restrict__ out_extra, const int deep_threshold,
    const unsigned base_grid, float *__restrict__ out_alpha, const int seg_count, const int seg_min,
    const float4 *__restrict__ seg_raw, const int *__restrict__ seg_last, const float *__restrict__ seg_extra,
    const int *__restrict__ seg_kstar, const int *__restrict__ seg_flags) {
  __shared__ SplatA sA[kChunk];
  __shared__ SplatB sB[kChunk];
  __shared__ SplatC sC[kChunk];
  int2 range = make_int2(0, 0);
  unsigned blk = blockIdx.x;
  const int seg_k = (int)(blk / (4u * base_grid));
  blk -= (unsigned)seg_k * (4u * base_grid);
  const TileJob job = tile_job(blk, base_grid, tiles_x, num_tiles / tiles_x, tile_bins, deep_threshold, range);
  const int tile = job.tile, allowed = job.allowed;
  if (tile < 0) return;
  const int len = range.y - range.x;
  if (allowed == 15 || len <= seg_min) return;
  const int sub = __builtin_ctz(allowed);
  if (!((seg_flags[4 * tile + sub] >> seg_k) & 1)) return;
  const int sl = seg_len_of(len, seg_count);
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int lane = threadIdx.x;
  const int col = tx * 16 + 8 * (sub & 1) + (lane & 7), row = ty * 16 + 8 * (sub >> 1) + (lane >> 3);
  const bool inside = col < img_w && row < img_h;
  const size_t pid = inside ? (size_t)row * img_w + col : 0;
  const float fx = (float)col, fy = (float)row;
  const float tx0 = (float)(tx * 16), ty0 = (float)(ty * 16);
  const bool mine = inside && seg_kstar[pid] == seg_k;
  const float4 st = seg_raw[pid];
  float cr = st.x, cg = st.y, cb = st.z, ce = 0.f;
  float T = mine ? st.w : -1.f;
  int last = seg_last[pid];
  if constexpr (RGBD) ce = seg_extra[pid];
  bool live = __any(T > 0.f);
  for (int base = range.x + seg_k * sl; base < range.y && live; base += kChunk) {
    const int sidx = base + lane;
    const int count = stage_chunk_flat(lane, sidx < range.y, sidx, tx0, ty0, ids_sorted, xys, conics, colors, opacities, sA,
                                       sB, sC, RGBD ? extra : nullptr, allowed);
    __syncthreads();
    for (int t = 0; t < count; ++t) {
      if ((t & 7) == 7 && !__any(T > 0.f)) break;
      const SplatA A = sA[t];
      const SplatB B = sB[t];
      const SplatC C = sC[t];
      const float dx = A.x - fx, dy = A.y - fy;
      const float sigma = sigma_in_tile16_order(A.ha, A.b, B.hc, dx, dy);
      const float alpha = fminf(GSR_ALPHA_MAX_FWD, B.opac * __expf(-sigma));
      const float Tp = T;
      const float next_T = Tp * (1.f - alpha);
      const bool hit = !(sigma < 0.f || alpha < GSR_ALPHA_MIN);
      const bool go = next_T > GSR_T_EPS;
      const bool draw = hit && go;
      const float dead = __uint_as_float(__float_as_uint(Tp) | 0x80000000u);  // -|T|
      const float upd = go ? next_T : dead;
      const float vis = draw ? alpha * Tp : 0.f;
      cr += B.r * vis;
      cg += B.g * vis;
      cb += C.blue * vis;
      if constexpr (RGBD) ce += C.extra * vis;
      last = draw ? C.sidx : last;
      T = hit ? upd : Tp;
    }
    __syncthreads();
    live = __any(T > 0.f);
  }
  if (mine) {
    const float Tp = fabsf(T);
    final_Ts[pid] = Tp;
    if (out_alpha) out_alpha[pid] = 1.f - Tp;
    final_idx[pid] = last;
    out_img[3 * pid] = cr + Tp * background[0];
    out_img[3 * pid + 1] = cg + Tp * background[1];
    out_img[3 * pid + 2] = cb + Tp * background[2];
    if constexpr (RGBD) out_extra[pid] = ce + Tp * bg_extra;
  }
}

// ------------------------------------------------------------- scan mapping
// north_star's other mapping ("wave-64 prefix-scan for per-pixel transmittance"), built to be
// MEASURED against the serial walk above: lanes run over the 64 staged
// SPLATS, the wave takes the pixels of its 8x8 sub-tile one at a time.  Per (pixel, chunk): every
// lane evaluates its splat's alpha at that pixel, a multiplicative DPP prefix scan of (1 - alpha)
// gives every splat its transmittance, the termination rule becomes one ballot (T falls
// monotonically, so "next_T <= 1e-4" holds for every hit behind the first one that fails), and
// the colour is a wave-wide sum.  One workgroup per (tile, sub-tile); a pixel's state sits in the
// lane of the same number.  Same staging, same sigma / alpha expressions as the serial kernel; T is
// a tree-ordered product instead of a sequential one, so results agree to rounding, not bitwise.
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143
#define DPP_WAVE_SHR1 0x138
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_or(float identity, float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(identity), __float_as_uint(v), CTRL, ROWMASK,
                                                     0xf, false));
}
__device__ __forceinline__ float wave_prefix_product(float v) {  // inclusive
  v *= dpp_or<DPP_ROW_SHR(1), 0xf>(1.f, v);
  v *= dpp_or<DPP_ROW_SHR(2), 0xf>(1.f, v);
  v *= dpp_or<DPP_ROW_SHR(4), 0xf>(1.f, v);
  v *= dpp_or<DPP_ROW_SHR(8), 0xf>(1.f, v);
  v *= dpp_or<DPP_ROW_BCAST15, 0xa>(1.f, v);
  v *= dpp_or<DPP_ROW_BCAST31, 0xc>(1.f, v);
  return v;
}
__device__ __forceinline__ float wave_total(float v) {  // the sum, valid in lane 63
  v += dpp_or<DPP_ROW_SHR(1), 0xf>(0.f, v);
  v += dpp_or<DPP_ROW_SHR(2), 0xf>(0.f, v);
  v += dpp_or<DPP_ROW_SHR(4), 0xf>(0.f, v);
  v += dpp_or<DPP_ROW_SHR(8), 0xf>(0.f, v);
  v += dpp_or<DPP_ROW_BCAST15, 0xa>(0.f, v);
  v += dpp_or<DPP_ROW_BCAST31, 0xc>(0.f, v);
  return v;
}
__device__ __forceinline__ float lane_value(float v, int l) {
  return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), l));
}

__global__ __launch_bounds__(64) void raster_fwd_scan_kernel(
    const int tiles_x, const int num_tiles, const int img_w, const int img_h,
    const int *__restrict__ ids_sorted, const int2 *__restrict__ tile_bins,
    const float2 *__restrict__ xys, const float *__restrict__ conics,
    const float *__restrict__ colors, const float *__restrict__ opacities,
    const float *__restrict__ background, float *__restrict__ out_img,
    float *__restrict__ final_Ts, int *__restrict__ final_idx, const unsigned base_grid) {
  __shared__ SplatA sA[kChunk];
  __shared__ SplatB sB[kChunk];
  __shared__ SplatC sC[kChunk];
  const int p = blockIdx.x / base_grid;  // sub-tile; blocks b and b + k base_grid share an XCD
  const int tile = gsr_xcd_remap(blockIdx.x % base_grid, tiles_x, num_tiles / tiles_x);
  if (tile < 0) return;
  const int2 range = tile_bins[tile];
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int lane = threadIdx.x;
  const int sx0 = tx * 16 + 8 * (p & 1), sy0 = ty * 16 + 8 * (p >> 1);
  const int col = sx0 + (lane & 7), row = sy0 + (lane >> 3);
  const bool inside = col < img_w && row < img_h;
  float T = inside ? 1.f : -1.f, cr = 0.f, cg = 0.f, cb = 0.f;
  int last = 0;
  unsigned long long live = __ballot(T > 0.f);
  for (int base = range.x; base < range.y && live != 0; base += kChunk) {
    const int sidx = base + lane;
    const int count = stage_chunk(lane, sidx < range.y, sidx, (float)(tx * 16), (float)(ty * 16), ids_sorted, xys,
                                  conics, colors, opacities, sA, sB, sC, nullptr, nullptr, nullptr, 1 << p);
    __syncthreads();
    const bool have = lane < count;
    const SplatA A = sA[have ? lane : 0];
    const SplatB B = sB[have ? lane : 0];
    const SplatC C = sC[have ? lane : 0];
    unsigned long long todo = count > 0 ? live : 0ull;
    while (todo) {
      const int q = __builtin_ctzll(todo);
      todo &= todo - 1;
      const float fx = (float)(sx0 + (q & 7)), fy = (float)(sy0 + (q >> 3));
      const float Tq = lane_value(T, q);
      const float dx = A.x - fx, dy = A.y - fy;
      const float sigma = (A.ha * dx * dx + B.hc * dy * dy) + (A.b * dx) * dy;
      const float alpha = fminf(GSR_ALPHA_MAX_FWD, B.opac * __expf(-sigma));
      const bool hit = have && !(sigma < 0.f || alpha < GSR_ALPHA_MIN);
      const float incl = wave_prefix_product(hit ? 1.f - alpha : 1.f);
      const float excl = dpp_or<DPP_WAVE_SHR1, 0xf>(1.f, incl);
      const float Tb = Tq * excl, next_T = Tq * incl;
      const bool go = next_T > GSR_T_EPS;
      const bool draw = hit && go;
      const float vis = draw ? alpha * Tb : 0.f;
      const float sr = wave_total(B.r * vis), sg = wave_total(B.g * vis), sb = wave_total(C.blue * vis);
      const unsigned long long fail = __ballot(hit && !go), drawn = __ballot(draw);
      float newT;
      if (fail) {  // the pixel is finished in front of the first hit that fails: keep T there, negated
        newT = -lane_value(Tb, __builtin_ctzll(fail));
        live &= ~(1ull << q);
      } else {
        newT = lane_value(next_T, 63);
      }
      const int lastq = drawn ? __builtin_amdgcn_readlane(C.sidx, 63 - __builtin_clzll(drawn)) : -1;
      // (read lane 63 under the FULL exec mask: inside the one-lane branch below the compiler sinks the
      //  reduction's last add into it, and lane 63's value is then never computed)
      const float tr = lane_value(sr, 63), tg = lane_value(sg, 63), tbl = lane_value(sb, 63);
      const bool mine = lane == q;
      T = mine ? newT : T;
      cr += mine ? tr : 0.f;
      cg += mine ? tg : 0.f;
      cb += mine ? tbl : 0.f;
      last = (mine && lastq >= 0) ? lastq : last;
    }
    __syncthreads();
  }
  if (inside) {
    const size_t pid = (size_t)row * img_w + col;
    const float Tp = fabsf(T);
    final_Ts[pid] = Tp;
    final_idx[pid] = last;
    out_img[3 * pid] = cr + Tp * background[0];
    out_img[3 * pid + 1] = cg + Tp * background[1];
    out_img[3 * pid + 2] = cb + Tp * background[2];
  }
}

// ----------------------------------------------------------------- generic
// One lane per pixel, bw*bw lanes per tile, batches of bw*bw splats in LDS.
// `channels` (<= CMAX) of the `cstride` interleaved channels starting at the pointers given:
// more than 32 channels are composited 32 at a time (the walk, T and final_idx are the same in
// every pass; the reference holds all channels of its 256 pixels in shared memory as __half and
// stops at what fits, ~90).
template <int CMAX>
__global__ __launch_bounds__(256) void raster_fwd_generic_kernel(
    const int tiles_x, const int img_w, const int img_h, const int channels, const int cstride,
    const int *__restrict__ ids_sorted, const int2 *__restrict__ tile_bins,
    const float2 *__restrict__ xys, const float *__restrict__ conics,
    const float *__restrict__ colors, const float *__restrict__ opacities,
    const float *__restrict__ background, float *__restrict__ out_img,
    float *__restrict__ final_Ts, int *__restrict__ final_idx) {
  __shared__ int s_id[256];
  __shared__ float s_x[256], s_y[256], s_o[256], s_a[256], s_b[256], s_c[256];

  const int bsize = blockDim.x * blockDim.y;
  const int tr = threadIdx.y * blockDim.x + threadIdx.x;
  const int tile = blockIdx.y * tiles_x + blockIdx.x;
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = blockIdx.y * blockDim.y + threadIdx.y;
  const float px = (float)col, py = (float)row;
  const bool inside = col < img_w && row < img_h;
  bool done = !inside;

  float acc[CMAX];
#pragma unroll
  for (int c = 0; c < CMAX; ++c) acc[c] = 0.f;
  float T = 1.f;
  int last = 0;

  const int2 range = tile_bins[tile];
  for (int base = range.x; base < range.y; base += bsize) {
    if (__syncthreads_count(done ? 1 : 0) >= bsize) break;
    const int idx = base + tr;
    if (idx < range.y) {
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
    __syncthreads();
    const int count = min(bsize, range.y - base);
    for (int t = 0; t < count && !done; ++t) {
      const float dx = s_x[t] - px, dy = s_y[t] - py;
      const float sigma = 0.5f * (s_a[t] * dx * dx + s_c[t] * dy * dy) + s_b[t] * dx * dy;
      const float alpha = fminf(GSR_ALPHA_MAX_FWD, s_o[t] * __expf(-sigma));
      if (sigma < 0.f || alpha < GSR_ALPHA_MIN) continue;
      const float next_T = T * (1.f - alpha);
      if (next_T <= GSR_T_EPS) {
        done = true;
        break;
      }
      const float vis = alpha * T;
      const float *col_g = colors + (size_t)s_id[t] * cstride;
#pragma unroll
      for (int c = 0; c < CMAX; ++c)
        if (c < channels) acc[c] += col_g[c] * vis;
      T = next_T;
      last = base + t;
    }
  }

  if (inside) {
    const size_t pid = (size_t)row * img_w + col;
    final_Ts[pid] = T;
    final_idx[pid] = last;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < channels) out_img[pid * cstride + c] = acc[c] + T * background[c];
  }
}

int launch_generic(int tiles_x, int tiles_y, unsigned bw, unsigned img_w, unsigned img_h,
                   unsigned channels, const int32_t *ids, const int32_t *bins, const float *xys,
                   const float *conics, const float *colors, const float *opac,
                   const float *background, float *out_img, float *final_Ts, int32_t *final_idx,
                   hipStream_t s) {
  const dim3 grd(tiles_x, tiles_y), blk(bw, bw);
#define GSR_LAUNCH_FWD(CM, C0, CN)                                                              \
  hipLaunchKernelGGL(raster_fwd_generic_kernel<CM>, grd, blk, 0, s, tiles_x, (int)img_w,        \
                     (int)img_h, (int)(CN), (int)channels, ids, reinterpret_cast<const int2 *>(bins), \
                     reinterpret_cast<const float2 *>(xys), conics, colors + (C0), opac,        \
                     background + (C0), out_img + (C0), final_Ts, final_idx)
  if (channels <= 4) GSR_LAUNCH_FWD(4, 0, channels);
  else if (channels <= 8) GSR_LAUNCH_FWD(8, 0, channels);
  else if (channels <= 16) GSR_LAUNCH_FWD(16, 0, channels);
  else
    for (unsigned c0 = 0; c0 < channels; c0 += 32) GSR_LAUNCH_FWD(32, c0, std::min(32u, channels - c0));
#undef GSR_LAUNCH_FWD
  GSR_CHECK_LAUNCH("rasterize_forward(generic)");
  return GSR_OK;
}

int check_common(const char *who, int tiles_x, int tiles_y, unsigned bw, unsigned img_w,
                 unsigned img_h, unsigned channels) {
  GSR_REQUIRE(bw >= 2 && bw <= 16, "%s: block_width must be in [2,16]", who);
  GSR_REQUIRE(img_w > 0 && img_h > 0, "%s: empty image", who);
  GSR_REQUIRE(tiles_x == (int)gsr_cdiv(img_w, bw) && tiles_y == (int)gsr_cdiv(img_h, bw),
              "%s: tile bounds (%d,%d) do not match image %ux%u / block %u", who, tiles_x, tiles_y,
              img_w, img_h, bw);
  GSR_REQUIRE(channels >= 1 && channels <= GSR_MAX_CHANNELS, "%s: channels must be in [1,%d]", who,
              GSR_MAX_CHANNELS);
  return GSR_OK;
}

}  // namespace

GSR_EXPORT int gsr_rasterize_forward_nd(int tiles_x, int tiles_y, unsigned block_width,
                                        unsigned img_width, unsigned img_height, unsigned channels,
                                        const int32_t *gaussian_ids_sorted,
                                        const int32_t *tile_bins, const float *xys,
                                        const float *conics, const float *colors,
                                        const float *opacities, const float *background,
                                        float *out_img, float *final_Ts, int32_t *final_idx,
                                        gsr_stream_t stream) {
  int rc = check_common("rasterize_forward", tiles_x, tiles_y, block_width, img_width, img_height,
                        channels);
  if (rc != GSR_OK) return rc;
  GSR_REQUIRE(gaussian_ids_sorted && tile_bins && xys && conics && colors && opacities &&
                  background && out_img && final_Ts && final_idx,
              "rasterize_forward: null pointer");
  return launch_generic(tiles_x, tiles_y, block_width, img_width, img_height, channels,
                        gaussian_ids_sorted, tile_bins, xys, conics, colors, opacities, background,
                        out_img, final_Ts, final_idx, (hipStream_t)stream);
}

GSR_EXPORT int gsr_rasterize_forward(int tiles_x, int tiles_y, unsigned block_width,
                                     unsigned img_width, unsigned img_height,
                                     const int32_t *gaussian_ids_sorted, const int32_t *tile_bins,
                                     const float *xys, const float *conics, const float *colors,
                                     const float *opacities, const float *background,
                                     float *out_img, float *final_Ts, int32_t *final_idx,
                                     int deep_tile_threshold, gsr_stream_t stream) {
  return gsr_rasterize_forward_ex(tiles_x, tiles_y, block_width, img_width, img_height, gaussian_ids_sorted,
                                  tile_bins, xys, conics, colors, opacities, background, out_img, final_Ts,
                                  final_idx, deep_tile_threshold, nullptr, nullptr, 0, stream);
}

GSR_EXPORT int gsr_rasterize_forward_ex(int tiles_x, int tiles_y, unsigned block_width,
                                        unsigned img_width, unsigned img_height,
                                        const int32_t *gaussian_ids_sorted, const int32_t *tile_bins,
                                        const float *xys, const float *conics, const float *colors,
                                        const float *opacities, const float *background,
                                        float *out_img, float *final_Ts, int32_t *final_idx,
                                        int deep_tile_threshold, float *out_alpha, void *zero_ptr,
                                        size_t zero_bytes, gsr_stream_t stream) {
  int rc = check_common("rasterize_forward", tiles_x, tiles_y, block_width, img_width, img_height, 3);
  if (rc != GSR_OK) return rc;
  GSR_REQUIRE(gaussian_ids_sorted && tile_bins && xys && conics && colors && opacities &&
                  background && out_img && final_Ts && final_idx,
              "rasterize_forward: null pointer");
  GSR_REQUIRE((out_alpha == nullptr && zero_ptr == nullptr) || block_width == 16,
              "rasterize_forward_ex: out_alpha / zero_ptr need block_width 16");
  GSR_REQUIRE(zero_ptr == nullptr || ((zero_bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(zero_ptr) & 3) == 0 &&
                                      zero_bytes < ((size_t)1 << 34)),
              "rasterize_forward_ex: zero_ptr / zero_bytes must be multiples of 4 (and below 16 GB)");
  if (block_width != 16)
    return launch_generic(tiles_x, tiles_y, block_width, img_width, img_height, 3,
                          gaussian_ids_sorted, tile_bins, xys, conics, colors, opacities,
                          background, out_img, final_Ts, final_idx, (hipStream_t)stream);
  const int num_tiles = tiles_x * tiles_y;
  const unsigned base = gsr_xcd_grid(tiles_x, num_tiles / tiles_x);
  const int deep = gsr_prepare_jobs(deep_tile_threshold, tiles_x, tiles_y, tile_bins, (hipStream_t)stream);
  hipLaunchKernelGGL(raster_fwd_tile16_kernel<false>, dim3(deep ? 4 * base : base), dim3(64), 0,
                     (hipStream_t)stream, tiles_x, num_tiles, (int)img_width, (int)img_height, gaussian_ids_sorted,
                     reinterpret_cast<const int2 *>(tile_bins),
                     reinterpret_cast<const float2 *>(xys), conics, colors, opacities, background,
                     out_img, final_Ts, final_idx, (const float *)nullptr, 0.f, (float *)nullptr, deep, base,
                     out_alpha, static_cast<unsigned *>(zero_ptr), (unsigned)(zero_bytes >> 2), 0, (int *)nullptr, 0);
  GSR_CHECK_LAUNCH("rasterize_forward(tile16)");
  return GSR_OK;
}

GSR_EXPORT size_t gsr_rasterize_forward_seg_workspace_bytes(unsigned img_height, unsigned img_width, int segments) {
  if (segments < 2) return 0;
  const size_t px = (size_t)img_height * img_width;
  const size_t tiles = (size_t)gsr_cdiv(img_width, 16) * gsr_cdiv(img_height, 16);
  // per run: raw states (float4) | last drawn indices | the extra channel's sums; then per pixel its crossing run,
  // per (tile, sub-tile) the mask of runs to re-walk and the 16 saturation marks
  return (size_t)segments * px * (16 + 4 + 4) + px * 4 + tiles * 16 + tiles * 4 * 16 * 4;
}

GSR_EXPORT int gsr_rasterize_forward_seg(int tiles_x, int tiles_y, unsigned img_width, unsigned img_height,
                                         const int32_t *gaussian_ids_sorted, const int32_t *tile_bins,
                                         const float *xys, const float *conics, const float *colors,
                                         const float *extra, const float *opacities, const float *background,
                                         float extra_background, float *out_img, float *out_extra, float *final_Ts,
                                         int32_t *final_idx, int deep_tile_threshold, float *out_alpha, void *zero_ptr,
                                         size_t zero_bytes, int segments, int segment_min_entries, void *workspace,
                                         size_t workspace_bytes, gsr_stream_t stream) {
  if (segments < 2 || deep_tile_threshold <= 0) {
    if (extra)
      return gsr_rasterize_forward_rgbd(tiles_x, tiles_y, img_width, img_height, gaussian_ids_sorted, tile_bins, xys,
                                        conics, colors, extra, opacities, background, extra_background, out_img,
                                        out_extra, final_Ts, final_idx, deep_tile_threshold, out_alpha, zero_ptr,
                                        zero_bytes, stream);
    return gsr_rasterize_forward_ex(tiles_x, tiles_y, 16, img_width, img_height, gaussian_ids_sorted, tile_bins, xys,
                                    conics, colors, opacities, background, out_img, final_Ts, final_idx,
                                    deep_tile_threshold, out_alpha, zero_ptr, zero_bytes, stream);
  }
  int rc = check_common("rasterize_forward_seg", tiles_x, tiles_y, 16, img_width, img_height, 3);
  if (rc != GSR_OK) return rc;
  GSR_REQUIRE(gaussian_ids_sorted && tile_bins && xys && conics && colors && opacities && background && out_img &&
                  final_Ts && final_idx,
              "rasterize_forward_seg: null pointer");
  GSR_REQUIRE((extra == nullptr) == (out_extra == nullptr), "rasterize_forward_seg: extra and out_extra go together");
  GSR_REQUIRE(segments <= 16, "rasterize_forward_seg: at most 16 segments");
  GSR_REQUIRE(zero_ptr == nullptr || ((zero_bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(zero_ptr) & 3) == 0 &&
                                      zero_bytes < ((size_t)1 << 34)),
              "rasterize_forward_seg: zero_ptr / zero_bytes must be multiples of 4 (and below 16 GB)");
  GSR_REQUIRE(workspace && workspace_bytes >= gsr_rasterize_forward_seg_workspace_bytes(img_height, img_width, segments) &&
                  (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
              "rasterize_forward_seg: workspace too small or not 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int num_tiles = tiles_x * tiles_y;
  const unsigned base = gsr_xcd_grid(tiles_x, tiles_y);
  const size_t px = (size_t)img_height * img_width;
  const int deep_arg = gsr_prepare_jobs(deep_tile_threshold, tiles_x, tiles_y, tile_bins, s);  // (threshold | order flag)
  deep_tile_threshold = gsr_deep_threshold(deep_tile_threshold);
  const int seg_min = segment_min_entries > deep_tile_threshold ? segment_min_entries : deep_tile_threshold;
  char *ws = static_cast<char *>(workspace);
  float4 *raw = reinterpret_cast<float4 *>(ws);
  int *lastp = reinterpret_cast<int *>(ws + (size_t)segments * px * 16);
  float *extrap = reinterpret_cast<float *>(lastp + (size_t)segments * px);  // (written and read with `extra` only)
  int *kstarp = reinterpret_cast<int *>(extrap + (size_t)segments * px);
  int *flagsp = kstarp + px;
  float *marksp = reinterpret_cast<float *>(flagsp + 4 * (size_t)num_tiles);
  hipLaunchKernelGGL(seg_marks_clear_kernel, dim3((unsigned)((64 * (size_t)num_tiles + 255) / 256)), dim3(256), 0, s, marksp,
                     64u * (unsigned)num_tiles);
#define GSR_LAUNCH_FWD_SEG(RGBD_)                                                                                       \
  hipLaunchKernelGGL((raster_fwd_tile16_kernel<RGBD_, true>), dim3((unsigned)segments * 4u * base), dim3(64), 0, s,      \
                     tiles_x, num_tiles, (int)img_width, (int)img_height, gaussian_ids_sorted,                          \
                     reinterpret_cast<const int2 *>(tile_bins), reinterpret_cast<const float2 *>(xys), conics, colors,   \
                     opacities, background, out_img, final_Ts, final_idx, extra, extra_background, out_extra,           \
                     deep_arg, base, out_alpha, static_cast<unsigned *>(zero_ptr),                                       \
                     (unsigned)(zero_bytes >> 2), 0, (int *)nullptr, 0, segments, seg_min, raw, lastp, extrap, marksp);  \
  hipLaunchKernelGGL(raster_fwd_segresolve_kernel<RGBD_>, dim3(4u * base), dim3(64), 0, s, tiles_x, num_tiles,          \
                     (int)img_width, (int)img_height, reinterpret_cast<const int2 *>(tile_bins), background, out_img,    \
                     final_Ts, final_idx, extra_background, out_extra, deep_arg, base, out_alpha, segments, seg_min,    \
                     raw, lastp, extrap, kstarp, flagsp);                                                                \
  hipLaunchKernelGGL(raster_fwd_segrewalk_kernel<RGBD_>, dim3((unsigned)segments * 4u * base), dim3(64), 0, s,          \
                     tiles_x, num_tiles, (int)img_width, (int)img_height, gaussian_ids_sorted,                          \
                     reinterpret_cast<const int2 *>(tile_bins), reinterpret_cast<const float2 *>(xys), conics, colors,   \
                     opacities, background, out_img, final_Ts, final_idx, extra, extra_background, out_extra, deep_arg, \
                     base, out_alpha, segments, seg_min, (const float4 *)raw, (const int *)lastp,                       \
                     (const float *)extrap, (const int *)kstarp, (const int *)flagsp)
  if (extra) {
    GSR_LAUNCH_FWD_SEG(true);
  } else {
    GSR_LAUNCH_FWD_SEG(false);
  }
#undef GSR_LAUNCH_FWD_SEG
  GSR_CHECK_LAUNCH("rasterize_forward_seg");
  return GSR_OK;
}

GSR_EXPORT int gsr_rasterize_forward_scan(int tiles_x, int tiles_y, unsigned img_width, unsigned img_height,
                                          const int32_t *gaussian_ids_sorted, const int32_t *tile_bins,
                                          const float *xys, const float *conics, const float *colors,
                                          const float *opacities, const float *background, float *out_img,
                                          float *final_Ts, int32_t *final_idx, gsr_stream_t stream) {
  int rc = check_common("rasterize_forward_scan", tiles_x, tiles_y, 16, img_width, img_height, 3);
  if (rc != GSR_OK) return rc;
  GSR_REQUIRE(gaussian_ids_sorted && tile_bins && xys && conics && colors && opacities && background && out_img &&
                  final_Ts && final_idx,
              "rasterize_forward_scan: null pointer");
  const int num_tiles = tiles_x * tiles_y;
  const unsigned base = gsr_xcd_grid(tiles_x, tiles_y);
  hipLaunchKernelGGL(raster_fwd_scan_kernel, dim3(4 * base), dim3(64), 0, (hipStream_t)stream, tiles_x, num_tiles,
                     (int)img_width, (int)img_height, gaussian_ids_sorted, reinterpret_cast<const int2 *>(tile_bins),
                     reinterpret_cast<const float2 *>(xys), conics, colors, opacities, background, out_img, final_Ts,
                     final_idx, base);
  GSR_CHECK_LAUNCH("rasterize_forward_scan");
  return GSR_OK;
}

GSR_EXPORT int gsr_rasterize_forward_rgbd(int tiles_x, int tiles_y, unsigned img_width, unsigned img_height,
                                          const int32_t *gaussian_ids_sorted, const int32_t *tile_bins,
                                          const float *xys, const float *conics, const float *colors,
                                          const float *extra, const float *opacities,
                                          const float *background, float extra_background, float *out_img,
                                          float *out_extra, float *final_Ts, int32_t *final_idx,
                                          int deep_tile_threshold, float *out_alpha, void *zero_ptr,
                                          size_t zero_bytes, gsr_stream_t stream) {
  int rc = check_common("rasterize_forward_rgbd", tiles_x, tiles_y, 16, img_width, img_height, 3);
  if (rc != GSR_OK) return rc;
  GSR_REQUIRE(gaussian_ids_sorted && tile_bins && xys && conics && colors && extra && opacities && background &&
                  out_img && out_extra && final_Ts && final_idx,
              "rasterize_forward_rgbd: null pointer");
  GSR_REQUIRE(zero_ptr == nullptr || ((zero_bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(zero_ptr) & 3) == 0 &&
                                      zero_bytes < ((size_t)1 << 34)),
              "rasterize_forward_rgbd: zero_ptr / zero_bytes must be multiples of 4 (and below 16 GB)");
  const int num_tiles = tiles_x * tiles_y;
  const unsigned base = gsr_xcd_grid(tiles_x, num_tiles / tiles_x);
  const int deep = gsr_prepare_jobs(deep_tile_threshold, tiles_x, tiles_y, tile_bins, (hipStream_t)stream);
  hipLaunchKernelGGL(raster_fwd_tile16_kernel<true>, dim3(deep ? 4 * base : base), dim3(64), 0,
                     (hipStream_t)stream, tiles_x, num_tiles, (int)img_width, (int)img_height, gaussian_ids_sorted,
                     reinterpret_cast<const int2 *>(tile_bins),
                     reinterpret_cast<const float2 *>(xys), conics, colors, opacities, background,
                     out_img, final_Ts, final_idx, extra, extra_background, out_extra, deep, base,
                     out_alpha, static_cast<unsigned *>(zero_ptr), (unsigned)(zero_bytes >> 2), 0, (int *)nullptr, 0);
  GSR_CHECK_LAUNCH("rasterize_forward_rgbd");
  return GSR_OK;
}

GSR_EXPORT int gsr_rasterize_forward_round(int round, int tiles_x, int tiles_y, unsigned img_width,
                                           unsigned img_height, const int32_t *gaussian_ids_sorted,
                                           const int32_t *tile_bins, int idx_base, const float *xys,
                                           const float *conics, const float *colors, const float *extra,
                                           const float *opacities, const float *background, float extra_background,
                                           float *out_img, float *out_extra, float *final_Ts, int32_t *final_idx,
                                           int32_t *tile_flags, int deep_tile_threshold, float *out_alpha,
                                           void *zero_ptr, size_t zero_bytes, gsr_stream_t stream) {
  int rc = check_common("rasterize_forward_round", tiles_x, tiles_y, 16, img_width, img_height, 3);
  if (rc != GSR_OK) return rc;
  GSR_REQUIRE(round == 1 || round == 2, "rasterize_forward_round: round must be 1 or 2");
  GSR_REQUIRE(gaussian_ids_sorted && tile_bins && xys && conics && colors && opacities && background && out_img &&
                  final_Ts && final_idx,
              "rasterize_forward_round: null pointer");
  GSR_REQUIRE((extra == nullptr) == (out_extra == nullptr), "rasterize_forward_round: extra and out_extra go together");
  GSR_REQUIRE(tile_flags != nullptr, "rasterize_forward_round: tile_flags is required");
  GSR_REQUIRE(idx_base >= 0, "rasterize_forward_round: idx_base < 0");
  GSR_REQUIRE(zero_ptr == nullptr || ((zero_bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(zero_ptr) & 3) == 0 &&
                                      zero_bytes < ((size_t)1 << 34)),
              "rasterize_forward_round: zero_ptr / zero_bytes must be multiples of 4 (and below 16 GB)");
  const int num_tiles = tiles_x * tiles_y;
  const unsigned base = gsr_xcd_grid(tiles_x, tiles_y);
  const int deep = gsr_deep_threshold(deep_tile_threshold);  // (two-round lists: the static block order)
  const dim3 grd(deep ? 4 * base : base), blk(64);
  if (extra)
    hipLaunchKernelGGL(raster_fwd_tile16_kernel<true>, grd, blk, 0, (hipStream_t)stream, tiles_x, num_tiles,
                       (int)img_width, (int)img_height, gaussian_ids_sorted, reinterpret_cast<const int2 *>(tile_bins),
                       reinterpret_cast<const float2 *>(xys), conics, colors, opacities, background, out_img, final_Ts,
                       final_idx, extra, extra_background, out_extra, deep, base, out_alpha,
                       static_cast<unsigned *>(zero_ptr), (unsigned)(zero_bytes >> 2), round, tile_flags, idx_base);
  else
    hipLaunchKernelGGL(raster_fwd_tile16_kernel<false>, grd, blk, 0, (hipStream_t)stream, tiles_x, num_tiles,
                       (int)img_width, (int)img_height, gaussian_ids_sorted, reinterpret_cast<const int2 *>(tile_bins),
                       reinterpret_cast<const float2 *>(xys), conics, colors, opacities, background, out_img, final_Ts,
                       final_idx, (const float *)nullptr, 0.f, (float *)nullptr, deep, base, out_alpha,
                       static_cast<unsigned *>(zero_ptr), (unsigned)(zero_bytes >> 2), round, tile_flags, idx_base);
  GSR_CHECK_LAUNCH("rasterize_forward_round");
  return GSR_OK;
}

GSR_EXPORT size_t gsr_tile_jobs_ints(int tiles_x, int tiles_y) {
  // two arrays of 4 base_grid + the address of the statistics; 0 = this grid has no job order (beyond the sort's
  // tables -- more than 4 096 tile slots per XCD, i.e. above 3840 x 2160: the launches run in the static order)
  if (tiles_x <= 0 || tiles_y <= 0) return 0;
  const unsigned base = gsr_xcd_grid(tiles_x, tiles_y);
  if (base / 8u > (unsigned)gsr::kJobChunks * 64u) return 0;
  return 8 * (size_t)base + 2;
}

GSR_EXPORT int gsr_tile_jobs_build(int tiles_x, int tiles_y, int32_t *tile_bins, int deep_arg_first, int deep_arg_second,
                                   gsr_stream_t stream) {
  GSR_REQUIRE(tiles_x > 0 && tiles_y > 0 && tile_bins, "tile_jobs_build: bad arguments");
  GSR_REQUIRE(gsr_deep_ordered(deep_arg_first) && (deep_arg_second <= 0 || gsr_deep_ordered(deep_arg_second)),
              "tile_jobs_build: arguments without GSR_DEEP_ORDERED");
  GSR_REQUIRE(deep_arg_second <= 0 || gsr_deep_second(deep_arg_first) != gsr_deep_second(deep_arg_second),
              "tile_jobs_build: both orders name the same array");
  const unsigned base = gsr_xcd_grid(tiles_x, tiles_y);
  // beyond the sort's tables (4096 x 2160 and larger): nothing to build -- gsr_prepare_jobs hands such a grid's launches
  // the plain threshold before it looks at GSR_DEEP_PREBUILT, and nobody reads the tail
  if (base / 8u > (unsigned)gsr::kJobChunks * 64u) return GSR_OK;
  int *jobs = tile_bins + 2 * (size_t)tiles_x * tiles_y;
  hipLaunchKernelGGL(gsr::tile_jobs_kernel, dim3(deep_arg_second > 0 ? 16 : 8), dim3(1024), 0, (hipStream_t)stream, tiles_x,
                     tiles_y, base, reinterpret_cast<const int2 *>(tile_bins), deep_arg_first, deep_arg_second, jobs,
                     gsr::gsr_job_stats_buffer((hipStream_t)stream), gsr::gsr_job_split_ratio());
  GSR_CHECK_LAUNCH("tile_jobs_build");
  return GSR_OK;
}

// internal: see gsr_debug_wave_trace (raster_bwd.hip)
int gsr_set_fwd_wave_trace(unsigned long long *buf, unsigned capacity) {
  gsr::WaveTrace t = {buf, capacity};
  GSR_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_fwd_trace), &t, sizeof(t)));
  return GSR_OK;
}

// internal: see gsr_debug_count_staged (raster_bwd.hip)
int gsr_set_fwd_staged_counter(unsigned long long *counter) {
  GSR_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_fwd_staged), &counter, sizeof(counter)));
  return GSR_OK;
}
