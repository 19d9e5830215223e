// raster_common.h -- pieces shared by the two wave-per-tile compositing kernels.
#pragma once
#include "gsr_common.h"

namespace gsr {

constexpr int kChunk = 64;  // splats staged per LDS fill (one per lane)

// LDS record of one staged splat; consumed by wave-uniform broadcast reads.
struct __align__(16) SplatA { float x, y, ha, b; };     // ha = a/2
struct __align__(16) SplatB { float hc, opac, r, g; };  // hc = c/2
struct __align__(16) SplatC { float blue; int sidx; int mask; float extra; };  // extra: optional 4th channel
// (0.5*(a dx^2 + c dy^2) == (a/2) dx^2 + (c/2) dy^2 exactly: scaling by a power
//  of two commutes with rounding)
// sidx = index in the sorted list; mask = which of the tile's four 8x8
// sub-tiles the splat can reach (bit p <-> sub-tile (p&1, p>>1)).

// Pixel ownership inside a 16x16 tile: lane l sits at (l&7, l>>3) of EVERY 8x8
// sub-tile, i.e. it owns the 4 pixels (l&7 + 8*(p&1), (l>>3) + 8*(p>>1)),
// p = 0..3.  One wave-instruction therefore covers a whole sub-tile, and a
// sub-tile a splat cannot reach is skipped with a scalar branch.

// Can ANY pixel centre of the (w+1) x (w+1) pixel square whose first pixel is
// (rx0, ry0) get alpha = opac*exp(-sigma) >= 1/255 from this splat?  The tile
// lists of the reference are built from the 3-sigma *square* bounding box
// (forward.cu:73): about half of a tile's entries never pass the alpha test at
// any of its pixels, and of the rest ~1/3 of the (splat, sub-tile) pairs don't
// either.  Dropping them cannot change a result: the compositing rule skips them
// pixel by pixel (alpha < 1/255 -> continue).  The test is conservative: sigma is
// minimised over the continuous rectangle (<= its minimum over the pixel
// centres) and `smax` carries a 1 % margin in alpha, far above fp32 rounding.
struct Reach {
  float x, y, a, b, c;
  float nb_c, nb_a;  // -b/c, -b/a
  float smax;        // sigma bound: +inf = always keep, < 0 = can never reach
};

__device__ __forceinline__ Reach make_reach(float x, float y, float a, float b, float c, float opac) {
  Reach r{x, y, a, b, c, 0.f, 0.f, 0.f};
  // alpha >= 1/255  <=>  sigma <= log(255*opac)
  r.smax = __logf(255.f * opac) + 0.01f;
  if (!(r.smax >= 0.f)) r.smax = (opac == opac) ? -1.f : INFINITY;  // too faint anywhere (NaN: keep)
  if (!(a > 0.f && c > 0.f && a * c - b * b > 0.f)) {
    r.smax = INFINITY;  // not positive definite: no culling
  } else {
    r.nb_c = -b / c;
    r.nb_a = -b / a;
  }
  return r;
}

__device__ __forceinline__ bool reaches_rect(const Reach &r, float rx0, float ry0, float w) {
  if (r.smax == INFINITY) return true;
  if (r.smax < 0.f) return false;
  const float u0 = rx0 - r.x, u1 = u0 + w, v0 = ry0 - r.y, v1 = v0 + w;
  if (u0 <= 0.f && u1 >= 0.f && v0 <= 0.f && v1 >= 0.f) return true;  // centre inside
  // convex quadratic, unconstrained minimum (0,0) outside the rectangle ->
  // the minimum over the rectangle lies on one of its four edges
  auto edge_u = [&](float ue) {
    const float v = fminf(fmaxf(r.nb_c * ue, v0), v1);
    return 0.5f * (r.a * ue * ue + r.c * v * v) + r.b * ue * v;
  };
  auto edge_v = [&](float ve) {
    const float u = fminf(fmaxf(r.nb_a * ve, u0), u1);
    return 0.5f * (r.a * u * u + r.c * ve * ve) + r.b * u * ve;
  };
  const float smin = fminf(fminf(edge_u(u0), edge_u(u1)), fminf(edge_v(v0), edge_v(v1)));
  return smin <= r.smax;
}

// 4-bit reach mask over the sub-tiles of the tile at (tx0, ty0); 0 = drop.
__device__ __forceinline__ int splat_reach_mask(float x, float y, float a, float b, float c,
                                                float opac, float tx0, float ty0) {
  const Reach r = make_reach(x, y, a, b, c, opac);
  int m = 0;
#pragma unroll
  for (int p = 0; p < 4; ++p)
    m |= reaches_rect(r, tx0 + 8.f * (p & 1), ty0 + 8.f * (p >> 1), 7.f) ? (1 << p) : 0;
  return m;
}

// ---- deep tiles ---------------------------------------------------------------
// One wave per tile makes a kernel as slow as its deepest tiles: a tile whose list is
// several times the average keeps one wave busy long after the rest of the chip has
// drained (clustered scenes: 10 % of the tiles with 10x the depth).  With
// deep_threshold > 0 the launch carries three more blocks per tile slot; for a tile
// whose list is longer than the threshold the four blocks each take ONE 8x8 sub-tile
// (the regular block sub-tile 0), for every other tile the extra blocks exit at once.
// A sub-tile wave runs the very same per-pixel instructions on its 64 pixels (the other
// three sub-tiles are masked off wave-uniformly) and stages only the splats that reach
// its sub-tile, so results are bit-identical to the one-wave-per-tile walk.
// Block b of the extra range [base_grid, 4 base_grid) sits on XCD b % 8 like the regular
// block of the same tile (the four waves share the splat records in one L2).
// ORDER (round 5): the hardware starts workgroups in block order, and a deep tile is the longest job of the launch.
// With the extra blocks BEHIND the regular ones (rounds 2-4) three quarters of every deep tile's work were the last
// waves to start -- the opposite of longest-job-first.  The extra range now comes first: blocks [0, 3 base_grid) are
// the extra sub-tile waves (the ones of shallow tiles exit at once), blocks [3 base_grid, 4 base_grid) the regular
// ones.  The XCD of a block is unchanged (base_grid is a multiple of 8).  GSR_DEEP_EXTRAS_FIRST=0 builds the old order.
#ifndef GSR_DEEP_EXTRAS_FIRST
#define GSR_DEEP_EXTRAS_FIRST 1
#endif
struct TileJob {
  int tile;     // < 0: nothing to do
  int allowed;  // sub-tile mask this wave owns (15 = the whole tile)
};
__device__ __forceinline__ TileJob tile_job(const unsigned b_launch, const unsigned base_grid, const int tiles_x,
                                            const int tiles_y, const int2 *__restrict__ tile_bins,
                                            const int deep_threshold, int2 &range) {
  TileJob j{-1, 15};
  unsigned b = b_launch;
  if (GSR_DEEP_EXTRAS_FIRST && deep_threshold > 0) b = b_launch < 3u * base_grid ? b_launch + base_grid : b_launch - 3u * base_grid;
  if (b < base_grid) {
    j.tile = gsr_xcd_remap(b, tiles_x, tiles_y);
    if (j.tile < 0) return j;
    range = tile_bins[j.tile];
    if (deep_threshold > 0 && range.y - range.x > deep_threshold) j.allowed = 1;
    return j;
  }
  const unsigned e = b - base_grid, x = e & 7u, q = e >> 3;
  const int tile = gsr_xcd_remap((q / 3u) * 8u + x, tiles_x, tiles_y);
  if (tile < 0) return j;
  range = tile_bins[tile];
  if (range.y - range.x <= deep_threshold) return j;
  j.tile = tile;
  j.allowed = 2 << (q % 3u);
  return j;
}

// Depth segments (raster_fwd.hip / raster_bwd.hip): a split tile's list of `len` entries is cut into at most K runs of
// whole 64-entry chunks; run k is [k seg_len, (k + 1) seg_len) of the list.
__device__ __forceinline__ int seg_len_of(const int len, const int K) { return ((len + K * 64 - 1) / (K * 64)) * 64; }

// ---- measurement hook (gsr_debug_wave_trace): one 32-byte record per wave that ran a tile to its end ------------
// {constant 100-MHz clock at entry, at exit, tile | allowed << 32, list length | hw id << 32}: who ran when and for
// how long -- the tail / imbalance of a launch (tools/exp/wave_trace.py).  hw id = HW_REG_HW_ID (SIMD [5:4], CU [11:8],
// SH [12], SE [15:13]) | XCC_ID << 20.
struct WaveTrace {
  unsigned long long *buf;
  unsigned capacity;
};
__device__ __forceinline__ unsigned long long trace_begin(const WaveTrace &tr) {
  return tr.buf ? wall_clock64() : 0ull;
}
__device__ __forceinline__ void trace_end(const WaveTrace &tr, const unsigned long long t0, const int tile,
                                          const int allowed, const int len) {
  if (tr.buf && blockIdx.x < tr.capacity && threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
    unsigned long long *r = tr.buf + 4ull * blockIdx.x;
    r[0] = t0;
    r[1] = wall_clock64();
    r[2] = (unsigned long long)(unsigned)tile | ((unsigned long long)(unsigned)allowed << 32);
    r[3] = (unsigned long long)(unsigned)len | ((unsigned long long)((hw & 0xffffu) | (xcc << 20)) << 32);
  }
}

// Stage up to 64 splats (one per `live` lane, sorted index `sidx`) into LDS,
// dropping the ones that cannot reach the tile; returns how many were kept.
// Kept splats stay in lane order.  Wave-synchronous (one wave per workgroup).
__device__ __forceinline__ int stage_chunk(
    const int lane, const bool live, const int sidx, const float tx0, const float ty0,
    const int *__restrict__ ids_sorted, const float2 *__restrict__ xys,
    const float *__restrict__ conics, const float *__restrict__ colors,
    const float *__restrict__ opacities, SplatA *sA, SplatB *sB, SplatC *sC, int *sId,
    const float *__restrict__ extra = nullptr, unsigned long long *staged_counter = nullptr,
    const int allowed = 15) {
  // measurement hook (gsr_debug_count_staged): list entries read by this launch
  if (staged_counter) {
    const int n = __popcll(__ballot(live));
    if (lane == 0) atomicAdd(staged_counter, (unsigned long long)n);
  }
  int mask = 0;
  int g = 0;
  float2 xy = make_float2(0.f, 0.f);
  float a = 0.f, b = 0.f, c = 0.f, opac = 0.f;
  if (live) {
    g = ids_sorted[sidx];
    xy = xys[g];
    a = conics[3 * g];
    b = conics[3 * g + 1];
    c = conics[3 * g + 2];
    opac = opacities[g];
    mask = splat_reach_mask(xy.x, xy.y, a, b, c, opac, tx0, ty0);
#ifdef GSR_NO_CULL
    mask = 15;  // experiment: keep every list entry
#endif
    mask &= allowed;  // a wave that owns one sub-tile of a deep tile keeps only what reaches it
  }
  const unsigned long long kept = __ballot(mask != 0);
  if (mask != 0) {
    const int slot = __builtin_amdgcn_mbcnt_hi((unsigned)(kept >> 32),
                                               __builtin_amdgcn_mbcnt_lo((unsigned)kept, 0u));
    sA[slot] = SplatA{xy.x, xy.y, 0.5f * a, b};
    sB[slot] = SplatB{0.5f * c, opac, colors[3 * g], colors[3 * g + 1]};
    sC[slot] = SplatC{colors[3 * g + 2], sidx, mask, extra ? extra[g] : 0.f};
    if (sId) sId[slot] = g;
  }
  return __popcll(kept);
}

}  // namespace gsr
