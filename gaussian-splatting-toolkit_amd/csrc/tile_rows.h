// tile_rows.h -- which tiles of its bounding box does a splat go into?  Shared by the
// list-building kernels (binning_fast.hip, tile_partition2.hip).
#pragma once
#include "gsr_common.h"
#include "raster_common.h"

namespace {

// ---- tile lists: which tiles of its bounding box does a splat go into? -----
// Per Gaussian record (32 B) written by the count pass in index order and
// gathered by the emission pass in depth order (one 32-B sector per Gaussian
// instead of four scattered loads).
struct alignas(16) SplatRec {
  float x, y, a, b, c;
  float smax;     // see raster_common.h make_reach(): +inf = keep every box tile, < 0 = none
  unsigned box0;  // minx | miny << 16
  unsigned box1;  // box width | box height << 16   (0 | 0 when culled)
};
static_assert(sizeof(SplatRec) == 32, "SplatRec layout");

// What a record is made from.  Loaded unconditionally (a load under `radius > 0` is a branch with its own wait: fine
// for one Gaussian per lane, a round trip per item where a lane handles several) or only when visible.
struct SplatIn {
  float x, y, ca, cb, cc, opac;
  int radius;
};

__device__ __forceinline__ SplatIn load_splat_in(const int g, const float *__restrict__ xys,
                                                 const int *__restrict__ radii, const float *__restrict__ conics,
                                                 const float *__restrict__ opacities) {
  SplatIn s;
  s.radius = radii[g];
  s.x = xys[2 * g], s.y = xys[2 * g + 1];
  s.ca = s.cb = s.cc = s.opac = 0.f;
  if (conics) s.ca = conics[3 * g], s.cb = conics[3 * g + 1], s.cc = conics[3 * g + 2], s.opac = opacities[g];
  return s;
}

// The per-Gaussian record of the list builders: centre, conic, the sigma bound of the exact
// reach test and the tile box (have_conics false: every box tile counts, smax = inf).
__device__ __forceinline__ SplatRec splat_record_from(const SplatIn &s, const bool have_conics, const int tiles_x,
                                                      const int tiles_y, const int bw) {
  SplatRec rec{0.f, 0.f, 1.f, 0.f, 1.f, -1.f, 0u, 0u};
  if (s.radius > 0) {
    int minx, miny, maxx, maxy;
    gsr_tile_bbox(s.x, s.y, (float)s.radius, tiles_x, tiles_y, 0.f, bw, minx, miny, maxx, maxy);
    rec.x = s.x;
    rec.y = s.y;
    rec.smax = INFINITY;
    if (have_conics) {
      const gsr::Reach rc = gsr::make_reach(s.x, s.y, s.ca, s.cb, s.cc, s.opac);
      rec.a = rc.a;
      rec.b = rc.b;
      rec.c = rc.c;
      rec.smax = rc.smax;
    }
    if (maxx > minx && maxy > miny && !(rec.smax < 0.f)) {
      rec.box0 = (unsigned)minx | ((unsigned)miny << 16);
      rec.box1 = (unsigned)(maxx - minx) | ((unsigned)(maxy - miny) << 16);
    }
  }
  return rec;
}

__device__ __forceinline__ SplatRec make_splat_record(const int g, const float *__restrict__ xys,
                                                      const int *__restrict__ radii,
                                                      const float *__restrict__ conics,
                                                      const float *__restrict__ opacities, const int tiles_x,
                                                      const int tiles_y, const int bw) {
  SplatIn s{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, radii[g]};
  if (s.radius > 0) {
    s.x = xys[2 * g], s.y = xys[2 * g + 1];
    if (conics) s.ca = conics[3 * g], s.cb = conics[3 * g + 1], s.cc = conics[3 * g + 2], s.opac = opacities[g];
  }
  return splat_record_from(s, conics != nullptr, tiles_x, tiles_y, bw);
}

// derived per Gaussian, kept in LDS for the row loop
struct RowParams {
  float D;      // a c - b^2
  float umax;   // half x-extent of {sigma <= smax}
  float vmax;   // half y-extent
  float vstar;  // y offset of the rightmost point is -vstar, of the leftmost +vstar
};

__device__ __forceinline__ RowParams make_row_params(const SplatRec &r) {
  RowParams p{1.f, 0.f, 0.f, 0.f};
  if (r.smax >= 0.f && r.smax != INFINITY) {
    p.D = r.a * r.c - r.b * r.b;  // > 0 (make_reach sets smax = inf otherwise)
    const float t = 2.f * r.smax / p.D;
    p.umax = sqrtf(t * r.c);
    p.vmax = sqrtf(t * r.a);
    p.vstar = r.b * p.umax / r.c;
  }
  return p;
}

// Tiles [t0, t1) of tile row `ty` (inside the box) in which the splat can reach
// alpha >= 1/255, i.e. whose pixel-centre rectangle [16tx, 16tx+15] x [16ty, 16ty+15]
// meets the ellipse {sigma <= smax}.  The ellipse cut by the row's band is convex,
// so its x-projection is one interval [xl, xr]; xr is attained at the band's point
// closest (in y) to the ellipse's rightmost point, xl likewise.  Conservative:
// `smax` carries a 1 % margin in alpha (make_reach) and the interval is widened
// by 1e-3 of the ellipse's extent + 0.05 px against rounding in the square roots.
// The compositing kernels re-test per sub-tile / pixel, so keeping a dead pair is
// harmless; dropping a live one is what the margins exclude
// (tests/test_gpu_kernels.py::test_exact_lists_drop_only_dead_pairs).
__device__ __forceinline__ void row_range(const SplatRec &r, const RowParams &p, int ty, int &t0, int &t1) {
  const int minx = (int)(r.box0 & 0xffffu), bwid = (int)(r.box1 & 0xffffu);
  t0 = minx;
  t1 = minx + bwid;
  if (r.smax == INFINITY) return;
  if (r.smax < 0.f) {
    t1 = t0;
    return;
  }
  const float v0 = 16.f * (float)ty - r.y, v1 = v0 + 15.f;
  const float mv = 1e-3f * p.vmax + 0.05f, mu = 1e-3f * p.umax + 0.05f;
  if (v0 > p.vmax + mv || v1 < -p.vmax - mv) {
    t1 = t0;
    return;
  }
  const float two_as = 2.f * r.a * r.smax;
  const float vr = fminf(fmaxf(-p.vstar, v0), v1), vl = fminf(fmaxf(p.vstar, v0), v1);
  const float inv_a = 1.f / r.a;
  const float xr = (-r.b * vr + sqrtf(fmaxf(two_as - p.D * vr * vr, 0.f))) * inv_a + mu;
  const float xl = (-r.b * vl - sqrtf(fmaxf(two_as - p.D * vl * vl, 0.f))) * inv_a - mu;
  // 16 tx <= x + xr   and   16 tx + 15 >= x + xl
  const float f0 = fminf(fmaxf(ceilf((r.x + xl - 15.f) * 0.0625f), (float)t0), (float)t1);
  const float f1 = fminf(fmaxf(floorf((r.x + xr) * 0.0625f) + 1.f, (float)t0), (float)t1);
  t0 = (int)f0;
  t1 = (int)f1 > t0 ? (int)f1 : t0;
}


}  // namespace
