// binning_fast.hip -- the binning pipeline `_RasterizeGaussians.forward` uses
// (gfx950).  Same result as scan -> key emission -> 64-bit sort -> bin edges
// (binning.hip, the reference's pipeline: rasterizer/utils.py:106-182), i.e.
// `gaussian_ids_sorted` ordered by (tile, depth, Gaussian id) and `tile_bins`,
// bit for bit -- but organised for HBM traffic instead of for exposing the
// reference's intermediate arrays:
//
//   1. sort the N Gaussians by depth once      (32-bit keys, N elements)
//   2. scan their tile counts in that order    (offsets + total)
//   3. emit (tile id, Gaussian id) in depth order
//   4. STABLE radix sort by tile id only       (ceil(log2 T) bits: 2 passes)
//   5. tile ranges from the sorted tile ids
//
// A stable sort by tile of a depth-ordered stream is ordered by (tile, depth);
// ties in depth keep ascending Gaussian id because step 1 is stable too.  The
// I-sized arrays (I = #intersections, ~8-20 N) are touched by 44 B/intersection
// instead of 164 B (12 B emission + six 24-B passes over 64-bit keys + 8 B).
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "gsr_common.h"

// sort_mid.hip
size_t gsr_sort_mid_workspace_bytes(int n);
int gsr_sort_mid(int n, const unsigned *keys_in, int *vals_out, int key_bits, void *workspace,
                 size_t workspace_bytes, hipStream_t s);
int gsr_sort_mid_pairs(int n, const unsigned *keys_in, const int *vals_in, unsigned *keys_out,
                       int *vals_out, int key_bits, void *workspace, size_t workspace_bytes,
                       hipStream_t s);

namespace {

// the purpose-built sort wins between these sizes; rocPRIM elsewhere
constexpr int kMidSortMin = 1 << 16, kMidSortMax = 1 << 22;
inline bool use_mid_sort(int n) { return n > kMidSortMin && n <= kMidSortMax; }

struct TilesInOrder {
  const int *tiles;
  __device__ __forceinline__ int operator()(int g) const { return tiles[g]; }
};

__global__ __launch_bounds__(256) void depth_keys_kernel(const int n, const float *__restrict__ depths,
                                                         const int *__restrict__ radii,
                                                         unsigned *__restrict__ keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // visible splats have depth > 0 (bit pattern orders like the value); culled
  // ones emit nothing, park them at the front
  keys[i] = radii[i] > 0 ? __float_as_uint(depths[i]) : 0u;
}

__global__ __launch_bounds__(256) void emit_in_depth_order_kernel(
    const int n, const int *__restrict__ order, const int *__restrict__ cum_sorted,
    const float *__restrict__ xys, const int *__restrict__ radii, const int tiles_x,
    const int tiles_y, const int bw, unsigned *__restrict__ tile_keys, int *__restrict__ gaussian_ids) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int g = order[i];
  const int r = radii[g];
  if (r <= 0) return;
  int minx, miny, maxx, maxy;
  gsr_tile_bbox(xys[2 * g], xys[2 * g + 1], (float)r, tiles_x, tiles_y, 0.f, bw, minx, miny, maxx, maxy);
  int cur = (i == 0) ? 0 : cum_sorted[i - 1];
  for (int ty = miny; ty < maxy; ++ty) {
    const unsigned row = (unsigned)(ty * tiles_x);
    for (int tx = minx; tx < maxx; ++tx) {
      tile_keys[cur] = row + (unsigned)tx;
      gaussian_ids[cur] = g;
      ++cur;
    }
  }
}

__global__ __launch_bounds__(256) void tile_bins_clear_kernel(const int num_tiles, int2 *__restrict__ tile_bins) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < num_tiles) tile_bins[t] = make_int2(0, 0);
}

__global__ __launch_bounds__(256) void tile_bin_edges32_kernel(const int num_intersects,
                                                               const unsigned *__restrict__ tile_sorted,
                                                               int *__restrict__ tile_bins) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= num_intersects) return;
  const int cur = (int)tile_sorted[i];
  if (i == 0) tile_bins[2 * cur] = 0;
  if (i == num_intersects - 1) tile_bins[2 * cur + 1] = num_intersects;
  if (i == 0) return;
  const int prev = (int)tile_sorted[i - 1];
  if (prev != cur) {
    tile_bins[2 * prev + 1] = i;
    tile_bins[2 * cur] = i;
  }
}

// The depth sort handles only N ~ 1e6 pairs, a size rocPRIM serves with
// block-sort + merge passes (21 launches, ~170 us on MI355X).  Measured
// alternatives: Onesweep forced from 64 k items with the default 16 k items per
// workgroup (62 workgroups: ~200 us) or with 2 k items per workgroup (look-back
// chain over 488 workgroups: ~350 us).  The default stays; a purpose-built
// 4-pass sort for this size is future work.
using depth_sort_config = rocprim::default_config;

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

inline unsigned tile_bits(int num_tiles) {
  unsigned bits = 1;
  while ((1ll << bits) < (long long)num_tiles) ++bits;
  return bits;
}

size_t depth_sort_temp(int n) {
  size_t b = 0;
  (void)rocprim::radix_sort_pairs<depth_sort_config>(nullptr, b, (const unsigned *)nullptr,
                                                     (unsigned *)nullptr, rocprim::counting_iterator<int>(0),
                                                     (int *)nullptr, (size_t)n, 0, 31);
  return b;
}
size_t scan_temp(int n) {
  size_t b = 0;
  auto in = rocprim::make_transform_iterator((const int *)nullptr, TilesInOrder{nullptr});
  (void)rocprim::inclusive_scan(nullptr, b, in, (int *)nullptr, (size_t)n, rocprim::plus<int>());
  return b;
}
size_t tile_sort_temp(int I) {
  size_t b = 0;
  (void)rocprim::radix_sort_pairs(nullptr, b, (const unsigned *)nullptr, (unsigned *)nullptr,
                                  (const int *)nullptr, (int *)nullptr, (size_t)I, 0, 32);
  return b;
}

}  // namespace

GSR_EXPORT size_t gsr_depth_order_workspace_bytes(int num_points) {
  if (num_points <= 0) return 0;
  const size_t kb = align_up(sizeof(unsigned) * (size_t)num_points);
  // [depth keys][ either: rocPRIM (sorted keys + temp)  or: sort_mid workspace ; scan temp shares it ]
  const size_t rocprim_need = kb + align_up(std::max(depth_sort_temp(num_points), scan_temp(num_points)));
  const size_t mid_need = align_up(std::max(gsr_sort_mid_workspace_bytes(num_points), scan_temp(num_points)));
  return kb + (use_mid_sort(num_points) ? mid_need : rocprim_need);
}

GSR_EXPORT int gsr_depth_order(int num_points, const float *depths, const int32_t *radii,
                               const int32_t *num_tiles_hit, int32_t *order, int32_t *cum_sorted,
                               void *workspace, size_t workspace_bytes, gsr_stream_t stream) {
  GSR_REQUIRE(num_points >= 0, "depth_order: num_points < 0");
  if (num_points == 0) return GSR_OK;
  GSR_REQUIRE(depths && radii && num_tiles_hit && order && cum_sorted && workspace, "depth_order: null pointer");
  const size_t need = gsr_depth_order_workspace_bytes(num_points);
  if (workspace_bytes < need) {
    gsr_set_error("depth_order: workspace %zu < %zu bytes", workspace_bytes, need);
    return GSR_ENOMEM;
  }
  hipStream_t s = (hipStream_t)stream;
  char *ws = static_cast<char *>(workspace);
  const size_t kb = align_up(sizeof(unsigned) * (size_t)num_points);
  unsigned *keys_in = reinterpret_cast<unsigned *>(ws);
  char *rest = ws + kb;
  size_t rest_bytes = workspace_bytes - kb;
  hipLaunchKernelGGL(depth_keys_kernel, dim3(gsr_cdiv(num_points, 256)), dim3(256), 0, s, num_points,
                     depths, radii, keys_in);
  GSR_CHECK_LAUNCH("depth_order(keys)");
  if (use_mid_sort(num_points)) {
    int rc = gsr_sort_mid(num_points, keys_in, order, 31, rest, rest_bytes, s);
    if (rc != GSR_OK) return rc;
  } else {
    unsigned *keys_out = reinterpret_cast<unsigned *>(rest);
    size_t temp_bytes = rest_bytes - kb;
    GSR_CHECK_HIP(rocprim::radix_sort_pairs<depth_sort_config>(
        rest + kb, temp_bytes, (const unsigned *)keys_in, keys_out, rocprim::counting_iterator<int>(0),
        order, (size_t)num_points, 0u, 31u, s));
  }
  // the sort is finished with its workspace (stream order): reuse it for the scan
  auto in = rocprim::make_transform_iterator((const int *)order, TilesInOrder{num_tiles_hit});
  GSR_CHECK_HIP(rocprim::inclusive_scan(rest, rest_bytes, in, cum_sorted, (size_t)num_points,
                                        rocprim::plus<int>(), s));
  return GSR_OK;
}

GSR_EXPORT size_t gsr_bin_sorted_workspace_bytes(int num_intersects) {
  if (num_intersects <= 0) return 0;
  return 3 * align_up(4 * (size_t)num_intersects) +
         align_up(std::max(tile_sort_temp(num_intersects), gsr_sort_mid_workspace_bytes(num_intersects)));
}

GSR_EXPORT int gsr_bin_sorted(int num_points, int num_intersects, const int32_t *order,
                              const int32_t *cum_sorted, const float *xys, const int32_t *radii,
                              int tiles_x, int tiles_y, unsigned block_width,
                              int32_t *gaussian_ids_sorted, int32_t *tile_bins, void *workspace,
                              size_t workspace_bytes, gsr_stream_t stream) {
  GSR_REQUIRE(num_points >= 0 && num_intersects >= 0, "bin_sorted: negative size");
  GSR_REQUIRE(block_width >= 2 && block_width <= 16, "bin_sorted: block_width must be in [2,16]");
  GSR_REQUIRE(tiles_x > 0 && tiles_y > 0, "bin_sorted: empty tile grid");
  GSR_REQUIRE(tile_bins, "bin_sorted: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int num_tiles = tiles_x * tiles_y;
  hipLaunchKernelGGL(tile_bins_clear_kernel, dim3(gsr_cdiv(num_tiles, 256)), dim3(256), 0, s, num_tiles,
                     reinterpret_cast<int2 *>(tile_bins));
  GSR_CHECK_LAUNCH("bin_sorted(clear)");
  if (num_points == 0 || num_intersects == 0) return GSR_OK;
  GSR_REQUIRE(order && cum_sorted && xys && radii && gaussian_ids_sorted && workspace, "bin_sorted: null pointer");
  const size_t need = gsr_bin_sorted_workspace_bytes(num_intersects);
  if (workspace_bytes < need) {
    gsr_set_error("bin_sorted: workspace %zu < %zu bytes", workspace_bytes, need);
    return GSR_ENOMEM;
  }
  char *ws = static_cast<char *>(workspace);
  const size_t ib = align_up(4 * (size_t)num_intersects);
  unsigned *tile_in = reinterpret_cast<unsigned *>(ws);
  unsigned *tile_out = reinterpret_cast<unsigned *>(ws + ib);
  int *ids_in = reinterpret_cast<int *>(ws + 2 * ib);
  void *temp = ws + 3 * ib;
  size_t temp_bytes = workspace_bytes - 3 * ib;
  hipLaunchKernelGGL(emit_in_depth_order_kernel, dim3(gsr_cdiv(num_points, 256)), dim3(256), 0, s,
                     num_points, order, cum_sorted, xys, radii, tiles_x, tiles_y, (int)block_width, tile_in,
                     ids_in);
  GSR_CHECK_LAUNCH("bin_sorted(emit)");
  static const bool mid_tile_sort = [] {
    const char *e = getenv("GSR_TILE_SORT");
    return e && e[0] == 'm';
  }();
  if (mid_tile_sort) {
    int rc = gsr_sort_mid_pairs(num_intersects, tile_in, ids_in, tile_out, gaussian_ids_sorted,
                                (int)tile_bits(num_tiles), temp, temp_bytes, s);
    if (rc != GSR_OK) return rc;
  } else {
    GSR_CHECK_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, (const unsigned *)tile_in, tile_out,
                                            (const int *)ids_in, gaussian_ids_sorted,
                                            (size_t)num_intersects, 0u, tile_bits(num_tiles), s));
  }
  hipLaunchKernelGGL(tile_bin_edges32_kernel, dim3(gsr_cdiv(num_intersects, 256)), dim3(256), 0, s,
                     num_intersects, (const unsigned *)tile_out, tile_bins);
  GSR_CHECK_LAUNCH("bin_sorted(edges)");
  return GSR_OK;
}
