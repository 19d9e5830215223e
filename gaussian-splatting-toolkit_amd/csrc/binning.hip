// binning.hip -- tile binning: per-Gaussian tile counts -> offsets, (tile|depth)
// key emission, radix sort, per-tile ranges (gfx950).
//
// Restates rasterizer/utils.py:106-182 and the kernels forward.cu:94-154 of
// the reference.  The reference sorts full 64-bit keys with torch.sort
// (unstable) and gathers the values through an int64 index tensor; here the
// (key, gaussian id) pairs go through rocPRIM's LSD radix sort directly, only
// over the significant key bits (32 depth bits + ceil(log2(#tiles))), which is
// stable: equal (tile, depth) keys keep emission order, i.e. ascending
// Gaussian id -- a deterministic refinement of the reference's unspecified tie
// order.  All of it is HBM-bound integer work.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "gsr_common.h"

namespace {

__global__ __launch_bounds__(256) void map_intersects_kernel(
    const int n, const float *__restrict__ xys, const float *__restrict__ depths,
    const int *__restrict__ radii, const int *__restrict__ cum_tiles_hit,
    const int tiles_x, const int tiles_y, const int bw,
    int64_t *__restrict__ isect_ids, int *__restrict__ gaussian_ids) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int r = radii[i];
  if (r <= 0) return;
  int minx, miny, maxx, maxy;
  gsr_tile_bbox(xys[2 * i], xys[2 * i + 1], (float)r, tiles_x, tiles_y, 0.f, bw, minx, miny,
                maxx, maxy);
  int cur = (i == 0) ? 0 : cum_tiles_hit[i - 1];
  // depth > 0 for every visible splat, so its bit pattern sorts like the float
  const int64_t depth_id = (int64_t)__float_as_int(depths[i]);
  for (int ty = miny; ty < maxy; ++ty) {
    const int64_t row = (int64_t)ty * tiles_x;
    for (int tx = minx; tx < maxx; ++tx) {
      isect_ids[cur] = ((row + tx) << 32) | depth_id;
      gaussian_ids[cur] = i;
      ++cur;
    }
  }
}

// tile_bins is written exactly once per element: [first, one-past-last) for
// tiles that own keys, (0,0) for the rest -- one lane per sorted key marks the
// boundaries, one lane per tile zero-fills the tiles no key names.
__global__ __launch_bounds__(256) void tile_bins_clear_kernel(const int num_tiles,
                                                              int2 *__restrict__ tile_bins) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < num_tiles) tile_bins[t] = make_int2(0, 0);
}

__global__ __launch_bounds__(256) void tile_bin_edges_kernel(
    const int num_intersects, const int64_t *__restrict__ isect_ids_sorted,
    int *__restrict__ tile_bins) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= num_intersects) return;
  const int cur = (int)(isect_ids_sorted[i] >> 32);
  if (i == 0) tile_bins[2 * cur] = 0;
  if (i == num_intersects - 1) tile_bins[2 * cur + 1] = num_intersects;
  if (i == 0) return;
  const int prev = (int)(isect_ids_sorted[i - 1] >> 32);
  if (prev != cur) {
    tile_bins[2 * prev + 1] = i;
    tile_bins[2 * cur] = i;
  }
}

inline unsigned key_end_bit(int num_tiles) {
  unsigned bits = 0;
  while ((1ll << bits) < (long long)num_tiles) ++bits;  // ceil(log2(num_tiles))
  return 32u + (bits == 0 ? 1u : bits);
}

}  // namespace

GSR_EXPORT size_t gsr_cumsum_workspace_bytes(int num_points) {
  if (num_points <= 0) return 0;
  size_t bytes = 0;
  (void)rocprim::inclusive_scan(nullptr, bytes, (const int *)nullptr, (int *)nullptr,
                                (size_t)num_points, rocprim::plus<int>());
  return bytes;
}

GSR_EXPORT int gsr_cumsum_tiles(int num_points, const int32_t *num_tiles_hit,
                                int32_t *cum_tiles_hit, void *workspace, size_t workspace_bytes,
                                gsr_stream_t stream) {
  GSR_REQUIRE(num_points >= 0, "cumsum_tiles: num_points < 0");
  if (num_points == 0) return GSR_OK;
  GSR_REQUIRE(num_tiles_hit && cum_tiles_hit, "cumsum_tiles: null pointer");
  size_t need = gsr_cumsum_workspace_bytes(num_points);
  if (workspace_bytes < need || (need > 0 && !workspace)) {
    gsr_set_error("cumsum_tiles: workspace %zu < %zu bytes", workspace_bytes, need);
    return GSR_ENOMEM;
  }
  GSR_CHECK_HIP(rocprim::inclusive_scan(workspace, workspace_bytes, num_tiles_hit, cum_tiles_hit,
                                        (size_t)num_points, rocprim::plus<int>(),
                                        (hipStream_t)stream));
  return GSR_OK;
}

GSR_EXPORT int gsr_map_intersects(int num_points, int num_intersects, const float *xys,
                                  const float *depths, const int32_t *radii,
                                  const int32_t *cum_tiles_hit, int tiles_x, int tiles_y,
                                  unsigned block_width, int64_t *isect_ids, int32_t *gaussian_ids,
                                  gsr_stream_t stream) {
  GSR_REQUIRE(num_points >= 0 && num_intersects >= 0, "map_intersects: negative size");
  GSR_REQUIRE(block_width >= 2 && block_width <= 16, "map_intersects: block_width must be in [2,16]");
  GSR_REQUIRE(tiles_x > 0 && tiles_y > 0, "map_intersects: empty tile grid");
  if (num_points == 0 || num_intersects == 0) return GSR_OK;
  GSR_REQUIRE(xys && depths && radii && cum_tiles_hit && isect_ids && gaussian_ids,
              "map_intersects: null pointer");
  hipLaunchKernelGGL(map_intersects_kernel, dim3(gsr_cdiv(num_points, 256)), dim3(256), 0,
                     (hipStream_t)stream, num_points, xys, depths, radii, cum_tiles_hit, tiles_x,
                     tiles_y, (int)block_width, isect_ids, gaussian_ids);
  GSR_CHECK_LAUNCH("map_intersects");
  return GSR_OK;
}

GSR_EXPORT size_t gsr_sort_workspace_bytes(int num_intersects) {
  if (num_intersects <= 0) return 0;
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                  (const int *)nullptr, (int *)nullptr, (size_t)num_intersects, 0,
                                  64);
  return bytes;
}

GSR_EXPORT int gsr_sort_intersects(int num_intersects, int num_tiles, const int64_t *isect_ids,
                                   const int32_t *gaussian_ids, int64_t *isect_ids_sorted,
                                   int32_t *gaussian_ids_sorted, void *workspace,
                                   size_t workspace_bytes, gsr_stream_t stream) {
  GSR_REQUIRE(num_intersects >= 0, "sort_intersects: num_intersects < 0");
  GSR_REQUIRE(num_tiles > 0, "sort_intersects: num_tiles <= 0");
  if (num_intersects == 0) return GSR_OK;
  GSR_REQUIRE(isect_ids && gaussian_ids && isect_ids_sorted && gaussian_ids_sorted,
              "sort_intersects: null pointer");
  size_t need = gsr_sort_workspace_bytes(num_intersects);
  if (workspace_bytes < need || (need > 0 && !workspace)) {
    gsr_set_error("sort_intersects: workspace %zu < %zu bytes", workspace_bytes, need);
    return GSR_ENOMEM;
  }
  // keys are non-negative (tile id >= 0, depth bits of a positive float), so
  // the unsigned order over the low `end_bit` bits equals the signed order.
  GSR_CHECK_HIP(rocprim::radix_sort_pairs(
      workspace, workspace_bytes, reinterpret_cast<const uint64_t *>(isect_ids),
      reinterpret_cast<uint64_t *>(isect_ids_sorted), gaussian_ids, gaussian_ids_sorted,
      (size_t)num_intersects, 0u, key_end_bit(num_tiles), (hipStream_t)stream));
  return GSR_OK;
}

GSR_EXPORT int gsr_tile_bin_edges(int num_intersects, const int64_t *isect_ids_sorted,
                                  int num_tiles, int32_t *tile_bins, gsr_stream_t stream) {
  GSR_REQUIRE(num_intersects >= 0 && num_tiles >= 0, "tile_bin_edges: negative size");
  if (num_tiles == 0) return GSR_OK;
  GSR_REQUIRE(tile_bins, "tile_bin_edges: null pointer");
  hipLaunchKernelGGL(tile_bins_clear_kernel, dim3(gsr_cdiv(num_tiles, 256)), dim3(256), 0,
                     (hipStream_t)stream, num_tiles, reinterpret_cast<int2 *>(tile_bins));
  GSR_CHECK_LAUNCH("tile_bin_edges(clear)");
  if (num_intersects == 0) return GSR_OK;
  GSR_REQUIRE(isect_ids_sorted, "tile_bin_edges: null pointer");
  hipLaunchKernelGGL(tile_bin_edges_kernel, dim3(gsr_cdiv(num_intersects, 256)), dim3(256), 0,
                     (hipStream_t)stream, num_intersects, isect_ids_sorted, tile_bins);
  GSR_CHECK_LAUNCH("tile_bin_edges");
  return GSR_OK;
}
