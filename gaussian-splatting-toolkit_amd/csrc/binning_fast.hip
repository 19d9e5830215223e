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
#include <mutex>

#include <rocprim/rocprim.hpp>

#include "gsr_common.h"
#include "raster_common.h"
#include "tile_rows.h"

// sort_mid.hip
size_t gsr_sort_mid_workspace_bytes(int n);
int gsr_sort_mid(int n, const unsigned *keys_in, int *vals_out, int key_bits, void *workspace,
                 size_t workspace_bytes, hipStream_t s);
size_t gsr_sort_mid_depth_extra(int n, int rows);
int gsr_sort_mid_depth(int n, const float *depths, const int *radii, const int *counts, int rows, int *order,
                       int *cum, void *workspace, size_t workspace_bytes, hipStream_t s);
size_t gsr_sort_bucket_workspace_bytes(int n, int rows);
int gsr_sort_bucket_wave_cap(void);
int gsr_sort_bucket_depth(int n, const float *depths, const int *radii, int *order, void *workspace,
                          size_t workspace_bytes, int *stats, const float *xys, const float *conics,
                          const float *opacities, int tiles_x, int tiles_y, void *recs, const int *counts, int rows,
                          int *cum, hipStream_t s);
int gsr_sort_mid_pairs(int n, const unsigned *keys_in, const int *vals_in, unsigned *keys_out,
                       int *vals_out, int key_bits, void *workspace, size_t workspace_bytes,
                       hipStream_t s);

// tile_scatter.hip
int gsr_tile_band_rows(int tiles_x, int tiles_y, int *rows_per_band);
bool gsr_tile_scatter_supported(int tiles_x, int tiles_y);
size_t gsr_tile_scatter_workspace_bytes(int I, int tiles_per_band, int bands);
int gsr_tile_scatter(int I, const int *cum, int n, const unsigned *keys, const int *gids, int tiles_x, int tiles_y,
                     int *ids_sorted, int *tile_bins, int *count_out, int *slot_of_entry, void *workspace,
                     size_t workspace_bytes, hipStream_t s);

// tile_partition2.hip
bool gsr_tile_partition2_supported(int tiles_x, int tiles_y);
size_t gsr_tile_partition2_workspace_bytes(int n, int capacity, int tiles_x, int tiles_y);
int gsr_tile_partition2(int n, int capacity, const int *order, const void *recs, int tiles_x, int tiles_y,
                        int *ids_sorted, int *tile_bins, int *count_out, void *workspace, size_t workspace_bytes,
                        hipStream_t s);

namespace {

// the purpose-built sort wins between these sizes; rocPRIM elsewhere
constexpr int kMidSortMin = 1 << 16, kMidSortMax = 1 << 22;
inline bool use_mid_sort(int n) { return n > kMidSortMin && n <= kMidSortMax; }
// 64 k < N <= 4 M: one bucket pass + one in-LDS pass (sort_bucket.hip; with counts: they are gathered where the order
// is written and one look-back scan follows), as long as the buckets of recent views fit its in-LDS paths.  The bucket sort is correct for any input but slow on pathological ones (a
// million equal or almost equal depths end in one bucket); every call publishes its largest bucket to a pinned
// word, and a call that finds the last published value too large sorts with the four LSD passes instead -- and so do
// the next 32, 64, ... calls.  The hint is read without synchronisation (it lags by the views in flight) and never
// changes a result.  GSR_DEPTH_SORT=radix | bucket pins the choice (A/B measurements, tests).
struct BucketHint {
  int *pinned = nullptr;  // [0]: largest visible bucket of the most recent bucket sort that finished
  int cooldown = 0, fails = 0;
};
BucketHint g_bucket_hint[16];
std::mutex g_bucket_hint_mutex;

// -> use the bucket sort; *stats: where the call publishes (nullptr: nowhere)
// (below 64 k items the bucket sort's five launches cost what rocPRIM's do -- 60 us at 10 k items, measured -- and its
// sample of 256 runs of 16 items needs at least 4096)
inline bool bucket_sort_range(int n) { return use_mid_sort(n); }
inline bool use_bucket_sort(int n, hipStream_t s, int **stats) {
  *stats = nullptr;
  if (!bucket_sort_range(n)) return false;
  const char *e = getenv("GSR_DEPTH_SORT");
  if (e && e[0] == 'r') return false;
  if (e && e[0] == 'b') return true;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return true;
  std::lock_guard<std::mutex> lock(g_bucket_hint_mutex);
  BucketHint &h = g_bucket_hint[dev];
  if (!h.pinned) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) {
      (void)hipGetLastError();  // (e.g. the legacy stream while another one captures: not this call's error)
      return true;
    }
    if (cap != hipStreamCaptureStatusNone) return true;  // (no allocation inside a capture)
    if (hipHostMalloc(reinterpret_cast<void **>(&h.pinned), 64, hipHostMallocDefault) != hipSuccess) {
      h.pinned = nullptr;
      (void)hipGetLastError();
      return true;
    }
    h.pinned[0] = 0;
  }
  if (h.cooldown > 0) {
    --h.cooldown;
    return false;
  }
  const int largest = *reinterpret_cast<volatile int *>(h.pinned);
  if (largest > gsr_sort_bucket_wave_cap()) {
    h.cooldown = 32 << std::min(h.fails, 6);
    ++h.fails;
    h.pinned[0] = 0;  // (the next bucket sort reports afresh)
    return false;
  }
  if (largest > 0) h.fails = 0;
  *stats = h.pinned;
  return true;
}

// element j of the (band, depth position) sequence the tile counts are scanned in:
// counts[band j / n][order[j % n]]
struct TilesInOrder {
  const int *tiles, *order;
  int n;
  __device__ __forceinline__ int operator()(int j) const {
    const int b = j / n;
    return tiles[(size_t)b * n + order[j - b * n]];
  }
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

// records only (gsr_count_reach with counts == NULL): what the two-level partition needs
__global__ __launch_bounds__(256) void reach_records_kernel(const int n, const float *__restrict__ xys,
                                                            const int *__restrict__ radii,
                                                            const float *__restrict__ conics,
                                                            const float *__restrict__ opacities, const int tiles_x,
                                                            const int tiles_y, SplatRec *__restrict__ recs) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < n) recs[g] = make_splat_record(g, xys, radii, conics, opacities, tiles_x, tiles_y, 16);
}

// One wave handles 64 Gaussians and walks their (Gaussian, tile row) items
// LOAD-BALANCED: rows are numbered consecutively (prefix sum of the box heights)
// and lane l takes rows l, l+64, ... -- a lane per Gaussian looping over its own
// box leaves most of the wave idle behind the largest splat.
//
//   mode 0 (count): Gaussians in index order; writes counts[band][g] and recs[g]
//   mode 1 (emit):  Gaussians in depth order (`order`), one band of tile rows per
//           blockIdx.y; the tiles of a row are written at
//           cum[band n + first Gaussian of the wave - 1] + rank, which keeps the
//           stream ordered by (band, depth position, tile row-major)
//
// With `recs == nullptr` (mode 1 only) every box tile is emitted: the reference's
// lists.  Both modes evaluate row_range() with the same instructions on the same
// record, so counts and emission agree exactly.  kBands: more than one band of
// `rows_per_band` tile rows (tile_scatter.hip); the single-band instantiation keeps
// one counter per Gaussian in LDS.
constexpr int kMaxBands = 16;
template <bool kBands>
__global__ __launch_bounds__(256) void tile_rows_kernel(
    const int mode, const int n, const int *__restrict__ order, const int *__restrict__ cum,
    const float *__restrict__ xys, const int *__restrict__ radii, const float *__restrict__ conics,
    const float *__restrict__ opacities, const int tiles_x, const int tiles_y, const int bw,
    SplatRec *__restrict__ recs, unsigned *__restrict__ tile_keys, int *__restrict__ gaussian_ids,
    int *__restrict__ counts, const int capacity, const int num_bands, const int rows_per_band) {
  __shared__ int s_pref[4][64];
  __shared__ int s_cnt[4][kBands ? kMaxBands : 1][64];
  __shared__ int s_gid[4][64];
  __shared__ SplatRec s_rec[4][64];
  __shared__ RowParams s_par[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i0 = (blockIdx.x * 4 + w) * 64, i = i0 + lane;
  const int band = kBands ? (int)blockIdx.y : 0;  // mode 1
  const int g = i < n ? (mode ? order[i] : i) : 0;
  SplatRec rec{0.f, 0.f, 1.f, 0.f, 1.f, -1.f, 0u, 0u};
  if (i < n) {
    if (mode && recs) {
      rec = recs[g];
    } else {
      rec = make_splat_record(g, xys, radii, conics, opacities, tiles_x, tiles_y, bw);
      if (!mode) recs[g] = rec;
    }
    if (kBands && mode) {  // keep the rows of this band only
      const int miny = (int)(rec.box0 >> 16), maxy = miny + (int)(rec.box1 >> 16);
      const int lo = max(miny, band * rows_per_band), hi = min(maxy, (band + 1) * rows_per_band);
      rec.box0 = (rec.box0 & 0xffffu) | ((unsigned)lo << 16);
      rec.box1 = (rec.box1 & 0xffffu) | ((unsigned)(hi > lo ? hi - lo : 0) << 16);
    }
  }
  const int rows = (int)(rec.box1 >> 16);
  int incl = rows;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  const int total = __shfl(incl, 63);
  s_pref[w][lane] = incl;
  if (kBands) {
    for (int b = 0; b < num_bands; ++b) s_cnt[w][b][lane] = 0;
  } else {
    s_cnt[w][0][lane] = 0;
  }
  s_gid[w][lane] = g;
  s_rec[w][lane] = rec;
  s_par[w][lane] = make_row_params(rec);
  __syncthreads();
  int out = 0;
  if (mode && i0 < n) {
    const long long ci = (long long)band * n + i0;
    out = ci > 0 ? cum[ci - 1] : 0;
  }
  for (int q0 = 0; q0 < total; q0 += 64) {
    const int q = q0 + lane;
    int k = 0;  // number of Gaussians whose rows all precede q
#pragma unroll
    for (int step = 32; step > 0; step >>= 1)
      if (s_pref[w][k + step - 1] <= q) k += step;
    k = k < 63 ? k : 63;
    const SplatRec r = s_rec[w][k];
    const int ty = (int)(r.box0 >> 16) + q - (k ? s_pref[w][k - 1] : 0);
    int t0 = 0, t1 = 0;
    if (q < total) row_range(r, s_par[w][k], ty, t0, t1);
    const int cnt = t1 - t0;
    if (mode) {
      int sc = cnt;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(sc, o);
        if (lane >= o) sc += t;
      }
      int pos = out + sc - cnt;
      const unsigned key0 = (unsigned)(ty * tiles_x);
      const int gid = s_gid[w][k];
      for (int tx = t0; tx < t1 && pos < capacity; ++tx, ++pos) {
        tile_keys[pos] = key0 + (unsigned)tx;
        gaussian_ids[pos] = gid;
      }
      out += __shfl(sc, 63);
    } else if (cnt > 0) {
      atomicAdd(&s_cnt[w][kBands ? ty / rows_per_band : 0][k], cnt);
    }
  }
  if (!mode && i < n) {
    if (kBands) {
      for (int b = 0; b < num_bands; ++b) counts[(size_t)b * n + i] = s_cnt[w][b][lane];
    } else {
      counts[i] = s_cnt[w][0][lane];
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
size_t scan_temp(size_t n) {
  size_t b = 0;
  auto in = rocprim::make_transform_iterator(rocprim::counting_iterator<int>(0), TilesInOrder{nullptr, nullptr, 1});
  (void)rocprim::inclusive_scan(nullptr, b, in, (int *)nullptr, n, rocprim::plus<int>());
  return b;
}
size_t tile_sort_temp(int I) {
  size_t b = 0;
  (void)rocprim::radix_sort_pairs(nullptr, b, (const unsigned *)nullptr, (unsigned *)nullptr,
                                  (const int *)nullptr, (int *)nullptr, (size_t)I, 0, 32);
  return b;
}

}  // namespace

GSR_EXPORT int gsr_tile_bands(int tiles_x, int tiles_y) {
  const int b = gsr_tile_band_rows(tiles_x, tiles_y, nullptr);
  return b > 0 ? b : 1;
}

GSR_EXPORT size_t gsr_depth_order_workspace_bytes(int num_points, int num_bands) {
  if (num_points <= 0) return 0;
  if (num_bands < 1) num_bands = 1;
  const size_t kb = align_up(sizeof(unsigned) * (size_t)num_points);
  const size_t st = scan_temp((size_t)num_points * num_bands);
  // [depth keys][ either: rocPRIM (sorted keys + temp)  or: sort_mid workspace ; scan temp shares it ]
  const size_t rocprim_need = kb + align_up(std::max(depth_sort_temp(num_points), st));
  const size_t mid_need = align_up(std::max(gsr_sort_mid_workspace_bytes(num_points) +
                                                gsr_sort_mid_depth_extra(num_points, num_bands), st));
  const size_t need = kb + (use_mid_sort(num_points) ? mid_need : rocprim_need);
  return bucket_sort_range(num_points) ? std::max(need, gsr_sort_bucket_workspace_bytes(num_points, num_bands)) : need;
}

GSR_EXPORT int gsr_depth_order(int num_points, const float *depths, const int32_t *radii,
                               const int32_t *num_tiles_hit, int num_bands, int32_t *order, int32_t *cum_sorted,
                               void *workspace, size_t workspace_bytes, gsr_stream_t stream) {
  GSR_REQUIRE(num_points >= 0, "depth_order: num_points < 0");
  GSR_REQUIRE(num_bands >= 1 && num_bands <= kMaxBands, "depth_order: num_bands must be in [1,16]");
  GSR_REQUIRE((long long)num_points * num_bands < (1ll << 31), "depth_order: num_points * num_bands too large");
  if (num_points == 0) return GSR_OK;
  GSR_REQUIRE(depths && radii && order && workspace, "depth_order: null pointer");
  GSR_REQUIRE((num_tiles_hit == nullptr) == (cum_sorted == nullptr),
              "depth_order: num_tiles_hit and cum_sorted are given (or left NULL: order only) together");
  const bool order_only = cum_sorted == nullptr;
  const size_t need = gsr_depth_order_workspace_bytes(num_points, num_bands);
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
  int *bucket_stats = nullptr;
  if (use_bucket_sort(num_points, s, &bucket_stats))
    return gsr_sort_bucket_depth(num_points, depths, radii, order, workspace, workspace_bytes, bucket_stats, nullptr,
                                 nullptr, nullptr, 0, 0, nullptr, num_tiles_hit, num_bands, cum_sorted, s);
  if (use_mid_sort(num_points))  // keys, sort, gather of the counts and their scan in 13 launches (sort_mid.hip)
    return gsr_sort_mid_depth(num_points, depths, radii, num_tiles_hit, num_bands, order, cum_sorted, rest,
                              rest_bytes, s);
  hipLaunchKernelGGL(depth_keys_kernel, dim3(gsr_cdiv(num_points, 256)), dim3(256), 0, s, num_points,
                     depths, radii, keys_in);
  GSR_CHECK_LAUNCH("depth_order(keys)");
  {
    unsigned *keys_out = reinterpret_cast<unsigned *>(rest);
    size_t temp_bytes = rest_bytes - kb;
    GSR_CHECK_HIP(rocprim::radix_sort_pairs<depth_sort_config>(
        rest + kb, temp_bytes, (const unsigned *)keys_in, keys_out, rocprim::counting_iterator<int>(0),
        order, (size_t)num_points, 0u, 31u, s));
  }
  if (order_only) return GSR_OK;
  // the sort is finished with its workspace (stream order): reuse it for the scan
  auto in = rocprim::make_transform_iterator(rocprim::counting_iterator<int>(0),
                                             TilesInOrder{num_tiles_hit, order, num_points});
  GSR_CHECK_HIP(rocprim::inclusive_scan(rest, rest_bytes, in, cum_sorted, (size_t)num_points * num_bands,
                                        rocprim::plus<int>(), s));
  return GSR_OK;
}

namespace {
__global__ void publish_int_kernel(const int *__restrict__ src, int *__restrict__ dst) { *dst = *src; }
}  // namespace

GSR_EXPORT int gsr_publish_int32(const int32_t *src, int32_t *dst, gsr_stream_t stream) {
  GSR_REQUIRE(src && dst, "publish_int32: null pointer");
  hipLaunchKernelGGL(publish_int_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, src, dst);
  GSR_CHECK_LAUNCH("publish_int32");
  return GSR_OK;
}

GSR_EXPORT size_t gsr_reach_record_bytes(void) { return sizeof(SplatRec); }

GSR_EXPORT int gsr_count_reach(int num_points, const float *xys, const int32_t *radii, const float *conics,
                               const float *opacities, int tiles_x, int tiles_y, int num_bands, int32_t *counts,
                               void *reach_records, gsr_stream_t stream) {
  GSR_REQUIRE(num_points >= 0, "count_reach: num_points < 0");
  if (num_points == 0) return GSR_OK;
  GSR_REQUIRE(xys && radii && conics && opacities && reach_records, "count_reach: null pointer");
  GSR_REQUIRE(tiles_x > 0 && tiles_y > 0 && tiles_x <= 65535 && tiles_y <= 65535, "count_reach: bad tile grid");
  GSR_REQUIRE((reinterpret_cast<uintptr_t>(reach_records) & 15) == 0, "count_reach: reach_records must be 16-byte aligned");
  int rpb = tiles_y;
  int bands = gsr_tile_band_rows(tiles_x, tiles_y, &rpb);
  if (bands < 1) bands = 1, rpb = tiles_y;  // grids the scatter does not serve: one count per Gaussian
  GSR_REQUIRE(num_bands == 1 || num_bands == bands, "count_reach: num_bands must be 1 or gsr_tile_bands()");
  if (num_bands == 1) bands = 1, rpb = tiles_y;
  if (!counts) {  // records only: the caller's lists come from the two-level partition (gsr_bin_sorted_needs_counts)
    GSR_REQUIRE(num_bands == 1, "count_reach: counts may be NULL with num_bands == 1 only");
    hipLaunchKernelGGL(reach_records_kernel, dim3(gsr_cdiv(num_points, 256)), dim3(256), 0, (hipStream_t)stream,
                       num_points, xys, radii, conics, opacities, tiles_x, tiles_y,
                       static_cast<SplatRec *>(reach_records));
    GSR_CHECK_LAUNCH("count_reach(records)");
    return GSR_OK;
  }
  auto kernel = bands > 1 ? tile_rows_kernel<true> : tile_rows_kernel<false>;
  hipLaunchKernelGGL(kernel, dim3(gsr_cdiv(num_points, 256)), dim3(256), 0, (hipStream_t)stream, 0,
                     num_points, (const int *)nullptr, (const int *)nullptr, xys, radii, conics, opacities,
                     tiles_x, tiles_y, 16, static_cast<SplatRec *>(reach_records), (unsigned *)nullptr,
                     (int *)nullptr, counts, 0, bands, rpb);
  GSR_CHECK_LAUNCH("count_reach");
  return GSR_OK;
}

// gsr_count_reach(counts == NULL) + gsr_depth_order(order only) in one call: when the depth order takes the bucket sort
// (sort_bucket.hip) its first launch writes the records too -- their 60 B per Gaussian stream while that launch
// builds its bucket map -- otherwise the two calls run one after the other.  Same records, same order.
GSR_EXPORT int gsr_reach_records_depth_order(int num_points, const float *xys, const int32_t *radii,
                                             const float *conics, const float *opacities, const float *depths,
                                             int tiles_x, int tiles_y, void *reach_records, int32_t *order,
                                             void *workspace, size_t workspace_bytes, gsr_stream_t stream) {
  GSR_REQUIRE(num_points >= 0, "reach_records_depth_order: num_points < 0");
  if (num_points == 0) return GSR_OK;
  GSR_REQUIRE(xys && radii && conics && opacities && depths && reach_records && order && workspace,
              "reach_records_depth_order: null pointer");
  GSR_REQUIRE(tiles_x > 0 && tiles_y > 0 && tiles_x <= 65535 && tiles_y <= 65535, "reach_records_depth_order: bad tile grid");
  GSR_REQUIRE((reinterpret_cast<uintptr_t>(reach_records) & 15) == 0,
              "reach_records_depth_order: reach_records must be 16-byte aligned");
  if (workspace_bytes < gsr_depth_order_workspace_bytes(num_points, 1)) {
    gsr_set_error("reach_records_depth_order: workspace too small");
    return GSR_ENOMEM;
  }
  hipStream_t s = (hipStream_t)stream;
  int *bucket_stats = nullptr;
  if (use_bucket_sort(num_points, s, &bucket_stats))
    return gsr_sort_bucket_depth(num_points, depths, radii, order, workspace, workspace_bytes, bucket_stats, xys,
                                 conics, opacities, tiles_x, tiles_y, reach_records, nullptr, 0, nullptr, s);
  const int rc = gsr_count_reach(num_points, xys, radii, conics, opacities, tiles_x, tiles_y, 1, nullptr,
                                 reach_records, stream);
  if (rc != GSR_OK) return rc;
  return gsr_depth_order(num_points, depths, radii, nullptr, 1, order, nullptr, workspace, workspace_bytes, stream);
}

// How the depth-ordered stream is partitioned by tile:
//   's' (default when the tile grid fits): single-pass counting sort, tile_scatter.hip
//   'r': rocPRIM radix sort on the tile bits + edge detection
//   'm': sort_mid.hip + edge detection
// GSR_TILE_SORT=r|m|s overrides (A/B measurements).  Grids of more than one
// band need the per-band counts of gsr_count_reach, i.e. reach records.
//   't': two-level partition (tile_partition2.hip), given ONE count per Gaussian (num_bands == 1)
//        and the reach records: grids above 16384 tiles, and smaller grids whose lists hold at
//        least GSR_P2_MIN (default 1 M) + GSR_P2_PER_N (default 0) x N entries.  It needs neither
//        the per-Gaussian counts nor their scan (gsr_bin_sorted_needs_counts), and with those
//        gone it is ahead of the single pass from ~2 M entries on (bench.py, 1080p: 200 k Gaussians
//        / 2.2 M entries 0.79 -> 0.77 ms per step, 1 M / 4.46 M 1.14 -> 1.12, 1 M / 11 M 1.08 -> 1.03;
//        on spatially coherent scenes -- many lanes of a step on one tile -- the single-pass walk
//        degrades further).  Not when the caller wants slot_of_entry.
char tile_sort_mode(int tiles_x, int tiles_y, bool have_records, int num_bands, int num_points, int num_intersects,
                    bool want_slots) {
  static const char forced = [] {
    const char *e = getenv("GSR_TILE_SORT");
    return e ? e[0] : '\0';
  }();
  // (from how many list entries the two-level partition is ahead of the single pass: measured in round 2, see above)
  constexpr long long p2_min = 1000000ll;
  constexpr double p2_per_n = 0.0;
  if (forced == 'r' || forced == 'm') return forced;
  const int bands = gsr_tile_band_rows(tiles_x, tiles_y, nullptr);
  const bool p2_ok = have_records && num_bands == 1 && !want_slots && gsr_tile_partition2_supported(tiles_x, tiles_y);
  if (forced == 't' && p2_ok) return 't';
  if (bands == 1) return (forced == '\0' && p2_ok && num_intersects >= p2_min + (long long)(p2_per_n * num_points)) ? 't' : 's';
  if (!have_records) return 'r';
  if (num_bands == 1) return p2_ok ? 't' : 'r';
  return bands > 1 ? 's' : 'r';
}

GSR_EXPORT size_t gsr_bin_sorted_workspace_bytes(int num_points, int num_intersects, int tiles_x, int tiles_y) {
  if (num_intersects <= 0) return 0;
  int rpb = 0;
  const int bands = gsr_tile_band_rows(tiles_x, tiles_y, &rpb);
  const size_t scatter = bands > 0 ? gsr_tile_scatter_workspace_bytes(num_intersects, rpb * tiles_x, bands) : 0;
  const size_t two_level = gsr_tile_partition2_supported(tiles_x, tiles_y)
                               ? gsr_tile_partition2_workspace_bytes(num_points, num_intersects, tiles_x, tiles_y)
                               : 0;
  return std::max(two_level,
                  3 * align_up(4 * (size_t)num_intersects) +
                      align_up(std::max({tile_sort_temp(num_intersects), gsr_sort_mid_workspace_bytes(num_intersects),
                                         scatter})));
}

GSR_EXPORT int gsr_bin_sorted_needs_counts(int num_points, int num_intersects, int tiles_x, int tiles_y,
                                           int device_sized, int want_slots) {
  if (tiles_x <= 0 || tiles_y <= 0 || num_points <= 0 || num_intersects <= 0) return 1;
  const char mode = tile_sort_mode(tiles_x, tiles_y, true, 1, num_points,
                                   device_sized ? (int)(0.75 * num_intersects) : num_intersects, want_slots != 0);
  return mode == 't' ? 0 : 1;
}

namespace {
// device_sized: the stream length is cum_sorted[bands * num_points - 1] on the device and
// `num_intersects` is the capacity the caller sized its buffers for
int bin_sorted_impl(bool device_sized, int num_points, int num_intersects, const int32_t *order,
                    const int32_t *cum_sorted, const float *xys, const int32_t *radii,
                    const void *reach_records, int tiles_x, int tiles_y, unsigned block_width, int num_bands,
                    int32_t *gaussian_ids_sorted, int32_t *tile_bins, int32_t *count_out, int32_t *slot_of_entry,
                    void *workspace, size_t workspace_bytes, gsr_stream_t stream) {
  GSR_REQUIRE(num_points >= 0 && num_intersects >= 0, "bin_sorted: negative size");
  GSR_REQUIRE(block_width >= 2 && block_width <= 16, "bin_sorted: block_width must be in [2,16]");
  GSR_REQUIRE(tiles_x > 0 && tiles_y > 0, "bin_sorted: empty tile grid");
  GSR_REQUIRE(tile_bins, "bin_sorted: null pointer");
  GSR_REQUIRE(reach_records == nullptr || block_width == 16, "bin_sorted: reach records are for block_width 16");
  GSR_REQUIRE(tiles_x <= 65535 && tiles_y <= 65535, "bin_sorted: tile grid too large");
  hipStream_t s = (hipStream_t)stream;
  const int num_tiles = tiles_x * tiles_y;
  // (device-sized: `num_intersects` is a capacity -- the caller's estimate plus head-room, 25 % and
  // rounding in rasterizer/rasterize.py; the path is chosen for the list length it stands for)
  const char mode = tile_sort_mode(tiles_x, tiles_y, reach_records != nullptr, num_bands, num_points,
                                   device_sized ? (int)(0.75 * num_intersects) : num_intersects,
                                   slot_of_entry != nullptr);
  GSR_REQUIRE(!device_sized || mode == 's' || mode == 't',
              "bin_sorted_dev: needs the tile scatter / two-level partition (<= 16384 tiles, or reach records)");
  GSR_REQUIRE(slot_of_entry == nullptr || mode == 's', "bin_sorted: slot_of_entry needs the single-pass tile scatter");
  int rpb = tiles_y, bands = 1;
  if (reach_records && num_bands > 1) {
    bands = gsr_tile_band_rows(tiles_x, tiles_y, &rpb);
    if (bands < 1) bands = 1, rpb = tiles_y;
  }
  GSR_REQUIRE(num_bands == bands, "bin_sorted: num_bands must be 1 or gsr_tile_bands() (with reach records)");
  const bool nothing = num_points == 0 || num_intersects == 0;
  // (the scatter path writes every entry of tile_bins itself, and so does the two-level one for rows of
  //  up to 256 tiles)
  if (nothing || (mode != 's' && !(mode == 't' && tiles_x <= 256))) {
    hipLaunchKernelGGL(tile_bins_clear_kernel, dim3(gsr_cdiv(num_tiles, 256)), dim3(256), 0, s, num_tiles,
                       reinterpret_cast<int2 *>(tile_bins));
    GSR_CHECK_LAUNCH("bin_sorted(clear)");
  }
  if (num_points == 0 || num_intersects == 0) return GSR_OK;
  GSR_REQUIRE(order && xys && radii && gaussian_ids_sorted && workspace, "bin_sorted: null pointer");
  GSR_REQUIRE(cum_sorted || mode == 't',
              "bin_sorted: cum_sorted may be NULL only where gsr_bin_sorted_needs_counts() says so");
  const size_t need = gsr_bin_sorted_workspace_bytes(num_points, num_intersects, tiles_x, tiles_y);
  if (workspace_bytes < need) {
    gsr_set_error("bin_sorted: workspace %zu < %zu bytes", workspace_bytes, need);
    return GSR_ENOMEM;
  }
  if (mode == 't')  // large grid, one count per Gaussian: row-partitioned emission + per-row column partition
    return gsr_tile_partition2(num_points, num_intersects, order, reach_records, tiles_x, tiles_y,
                               gaussian_ids_sorted, tile_bins, count_out, workspace, workspace_bytes, s);
  char *ws = static_cast<char *>(workspace);
  const size_t ib = align_up(4 * (size_t)num_intersects);
  unsigned *tile_in = reinterpret_cast<unsigned *>(ws);
  unsigned *tile_out = reinterpret_cast<unsigned *>(ws + ib);
  int *ids_in = reinterpret_cast<int *>(ws + 2 * ib);
  void *temp = ws + 3 * ib;
  size_t temp_bytes = workspace_bytes - 3 * ib;
  auto kernel = bands > 1 ? tile_rows_kernel<true> : tile_rows_kernel<false>;
  hipLaunchKernelGGL(kernel, dim3(gsr_cdiv(num_points, 256), bands), dim3(256), 0, s, 1, num_points,
                     order, cum_sorted, xys, radii, (const float *)nullptr, (const float *)nullptr, tiles_x,
                     tiles_y, (int)block_width,
                     const_cast<SplatRec *>(static_cast<const SplatRec *>(reach_records)), tile_in, ids_in,
                     (int *)nullptr, num_intersects, bands, rpb);
  GSR_CHECK_LAUNCH("bin_sorted(emit)");
  if (mode == 's')
    return gsr_tile_scatter(num_intersects, cum_sorted, num_points, tile_in, ids_in, tiles_x, tiles_y,
                            gaussian_ids_sorted, tile_bins, count_out, slot_of_entry, temp, temp_bytes, s);
  // (bands == 1 here: the stream is in depth order over the whole grid)
  if (mode == 'm') {
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
}  // namespace

GSR_EXPORT int gsr_bin_sorted(int num_points, int num_intersects, const int32_t *order,
                              const int32_t *cum_sorted, const float *xys, const int32_t *radii,
                              const void *reach_records, int tiles_x, int tiles_y, unsigned block_width,
                              int num_bands, int32_t *gaussian_ids_sorted, int32_t *tile_bins,
                              int32_t *slot_of_entry, void *workspace, size_t workspace_bytes, gsr_stream_t stream) {
  return bin_sorted_impl(false, num_points, num_intersects, order, cum_sorted, xys, radii, reach_records, tiles_x,
                         tiles_y, block_width, num_bands, gaussian_ids_sorted, tile_bins, nullptr, slot_of_entry,
                         workspace, workspace_bytes, stream);
}

GSR_EXPORT int gsr_bin_sorted_dev(int num_points, int capacity, const int32_t *order,
                                  const int32_t *cum_sorted, const float *xys, const int32_t *radii,
                                  const void *reach_records, int tiles_x, int tiles_y, unsigned block_width,
                                  int num_bands, int32_t *gaussian_ids_sorted, int32_t *tile_bins,
                                  int32_t *count_out, int32_t *slot_of_entry, void *workspace,
                                  size_t workspace_bytes, gsr_stream_t stream) {
  GSR_REQUIRE(capacity > 0 && num_points > 0, "bin_sorted_dev: capacity and num_points must be positive");
  return bin_sorted_impl(true, num_points, capacity, order, cum_sorted, xys, radii, reach_records, tiles_x,
                         tiles_y, block_width, num_bands, gaussian_ids_sorted, tile_bins, count_out, slot_of_entry,
                         workspace, workspace_bytes, stream);
}

// ---- two-round lists ---------------------------------------------------
// The lists of the nearest Gaussians are a prefix of every tile's list.  After a first compositing round
// over such prefix lists (gsr_rasterize_forward_round(1)) the tiles whose every pixel has finished need
// nothing more; a Gaussian of the remaining depth range whose tile box holds no unfinished tile can
// therefore be dropped BEFORE the second round's lists are built -- per Gaussian, with one summed-area
// lookup, without any per-tile mask inside the partition.
namespace {
// sat[(y + 1) * (tiles_x + 1) + (x + 1)] = number of flagged tiles in [0, x] x [0, y]
// pass 1: one workgroup per tile row, running sums along the row (+ the zero border)
__global__ __launch_bounds__(256) void tile_flag_rows_kernel(const int tiles_x, const int *__restrict__ flags,
                                                             int *__restrict__ sat) {
  __shared__ int wsum[4];
  __shared__ int carry;
  const int y = blockIdx.x, tid = threadIdx.x, W = tiles_x + 1;
  if (tid == 0) carry = 0;
  if (y == 0)
    for (int i = tid; i < W; i += 256) sat[i] = 0;
  if (tid == 0) sat[(y + 1) * W] = 0;
  __syncthreads();
  for (int base = 0; base < tiles_x; base += 256) {
    const int x = base + tid;
    const int v = x < tiles_x ? (flags[y * tiles_x + x] != 0) : 0;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o);
      if ((tid & 63) >= o) incl += t;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    int before = carry;
    for (int k = 0; k < (tid >> 6); ++k) before += wsum[k];
    if (x < tiles_x) sat[(y + 1) * W + x + 1] = before + incl;
    __syncthreads();
    if (tid == 0) carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
}
// pass 2: one thread per column, running sums down the rows (coalesced across the threads)
__global__ __launch_bounds__(256) void tile_flag_cols_kernel(const int tiles_x, const int tiles_y, int *__restrict__ sat,
                                                             int *__restrict__ flagged_out) {
  const int x = blockIdx.x * 256 + threadIdx.x, W = tiles_x + 1;
  if (x >= tiles_x) return;
  int run = 0;
  for (int y0 = 0; y0 < tiles_y; y0 += 8) {  // eight independent loads in flight per step of the running sum
    int v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = y0 + k < tiles_y ? sat[(y0 + k + 1) * W + x + 1] : 0;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (y0 + k < tiles_y) {
        run += v[k];
        sat[(y0 + k + 1) * W + x + 1] = run;
      }
  }
  if (x == tiles_x - 1 && flagged_out) *flagged_out = run;
}

// order_out[i] = order[i] if a tile of the Gaussian's box is still flagged, else `dummy` (the index of a culled
// record): coalesced, nothing is written into the records.
__global__ __launch_bounds__(256) void saturation_filter_kernel(const int count, const int *__restrict__ order,
                                                                const SplatRec *__restrict__ recs, const int tiles_x,
                                                                const int *__restrict__ sat, const int dummy,
                                                                int *__restrict__ order_out,
                                                                int *__restrict__ survivors) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  bool keep = false;
  if (i < count) {
    const int g = order[i];
    const uint2 box = *reinterpret_cast<const uint2 *>(&recs[g].box0);
    const int w = (int)(box.y & 0xffffu), h = (int)(box.y >> 16);
    if (w > 0 && h > 0) {
      const int x0 = (int)(box.x & 0xffffu), y0 = (int)(box.x >> 16), W = tiles_x + 1;
      const int c = sat[(y0 + h) * W + x0 + w] - sat[y0 * W + x0 + w] - sat[(y0 + h) * W + x0] + sat[y0 * W + x0];
      keep = c > 0;
    }
    order_out[i] = keep ? g : dummy;
  }
  if (survivors) {
    const unsigned long long m = __ballot(keep);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(survivors, __popcll(m));
  }
}
}  // namespace

GSR_EXPORT size_t gsr_saturation_filter_workspace_bytes(int tiles_x, int tiles_y) {
  if (tiles_x <= 0 || tiles_y <= 0) return 0;
  return sizeof(int) * (size_t)(tiles_x + 1) * (size_t)(tiles_y + 1);
}

GSR_EXPORT int gsr_saturation_filter(int count, const int32_t *order, const void *reach_records, int dummy_index,
                                     const int32_t *tile_flags, int tiles_x, int tiles_y, int32_t *order_out,
                                     void *workspace, size_t workspace_bytes, int32_t *stats_out, gsr_stream_t stream) {
  GSR_REQUIRE(count >= 0 && tiles_x > 0 && tiles_y > 0 && tiles_x <= 65535 && tiles_y <= 65535 && dummy_index >= 0,
              "saturation_filter: bad sizes");
  GSR_REQUIRE(tile_flags && workspace && workspace_bytes >= gsr_saturation_filter_workspace_bytes(tiles_x, tiles_y),
              "saturation_filter: null pointer or workspace too small");
  hipStream_t s = (hipStream_t)stream;
  int *sat = static_cast<int *>(workspace);
  // stats_out (nullable, int32[2], device memory): [0] = flagged tiles, [1] = Gaussians kept; zeroed here
  if (stats_out)
    if (int zrc = gsr_zero_async(stats_out, 2 * sizeof(int32_t), s)) return zrc;
  hipLaunchKernelGGL(tile_flag_rows_kernel, dim3(tiles_y), dim3(256), 0, s, tiles_x, tile_flags, sat);
  hipLaunchKernelGGL(tile_flag_cols_kernel, dim3(gsr_cdiv(tiles_x, 256)), dim3(256), 0, s, tiles_x, tiles_y, sat,
                     stats_out ? stats_out : (int *)nullptr);
  if (count > 0) {
    GSR_REQUIRE(order && reach_records && order_out, "saturation_filter: null pointer");
    hipLaunchKernelGGL(saturation_filter_kernel, dim3(gsr_cdiv(count, 256)), dim3(256), 0, s, count, order,
                       static_cast<const SplatRec *>(reach_records), tiles_x, (const int *)sat, dummy_index, order_out,
                       stats_out ? stats_out + 1 : (int *)nullptr);
  }
  GSR_CHECK_LAUNCH("saturation_filter");
  return GSR_OK;
}

// The two-level partition over a SUB-RANGE of the depth order (`order` points at its first element): lists of
// those `count` Gaussians only, tile_bins relative to gaussian_ids_sorted.  Count-free (reach records required).
GSR_EXPORT int gsr_tile_lists_subrange(int count, int capacity, const int32_t *order, const void *reach_records,
                                       int tiles_x, int tiles_y, int32_t *gaussian_ids_sorted, int32_t *tile_bins,
                                       int32_t *count_out, void *workspace, size_t workspace_bytes,
                                       gsr_stream_t stream) {
  GSR_REQUIRE(count >= 0 && capacity >= 1, "tile_lists_subrange: bad sizes");
  GSR_REQUIRE(tile_bins && gaussian_ids_sorted, "tile_lists_subrange: null pointer");
  GSR_REQUIRE(gsr_tile_partition2_supported(tiles_x, tiles_y), "tile_lists_subrange: tile grid not supported");
  hipStream_t s = (hipStream_t)stream;
  if (count == 0) {
    const int num_tiles = tiles_x * tiles_y;
    hipLaunchKernelGGL(tile_bins_clear_kernel, dim3(gsr_cdiv(num_tiles, 256)), dim3(256), 0, s, num_tiles,
                       reinterpret_cast<int2 *>(tile_bins));
    if (count_out)
      if (int zrc = gsr_zero_async(count_out, sizeof(int32_t), s)) return zrc;
    GSR_CHECK_LAUNCH("tile_lists_subrange(clear)");
    return GSR_OK;
  }
  GSR_REQUIRE(order && reach_records && workspace, "tile_lists_subrange: null pointer");
  if (tiles_x > 256) {  // (the two-level partition finishes tile_bins itself only for rows of up to 256 tiles)
    const int num_tiles = tiles_x * tiles_y;
    hipLaunchKernelGGL(tile_bins_clear_kernel, dim3(gsr_cdiv(num_tiles, 256)), dim3(256), 0, s, num_tiles,
                       reinterpret_cast<int2 *>(tile_bins));
  }
  return gsr_tile_partition2(count, capacity, order, reach_records, tiles_x, tiles_y, gaussian_ids_sorted, tile_bins,
                             count_out, workspace, workspace_bytes, s);
}

GSR_EXPORT size_t gsr_tile_lists_subrange_workspace_bytes(int count, int capacity, int tiles_x, int tiles_y) {
  return gsr_tile_partition2_workspace_bytes(count, capacity, tiles_x, tiles_y);
}
