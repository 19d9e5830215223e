/*
 * gsraster.h -- C ABI of the MI355X-native Gaussian-splatting rasterizer.
 *
 * This is the drop-in boundary for the reference's native module
 * `rasterizer.cuda` (pybind module `rasterizer_cuda` / `rasterizer.csrc`,
 * gs_toolkit/gs_components/rasterizer/cuda/csrc/ext.cpp:4-18, prototypes in
 * cuda/csrc/bindings.h:19-115).  The reference has no C ABI -- its boundary is
 * pybind11 + torch::Tensor; every entry point below names the `*_tensor`
 * wrapper it replaces.  The Python mirror of the reference's package
 * (`gaussian-splatting-toolkit_amd/rasterizer`) binds these with ctypes.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (hipMalloc'd, fp32 / int32 / int64 as
 *     named), contiguous, row-major, owned by the caller; inputs are never
 *     written; nothing is allocated or freed inside the library;
 *   - every output element is written by the call (the reference relies on
 *     torch::zeros pre-fill + early returns; here the kernels write the zeros
 *     themselves), EXCEPT the four gradient accumulators of
 *     gsr_rasterize_backward* which the call zero-fills itself before
 *     accumulating;
 *   - `stream` is a hipStream_t (NULL = the default stream); calls are
 *     asynchronous on that stream and never synchronise the device;
 *   - return value: 0 on success, a negative GSR_E* code otherwise, with a
 *     human-readable message available from gsr_last_error() (thread-local).
 *   - matrices are row-major; quaternions are (w,x,y,z); conics are the upper
 *     triangle (a,b,c) of the inverse 2-D covariance; tile id = ty*tiles_x+tx.
 */
#ifndef GSRASTER_H_
#define GSRASTER_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_VERSION 200 /* 0.2.0 */

#define GSR_OK 0
#define GSR_EINVAL -1   /* bad argument (shape / range / null pointer) */
#define GSR_ELAUNCH -2  /* HIP reported a launch / runtime error */
#define GSR_ENOMEM -3   /* workspace too small */

typedef void *gsr_stream_t; /* hipStream_t */

int gsr_version(void);
/* message of the most recent failing call on this thread ("" if none) */
const char *gsr_last_error(void);

/* ---- projection -------------------------------------------------------
 * replaces project_gaussians_forward_tensor (bindings.cu:107-160), kernel
 * forward.cu:13-90.  viewmat: >= 12 floats (top 3x4), projmat: 16 floats.
 * outputs: cov3d[n,6] xys[n,2] depths[n] radii[n](i32) conics[n,3]
 *          compensation[n] num_tiles_hit[n](i32)
 * Precomputed covariances (the `cov3D_precomp` of Inria-style callers): with
 * scales == NULL and quats == NULL, cov3d[n,6] (upper triangle xx xy xz yy yz zz,
 * world space) is an INPUT and is not written; gsr_project_backward with the same
 * two NULLs then stops at v_cov3d (v_scale / v_quat are not touched, may be NULL). */
int gsr_project_forward(int num_points, const float *means3d,
                        const float *scales, float glob_scale,
                        const float *quats, const float *viewmat,
                        const float *projmat, float fx, float fy, float cx,
                        float cy, unsigned img_height, unsigned img_width,
                        unsigned block_width, float clip_thresh, float *cov3d,
                        float *xys, float *depths, int32_t *radii,
                        float *conics, float *compensation,
                        int32_t *num_tiles_hit, gsr_stream_t stream);

/* replaces project_gaussians_backward_tensor (bindings.cu:164-216), kernel
 * backward.cu:305-347.  outputs: v_cov2d[n,3] v_cov3d[n,6] v_mean3d[n,3]
 * v_scale[n,3] v_quat[n,4].  Each of the cotangents v_xy, v_depth, v_conic,
 * v_compensation may be NULL (= all zeros). */
int gsr_project_backward(int num_points, const float *means3d,
                         const float *scales, float glob_scale,
                         const float *quats, const float *viewmat,
                         const float *projmat, float fx, float fy, float cx,
                         float cy, unsigned img_height, unsigned img_width,
                         const float *cov3d, const int32_t *radii,
                         const float *conics, const float *compensation,
                         const float *v_xy, const float *v_depth,
                         const float *v_conic, const float *v_compensation,
                         float *v_cov2d, float *v_cov3d, float *v_mean3d,
                         float *v_scale, float *v_quat, gsr_stream_t stream);

/* ---- spherical harmonics ----------------------------------------------
 * replaces compute_sh_forward_tensor / compute_sh_backward_tensor
 * (bindings.cu:58-103), kernels sh.cuh:188-224.  degree in 0..4 fixes the
 * coefficient layout [n, (degree+1)^2, 3]; degrees_to_use <= degree. */
int gsr_sh_forward(unsigned num_points, unsigned degree,
                   unsigned degrees_to_use, const float *viewdirs,
                   const float *coeffs, float *colors, gsr_stream_t stream);
int gsr_sh_backward(unsigned num_points, unsigned degree,
                    unsigned degrees_to_use, const float *viewdirs,
                    const float *v_colors, float *v_coeffs,
                    gsr_stream_t stream);

/* ---- binning ------------------------------------------------------------
 * inclusive int32 scan of num_tiles_hit; replaces the torch.cumsum of
 * compute_cumulative_intersects (utils.py:106-125).  The total is cum[n-1]
 * (device memory; the caller decides when to read it back). */
size_t gsr_cumsum_workspace_bytes(int num_points);
int gsr_cumsum_tiles(int num_points, const int32_t *num_tiles_hit,
                     int32_t *cum_tiles_hit, void *workspace,
                     size_t workspace_bytes, gsr_stream_t stream);

/* replaces map_gaussian_to_intersects_tensor (bindings.cu:218-251), kernel
 * forward.cu:94-127.  isect_ids[I] (i64) gaussian_ids[I] (i32). */
int gsr_map_intersects(int num_points, int num_intersects, const float *xys,
                       const float *depths, const int32_t *radii,
                       const int32_t *cum_tiles_hit, int tiles_x, int tiles_y,
                       unsigned block_width, int64_t *isect_ids,
                       int32_t *gaussian_ids, gsr_stream_t stream);

/* replaces torch.sort + torch.gather in bin_and_sort_gaussians
 * (utils.py:179-180): stable ascending radix sort of the (tile|depth) keys
 * over the significant bits only (32 depth bits + ceil(log2(num_tiles))). */
size_t gsr_sort_workspace_bytes(int num_intersects);
int gsr_sort_intersects(int num_intersects, int num_tiles,
                        const int64_t *isect_ids, const int32_t *gaussian_ids,
                        int64_t *isect_ids_sorted,
                        int32_t *gaussian_ids_sorted, void *workspace,
                        size_t workspace_bytes, gsr_stream_t stream);

/* replaces get_tile_bin_edges_tensor (bindings.cu:253-267), kernel
 * forward.cu:132-154.  tile_bins[num_tiles,2] (i32), (0,0) for empty tiles. */
int gsr_tile_bin_edges(int num_intersects, const int64_t *isect_ids_sorted,
                       int num_tiles, int32_t *tile_bins, gsr_stream_t stream);

/* ---- binning, fused pipeline ---------------------------------------------
 * What `_RasterizeGaussians.forward` needs from bin_and_sort_gaussians
 * (utils.py:128-182) is only `gaussian_ids_sorted` and `tile_bins`.  These two
 * calls produce exactly those (bit-identical to map + 64-bit sort + bin edges)
 * with ~4x less HBM traffic: Gaussians are ordered by depth once, intersections
 * are emitted in that order and then stably sorted by tile id only.
 *   gsr_depth_order : order[n] = Gaussian indices by (depth, index), culled
 *                     first; cum_sorted[n] = inclusive scan of num_tiles_hit in
 *                     that order (cum_sorted[n-1] = number of intersections).
 *   gsr_bin_sorted  : gaussian_ids_sorted[I], tile_bins[T,2].
 * All counts and list offsets are int32, as in the reference (`cum_tiles_hit`): a view with
 * 2^31 or more (Gaussian, tile) intersections is out of range -- sum num_tiles_hit in 64 bits
 * before sizing the lists if that can happen (the Python package does and raises).
 *
 * Exact lists (optional, block_width 16 only).  The reference lists every tile
 * of a splat's 3-sigma SQUARE (forward.cu:73-82); about half of those (splat,
 * tile) pairs cannot reach alpha >= 1/255 at any pixel of the tile, and the
 * compositing rule skips them pixel by pixel (forward.cu:349).  gsr_count_reach
 * counts, per Gaussian, the tiles that can, and fills one opaque record of
 * gsr_reach_record_bytes() bytes per Gaussian (16-byte aligned buffer).  Passing
 * the counts to gsr_depth_order (in place of num_tiles_hit) and the records to
 * gsr_bin_sorted builds the lists without the dead pairs: images and gradients
 * are unchanged, `gaussian_ids_sorted` is a subsequence of the reference's.
 * slot_of_entry (nullable, int32 as long as the lists): for the e-th list entry in
 * (band, depth position, tile) order -- the entries of the Gaussian at depth position
 * i in band b are e in [cum_sorted[b n + i - 1], cum_sorted[b n + i]) -- the index it got
 * in gaussian_ids_sorted: the inverse map gsr_rasterize_backward_det reduces along.
 * With reach_records == NULL gsr_bin_sorted reproduces the reference's lists
 * bit for bit.
 *
 * Large tile grids (above 16384 tiles; 4K at 16 px: 240 x 135) need the reach
 * records and are built in one of two ways, chosen by `num_bands` of
 * gsr_count_reach / gsr_depth_order / gsr_bin_sorted(_dev):
 *   num_bands = 1 (default of the Python package): TWO-LEVEL PARTITION -- the entries
 *     are emitted straight into per-tile-row segments and each row is then split by
 *     tile column, every store coalesced (csrc/tile_partition2.hip);
 *   num_bands = gsr_tile_bands(tiles_x, tiles_y) (4 at 4K): TILE-ROW BANDS -- the
 *     single-pass scatter band by band: gsr_count_reach writes counts[bands, n]
 *     (band-major), gsr_depth_order scans them in (band, depth) order into
 *     cum_sorted[bands * n]; slower on large lists, but hands out slot_of_entry
 *     (deterministic backward).
 * Grids up to 16384 tiles: num_bands = 1; the two-level partition from 1 M list
 * entries on, the single-pass scatter below (and whenever slot_of_entry is wanted).
 *
 * Lists without counts.  The two-level partition counts its entries itself (per tile
 * row, in depth order): where gsr_bin_sorted_needs_counts() returns 0 for the call
 * about to be made (same num_points / num_intersects-or-capacity / grid; reach records
 * and num_bands == 1 assumed), the caller may pass counts = NULL to gsr_count_reach
 * (records only, no walk over the tile rows), num_tiles_hit = cum_sorted = NULL to
 * gsr_depth_order (the order only: no gather, no scan) and cum_sorted = NULL to
 * gsr_bin_sorted(_dev); the number of entries then arrives through count_out of
 * gsr_bin_sorted_dev.  Same lists, two kernels and a scan less.
 * gsr_reach_records_depth_order is those two calls -- gsr_count_reach(counts = NULL) and
 * gsr_depth_order(num_tiles_hit = cum_sorted = NULL, num_bands = 1) -- as one: the same records and the same
 * order; where the depth order is built by the bucket sort (csrc/sort_bucket.hip: 64 k < num_points <= 4 M) the
 * records are written by its first launch.  workspace: gsr_depth_order_workspace_bytes(num_points, 1).
 *
 * How the depth sort is built (64 k < num_points <= 4 M, with or without counts) may depend on earlier calls on the same
 * device -- a hint in pinned memory sends calls to the radix passes after a view whose depths overflowed the
 * bucket sort's buckets -- the RESULT never does: both are the stable sort by (depth bits, index).
 * GSR_DEPTH_SORT=radix | bucket in the environment pins the choice. */
size_t gsr_reach_record_bytes(void);
int gsr_tile_bands(int tiles_x, int tiles_y);
int gsr_count_reach(int num_points, const float *xys, const int32_t *radii,
                    const float *conics, const float *opacities, int tiles_x,
                    int tiles_y, int num_bands, int32_t *counts,
                    void *reach_records, gsr_stream_t stream);
size_t gsr_depth_order_workspace_bytes(int num_points, int num_bands);
int gsr_depth_order(int num_points, const float *depths, const int32_t *radii,
                    const int32_t *num_tiles_hit, int num_bands,
                    int32_t *order, int32_t *cum_sorted, void *workspace,
                    size_t workspace_bytes, gsr_stream_t stream);
int gsr_reach_records_depth_order(int num_points, const float *xys, const int32_t *radii,
                                  const float *conics, const float *opacities, const float *depths,
                                  int tiles_x, int tiles_y, void *reach_records, int32_t *order,
                                  void *workspace, size_t workspace_bytes, gsr_stream_t stream);
size_t gsr_bin_sorted_workspace_bytes(int num_points, int num_intersects,
                                      int tiles_x, int tiles_y);
int gsr_bin_sorted_needs_counts(int num_points, int num_intersects, int tiles_x,
                                int tiles_y, int device_sized, int want_slots);
int gsr_bin_sorted(int num_points, int num_intersects, const int32_t *order,
                   const int32_t *cum_sorted, const float *xys,
                   const int32_t *radii, const void *reach_records, int tiles_x,
                   int tiles_y, unsigned block_width, int num_bands,
                   int32_t *gaussian_ids_sorted, int32_t *tile_bins,
                   int32_t *slot_of_entry, void *workspace,
                   size_t workspace_bytes, gsr_stream_t stream);

/* One int32 from device memory to any device-accessible address, e.g. pinned
 * host memory mapped into the device's address space: a one-thread kernel in
 * stream order instead of a copy operation.  Used to hand cum_sorted[n-1] to the
 * host as early as possible (right after gsr_depth_order). */
int gsr_publish_int32(const int32_t *src, int32_t *dst, gsr_stream_t stream);

/* gsr_bin_sorted without the host knowing the number of intersections: the
 * length of the lists is read on the device (the last element of cum_sorted, as
 * written by gsr_depth_order) and `capacity` is what gaussian_ids_sorted and the
 * workspace (gsr_bin_sorted_workspace_bytes(capacity, ...)) were sized for.  If the
 * length exceeds the capacity the lists are cut there (memory-safe, results
 * incomplete): the caller compares the two once the value has reached the host
 * and repeats the call with a larger capacity.  This removes the host round
 * trip of rasterizer/utils.py:124 (`cum_tiles_hit[-1].item()`) from the critical
 * path.  count_out (nullable) receives the uncut length; it only has to be
 * device-accessible, e.g. pinned host memory mapped into the device's address
 * space, which spares the copy as well.  Tile grids above 16384 tiles need the
 * reach records (bands, above); without them: GSR_EINVAL, use gsr_bin_sorted. */
int gsr_bin_sorted_dev(int num_points, int capacity, const int32_t *order,
                       const int32_t *cum_sorted, const float *xys,
                       const int32_t *radii, const void *reach_records,
                       int tiles_x, int tiles_y, unsigned block_width,
                       int num_bands, int32_t *gaussian_ids_sorted,
                       int32_t *tile_bins, int32_t *count_out,
                       int32_t *slot_of_entry,
                       void *workspace, size_t workspace_bytes,
                       gsr_stream_t stream);

/* ---- compositing ----------------------------------------------------------
 * replaces rasterize_forward_tensor (bindings.cu:269-328), kernel
 * forward.cu:278-395 (3 channels, fp32).  out_img[H,W,3] final_Ts[H,W]
 * final_idx[H,W](i32).
 * deep_tile_threshold (block_width 16 only; here, in gsr_rasterize_backward and in
 * the _rgbd variants): 0 = one wave per tile.  > 0: a tile whose list holds more
 * entries than this is composited by four waves, one per 8x8 sub-tile, so that a few
 * very deep tiles (clustered scenes) do not hold a kernel up after the rest of the
 * chip has drained.  Never changes a result (same per-pixel instruction sequence);
 * costs three idle workgroups per tile at launch.  A good value: 1.5x the mean list
 * length, not below 256 (a split tile costs ~2x the instructions per list entry).
 * deep_tile_threshold | GSR_DEEP_ORDERED (16x16 tiles; every entry that takes a
 * deep_tile_threshold except the two-round and the deterministic ones, which ignore
 * the flags): LONGEST JOB FIRST.  tile_bins must then be followed by
 * gsr_tile_jobs_ints(tiles_x, tiles_y) writable int32 (one allocation of
 * 2 * tiles + that many ints: room for TWO job arrays -- a launch uses the first
 * unless GSR_DEEP_SECOND is set -- and the address of the library's statistics); the compositing workgroups take their jobs from the array in
 * block order -- per XCD, its tiles' jobs (a whole tile, or the four sub-tile jobs of a
 * tile above the threshold) by decreasing half-octave of the list length ( a split
 * tile's jobs keyed by an eighth of it), stable inside a bucket -- so that the walks
 * that last longest start first and the launch does not end on a few long walks over an
 * emptying chip.  Never changes a result.
 * Who writes the array: the entry itself, with one small kernel in front of the
 * compositing launch -- or, with GSR_DEEP_PREBUILT also set, nobody: the caller has
 * called gsr_tile_jobs_build, which writes up to both arrays (say the forward's and,
 * with other parameters, the backward's) in ONE launch.  The buffer belongs to one
 * stream at a time.
 * With GSR_DEEP_SECOND (the backward's order) a launch whose lists are ALL ALIKE --
 * the longest within 1.5x the mean of the non-empty ones -- splits no tile, whatever
 * the threshold, and either order then keeps the static map's sequence: there is
 * nothing to balance and a split tile costs the backward 1.7x the instructions.
 * GSR_DEEP_TAIL_64THS(k), k = 0..63: the last k / 64 of the launch's whole-tile jobs
 * -- the shortest -- run as four sub-tile jobs each behind everything else, whatever
 * their length: quarter-length jobs to fill the launch's drain.
 * Bits: 0-21 threshold, 22-27 tail, 28 SECOND, 29 PREBUILT, 30 ORDERED. */
#define GSR_DEEP_ORDERED (1 << 30)
#define GSR_DEEP_PREBUILT (1 << 29)
#define GSR_DEEP_SECOND (1 << 28)
#define GSR_DEEP_TAIL_64THS(k) (((k) & 63) << 22)
size_t gsr_tile_jobs_ints(int tiles_x, int tiles_y);
int gsr_tile_jobs_build(int tiles_x, int tiles_y, int32_t *tile_bins, int deep_arg_first,
                        int deep_arg_second, gsr_stream_t stream);
/* The other mapping of the same compositing rule (measurement variant, 16x16 tiles, 3 channels):
 * lanes over the 64 staged splats, a wave-wide multiplicative prefix scan for the per-pixel
 * transmittance, ballot termination (forward.cu:349-385 is the serial loop it re-maps).  Same
 * outputs as gsr_rasterize_forward up to the rounding of T (a tree-ordered product). */
int gsr_rasterize_forward_scan(int tiles_x, int tiles_y, unsigned img_width, unsigned img_height,
                               const int32_t *gaussian_ids_sorted, const int32_t *tile_bins,
                               const float *xys, const float *conics, const float *colors,
                               const float *opacities, const float *background, float *out_img,
                               float *final_Ts, int32_t *final_idx, gsr_stream_t stream);

int gsr_rasterize_forward(int tiles_x, int tiles_y, unsigned block_width,
                          unsigned img_width, unsigned img_height,
                          const int32_t *gaussian_ids_sorted,
                          const int32_t *tile_bins, const float *xys,
                          const float *conics, const float *colors,
                          const float *opacities, const float *background,
                          float *out_img, float *final_Ts, int32_t *final_idx,
                          int deep_tile_threshold, gsr_stream_t stream);

/* replaces rasterize_backward_tensor (bindings.cu:476-528), kernel
 * backward.cu:133-303.  v_xy[n,2] v_conic[n,3] v_colors[n,3] v_opacity[n]
 * are zero-filled by the call, then accumulated with fp32 atomics.
 * v_output_alpha may be NULL (= all zeros), here and in the _nd variant. */
int gsr_rasterize_backward(unsigned img_height, unsigned img_width,
                           unsigned block_width, int num_points,
                           const int32_t *gaussian_ids_sorted,
                           const int32_t *tile_bins, const float *xys,
                           const float *conics, const float *colors,
                           const float *opacities, const float *background,
                           const float *final_Ts, const int32_t *final_idx,
                           const float *v_output, const float *v_output_alpha,
                           float *v_xy, float *v_conic, float *v_colors,
                           float *v_opacity, int deep_tile_threshold,
                           gsr_stream_t stream);

/* The same two calls with what `_RasterizeGaussians.forward` does around them folded in
 * (rasterize.py:145-176: `out_alpha = 1 - final_Ts`; the backward's zero-filled
 * accumulators).  block_width 16 only when the extras are used.
 *   out_alpha (nullable) [H,W]: 1 - final_Ts, written by the compositing kernel;
 *   zero_ptr / zero_bytes (nullable; multiples of 4): cleared by the forward launch --
 *     pass the backward's accumulators (v_xy .. v_opacity as one allocation), keep them
 *     untouched, and call gsr_rasterize_backward_ex with accumulators_zeroed = 1: the
 *     36 MB of stores at 1 M Gaussians disappear inside the VALU-bound forward kernel
 *     instead of taking a bandwidth-bound launch of their own. */
int gsr_rasterize_forward_ex(int tiles_x, int tiles_y, unsigned block_width,
                             unsigned img_width, unsigned img_height,
                             const int32_t *gaussian_ids_sorted,
                             const int32_t *tile_bins, const float *xys,
                             const float *conics, const float *colors,
                             const float *opacities, const float *background,
                             float *out_img, float *final_Ts, int32_t *final_idx,
                             int deep_tile_threshold, float *out_alpha,
                             void *zero_ptr, size_t zero_bytes, gsr_stream_t stream);
int gsr_rasterize_backward_ex(unsigned img_height, unsigned img_width,
                              unsigned block_width, int num_points,
                              const int32_t *gaussian_ids_sorted,
                              const int32_t *tile_bins, const float *xys,
                              const float *conics, const float *colors,
                              const float *opacities, const float *background,
                              const float *final_Ts, const int32_t *final_idx,
                              const float *v_output, const float *v_output_alpha,
                              float *v_xy, float *v_conic, float *v_colors,
                              float *v_opacity, int deep_tile_threshold,
                              int accumulators_zeroed, gsr_stream_t stream);

/* gsr_rasterize_forward_ex / _rgbd (16x16 tiles, 3 channels [+ one]) with DEPTH SEGMENTS: the list of every tile that is split over
 * four waves (deep_tile_threshold) and holds more than max(deep_tile_threshold, segment_min_entries) entries is cut
 * into `segments` (2..16) runs of whole 64-entry chunks.  Compositing is associative in (C, T): every run is walked
 * ONCE from T = 1 by its own waves, a resolve pass scales the runs' colour sums by the products of the runs in front and
 * finds the run in which each pixel crosses the stop rule's 1e-4, and only those (sub-tile, run) pairs are walked again
 * from the true incoming T -- the exact stop rule and final_idx of forward.cu:278-395 (round 6; rounds 4-5 walked every
 * list twice).  For tile grids too small to fill the chip (the
 * 480 x 270 phase of the reference's coarse-to-fine schedule, vanilla_gs.py:48-53, is 510 tiles).  Results equal
 * gsr_rasterize_forward_ex's to rounding.  workspace: gsr_rasterize_forward_seg_workspace_bytes(...) bytes, 16-byte
 * aligned.  segments < 2 or deep_tile_threshold <= 0: gsr_rasterize_forward_ex. */
size_t gsr_rasterize_forward_seg_workspace_bytes(unsigned img_height, unsigned img_width, int segments);
/* extra / out_extra: NULL, or the fourth channel of gsr_rasterize_forward_rgbd ([n] / [P], over extra_background) */
int gsr_rasterize_forward_seg(int tiles_x, int tiles_y, unsigned img_width, unsigned img_height,
                              const int32_t *gaussian_ids_sorted, const int32_t *tile_bins,
                              const float *xys, const float *conics, const float *colors,
                              const float *extra, const float *opacities, const float *background,
                              float extra_background, float *out_img, float *out_extra,
                              float *final_Ts, int32_t *final_idx, int deep_tile_threshold,
                              float *out_alpha, void *zero_ptr, size_t zero_bytes, int segments,
                              int segment_min_entries, void *workspace, size_t workspace_bytes,
                              gsr_stream_t stream);

/* gsr_rasterize_backward_ex / _rgbd (16x16 tiles) with DEPTH SEGMENTS: the list of every tile that is split over four waves
 * (deep_tile_threshold) and holds more than max(deep_tile_threshold, segment_min_entries) entries is cut into
 * `segments` (2..16) runs, each walked by its own waves; a pre-pass computes what every run does to the backward's
 * per-pixel state (backward.cu:133-303: T and the colour buffer -- an affine map per run), so the runs are
 * independent.  For tile grids too small to fill the chip and for scenes whose deepest tiles set the kernel's
 * duration.  Results equal gsr_rasterize_backward_ex's to rounding (T reaches a run as a product of run products).
 * workspace: gsr_rasterize_backward_seg_workspace_bytes(img_height, img_width, segments) bytes, 8-byte aligned.
 * segments < 2 or deep_tile_threshold <= 0: gsr_rasterize_backward_ex. */
size_t gsr_rasterize_backward_seg_workspace_bytes(unsigned img_height, unsigned img_width, int segments);
/* extra / v_output_extra / v_extra: NULL, or the fourth channel as in gsr_rasterize_backward_rgbd */
int gsr_rasterize_backward_seg(unsigned img_height, unsigned img_width, int num_points,
                               const int32_t *gaussian_ids_sorted, const int32_t *tile_bins,
                               const float *xys, const float *conics, const float *colors,
                               const float *extra, const float *opacities, const float *background,
                               float extra_background, const float *final_Ts, const int32_t *final_idx,
                               const float *v_output, const float *v_output_extra,
                               const float *v_output_alpha, float *v_xy, float *v_conic,
                               float *v_colors, float *v_extra, float *v_opacity,
                               int deep_tile_threshold, int accumulators_zeroed, int segments,
                               int segment_min_entries, void *workspace, size_t workspace_bytes,
                               gsr_stream_t stream);

/* generic channel count; replace nd_rasterize_forward_tensor /
 * nd_rasterize_backward_tensor (bindings.cu:330-469), kernels
 * forward.cu:159-276 / backward.cu:23-131.  Accumulation is fp32 here (the
 * reference accumulates in __half). 1 <= channels <= GSR_MAX_CHANNELS; above 32
 * channels the lists are walked once per 32 channels (the reference keeps all
 * channels of a tile's 256 pixels in 48 KB of shared memory: ~90 at most). */
#define GSR_MAX_CHANNELS 1024
int gsr_rasterize_forward_nd(int tiles_x, int tiles_y, unsigned block_width,
                             unsigned img_width, unsigned img_height,
                             unsigned channels,
                             const int32_t *gaussian_ids_sorted,
                             const int32_t *tile_bins, const float *xys,
                             const float *conics, const float *colors,
                             const float *opacities, const float *background,
                             float *out_img, float *final_Ts,
                             int32_t *final_idx, gsr_stream_t stream);
int gsr_rasterize_backward_nd(unsigned img_height, unsigned img_width,
                              unsigned block_width, unsigned channels,
                              int num_points,
                              const int32_t *gaussian_ids_sorted,
                              const int32_t *tile_bins, const float *xys,
                              const float *conics, const float *colors,
                              const float *opacities, const float *background,
                              const float *final_Ts, const int32_t *final_idx,
                              const float *v_output,
                              const float *v_output_alpha, float *v_xy,
                              float *v_conic, float *v_colors,
                              float *v_opacity, gsr_stream_t stream);

/* replaces compute_cov2d_bounds_tensor (bindings.cu:39-56).
 * conics[n,3] radii[n] (fp32) */
int gsr_cov2d_bounds(int num_pts, const float *cov2d, float *conics,
                     float *radii, gsr_stream_t stream);

/* ======================= "next" rows =======================
 * f2 -- fused photometric loss head.  Replaces, for the caller
 * GaussianSplattingModel.get_loss_dict (gs_toolkit/models/vanilla_gs.py:926-944),
 * `torch.abs(gt - pred).mean()` + `1 - pytorch_msssim.SSIM(data_range=1,
 * size_average=True, channel=3)(gt, pred)` and their autograd backward.
 * pred, gt: [H,W,3] fp32.  maps: scratch [9, H-10, W-10] fp32 written by the
 * forward and consumed by the backward.  sums: workspace of
 * GSR_LOSS_WORKSPACE_DOUBLES doubles (partial sums in GSR_LOSS_SUM_SLOTS slots each,
 * so that the atomics do not serialise on one address),
 * zeroed by the call.  *loss_out = (1-l)*L1 + l*(1 - SSIM) and terms_out[2] =
 * {L1 = mean |pred-gt|, SSIM mean} (terms_out may be NULL), written by a
 * one-wave kernel behind the main one.  clamp_pred != 0: the prediction is min(pred, 1), as after the models'
 * `torch.clamp(rgb, max=1.0)`, and the gradient is zero where pred > 1.
 * backward: v_pred[H,W,3] = upstream[0] * d loss / d pred (upstream on device). */
#define GSR_LOSS_SUM_SLOTS 64
#define GSR_LOSS_WORKSPACE_DOUBLES (2 * GSR_LOSS_SUM_SLOTS)
int gsr_l1_ssim_forward(unsigned img_height, unsigned img_width,
                        float ssim_lambda, int clamp_pred, const float *pred,
                        const float *gt, float *maps, double *sums,
                        float *loss_out, float *terms_out, gsr_stream_t stream);
int gsr_l1_ssim_backward(unsigned img_height, unsigned img_width,
                         float ssim_lambda, int clamp_pred,
                         const float *upstream, const float *pred,
                         const float *gt, const float *maps, float *v_pred,
                         gsr_stream_t stream);

/* f2, the photometric head of the co-gs model as its source computes it
 * (gs_toolkit/models/depth_gs.py:445-448: the `+ ssim_lambda * simloss` line is a
 * stand-alone expression statement, so `main_loss` = (1 - ssim_lambda) * L1 only):
 *   *loss_out = weight * mean |min(pred, 1) - gt|   (clamp_pred != 0; pred as is otherwise)
 * pred, gt: num_values fp32 each (an [H,W,3] image: 3 H W), 16-byte aligned.  sums:
 * workspace of GSR_LOSS_SUM_SLOTS doubles, zeroed by the call.  backward:
 * v_pred = upstream[0] * weight * sign(pred - gt) / num_values, 0 where pred > 1
 * under clamp_pred (upstream on the device). */
int gsr_l1_forward(long long num_values, float weight, int clamp_pred,
                   const float *pred, const float *gt, double *sums,
                   float *loss_out, gsr_stream_t stream);
int gsr_l1_backward(long long num_values, float weight, int clamp_pred,
                    const float *upstream, const float *pred, const float *gt,
                    float *v_pred, gsr_stream_t stream);

/* f2, depth head of the co-gs model (gs_toolkit/models/depth_gs.py:356-363, 531-538):
 *   pred = alpha > 0 ? depth / alpha : *depth_max      (depth_max: device float, the
 *                                                       detached maximum of `depth`)
 *   loss = mean over all pixels of |gt - pred| where gt > 0 (0 elsewhere)
 * depth = the depth pass of the compositing (depths as colours), alpha = 1 - T of the
 * RGB pass, gt = sensor / estimated depth; all [num_pixels] fp32.  sums: workspace of
 * GSR_LOSS_SUM_SLOTS doubles.  backward: cotangents of `depth` and `alpha`
 * (upstream[0] = d L / d loss, on the device). */
int gsr_depth_l1_forward(long long num_pixels, const float *depth,
                         const float *alpha, const float *gt,
                         const float *depth_max, double *sums, float *loss_out,
                         gsr_stream_t stream);
int gsr_depth_l1_backward(long long num_pixels, const float *upstream,
                          const float *depth, const float *alpha,
                          const float *gt, const float *depth_max,
                          float *v_depth, float *v_alpha, gsr_stream_t stream);

/* ---- SH colours from split coefficients (caller-side glue) --
 * gsr_sh_forward / gsr_sh_backward for models that keep the DC band and the
 * higher bands as two parameters (features_dc [n,3], features_rest [n,K-1,3])
 * and torch.cat them before every render (gs_toolkit/models/vanilla_gs.py:809,
 * `colors_crop = torch.cat(...)`): no concatenated copy is made, the gradients
 * are written straight into v_dc [n,3] and v_rest [n,K-1,3].  degree in [0,3]
 * (K = (degree+1)^2; degree 0: the DC band only, viewdirs / rest / v_rest may be NULL).  Optional epilogue of the models (vanilla_gs.py:826,
 * `torch.clamp(rgbs + 0.5, min=0.0)`): colors = sh + shift, cut at 0 when
 * clamp_zero != 0 -- a channel that was cut is stored as -0.0, so that the backward,
 * which takes those colours (clamped_colors, NULL if not clamped), blocks the gradient
 * exactly where sh + shift < 0 and passes it where it is >= 0, like torch.clamp. */
int gsr_sh_forward_split(unsigned num_points, unsigned degree,
                         unsigned degrees_to_use, const float *viewdirs,
                         const float *dc, const float *rest, float *colors,
                         float shift, int clamp_zero, gsr_stream_t stream);
int gsr_sh_backward_split(unsigned num_points, unsigned degree,
                          unsigned degrees_to_use, const float *viewdirs,
                          const float *v_colors, const float *clamped_colors,
                          float *v_dc, float *v_rest, gsr_stream_t stream);

/* SH backward over SEVERAL views at once: what per-view data parallelism exchanges instead of the SH
 * gradient itself.  One view's gradient is rank one per Gaussian, v_coeffs[g,k,:] = B_k(dir_g) v_colors[g,:], so
 * ranks all-gather their v_colors (12 B per Gaussian; already masked by the clamp epilogue, if any) and camera
 * positions instead of all-reducing 12 K bytes per Gaussian, and each forms
 *   scale * sum_r B_k(normalize(means3d[g] - camera_positions[r])) * v_colors[r][g]      (views in order r)
 * itself -- the same sum on every rank.  v_colors: view r at v_colors + r * v_colors_stride floats, [n,3] each;
 * camera_positions: view r at camera_positions + r * camera_stride floats.  Output: v_coeffs [n,K,3], or
 * (v_coeffs NULL) v_dc [n,3] and v_rest [n,K-1,3]; K = (degree+1)^2, degree in [0,3]; bands above degrees_to_use
 * are written as zeros.  The direction is formed as gsr_activate_forward forms it. */
int gsr_sh_backward_views(unsigned num_points, unsigned degree, unsigned degrees_to_use,
                          unsigned num_views, const float *means3d,
                          const float *camera_positions, size_t camera_stride,
                          const float *v_colors, size_t v_colors_stride, float scale,
                          float *v_dc, float *v_rest, float *v_coeffs, gsr_stream_t stream);

/* ---- RGB + one extra channel in ONE compositing pass -------
 * The models composite twice per view when they need a depth image: RGB, then
 * depths repeated as three colours over a zero background (vanilla_gs.py:840-855,
 * depth_gs.py:346-363).  These two calls composite colors [n,3] and extra [n]
 * (one scalar per Gaussian, e.g. its depth) together: out_img [H,W,3] over
 * `background`, out_extra [H,W] over `extra_background`; final_Ts / final_idx as
 * gsr_rasterize_forward.  block_width is 16.  The backward takes the cotangents of
 * both images (and of alpha, NULL = 0) and returns v_extra [n] next to the usual four.
 * out_alpha / zero_ptr / zero_bytes / accumulators_zeroed: as in gsr_rasterize_forward_ex /
 * gsr_rasterize_backward_ex (the accumulators are v_xy .. v_opacity, v_extra: 10 n floats). */
int gsr_rasterize_forward_rgbd(int tiles_x, int tiles_y, unsigned img_width,
                               unsigned img_height,
                               const int32_t *gaussian_ids_sorted,
                               const int32_t *tile_bins, const float *xys,
                               const float *conics, const float *colors,
                               const float *extra, const float *opacities,
                               const float *background, float extra_background,
                               float *out_img, float *out_extra,
                               float *final_Ts, int32_t *final_idx,
                               int deep_tile_threshold, float *out_alpha,
                               void *zero_ptr, size_t zero_bytes,
                               gsr_stream_t stream);
int gsr_rasterize_backward_rgbd(unsigned img_height, unsigned img_width,
                                int num_points,
                                const int32_t *gaussian_ids_sorted,
                                const int32_t *tile_bins, const float *xys,
                                const float *conics, const float *colors,
                                const float *extra, const float *opacities,
                                const float *background, float extra_background,
                                const float *final_Ts, const int32_t *final_idx,
                                const float *v_output,
                                const float *v_output_extra,
                                const float *v_output_alpha, float *v_xy,
                                float *v_conic, float *v_colors, float *v_extra,
                                float *v_opacity, int deep_tile_threshold,
                                int accumulators_zeroed, gsr_stream_t stream);

/* ---- two-round lists for deep scenes (block_width 16) --------------------
 * replaces, like gsr_bin_sorted + gsr_rasterize_forward / _backward, the list construction of
 * rasterizer/utils.py:106-182 and the walks of forward.cu:278-395 / backward.cu:133-303 -- with the SAME
 * results: on a scene whose tiles saturate early (3 M Gaussians at 4K: 97.7 M list entries built, 4.1 M ever
 * read) almost every entry lies behind the depth at which its tile's pixels have all finished.  The lists of
 * the NEAREST Gaussians (a prefix of the depth order) are a prefix of every tile's list:
 *   1. gsr_tile_lists_subrange over order[0 : n1]            -> lists 1 (ids at gaussian_ids_sorted + 0)
 *   2. gsr_rasterize_forward_round(1, lists 1)               -> final values where every pixel has finished, raw
 *                                                               per-pixel state + tile_flags (bit p: sub-tile p
 *                                                               holds raw state; zero them first) elsewhere
 *   3. gsr_saturation_filter over order[n1 : n]              -> order_out: the same ids, with `dummy_index` (the
 *                                                               index of a culled record, e.g. one appended row)
 *                                                               in place of every Gaussian whose tile box holds
 *                                                               no flagged tile
 *   4. gsr_tile_lists_subrange over order[n1 : n]            -> lists 2 (ids at gaussian_ids_sorted + idx_base,
 *                                                               tile_bins2 relative to idx_base)
 *   5. gsr_rasterize_forward_round(2, lists 2, idx_base)     -> resumes and finalises the flagged sub-tiles
 *   6. gsr_rasterize_backward_two(tile_bins, tile_bins2, idx_base)
 * A tile's list is its range in tile_bins followed by its range in tile_bins2; per pixel the instruction
 * sequence is that of one walk over the concatenation, so images, final_Ts and final_idx (an index into
 * gaussian_ids_sorted, as always) are bit-identical to the single walk over the full lists, gradients equal up to
 * the order of the float atomics.  extra / out_extra / v_output_extra / v_extra: the fourth channel of the
 * _rgbd calls, all NULL for three channels.  stats_out (nullable, int32[2], device-accessible): flagged tiles,
 * Gaussians kept. */
size_t gsr_tile_lists_subrange_workspace_bytes(int count, int capacity, int tiles_x, int tiles_y);
int gsr_tile_lists_subrange(int count, int capacity, const int32_t *order, const void *reach_records,
                            int tiles_x, int tiles_y, int32_t *gaussian_ids_sorted, int32_t *tile_bins,
                            int32_t *count_out, void *workspace, size_t workspace_bytes,
                            gsr_stream_t stream);
size_t gsr_saturation_filter_workspace_bytes(int tiles_x, int tiles_y);
int gsr_saturation_filter(int count, const int32_t *order, const void *reach_records, int dummy_index,
                          const int32_t *tile_flags, int tiles_x, int tiles_y, int32_t *order_out,
                          void *workspace, size_t workspace_bytes, int32_t *stats_out,
                          gsr_stream_t stream);
int gsr_rasterize_forward_round(int round, int tiles_x, int tiles_y, unsigned img_width,
                                unsigned img_height, const int32_t *gaussian_ids_sorted,
                                const int32_t *tile_bins, int idx_base, const float *xys,
                                const float *conics, const float *colors, const float *extra,
                                const float *opacities, const float *background, float extra_background,
                                float *out_img, float *out_extra, float *final_Ts, int32_t *final_idx,
                                int32_t *tile_flags, int deep_tile_threshold, float *out_alpha,
                                void *zero_ptr, size_t zero_bytes, gsr_stream_t stream);
int gsr_rasterize_backward_two(unsigned img_height, unsigned img_width, int num_points,
                               const int32_t *gaussian_ids_sorted, const int32_t *tile_bins,
                               const int32_t *tile_bins2, int idx_base2, const float *xys,
                               const float *conics, const float *colors, const float *extra,
                               const float *opacities, const float *background, float extra_background,
                               const float *final_Ts, const int32_t *final_idx, const float *v_output,
                               const float *v_output_extra, const float *v_output_alpha, float *v_xy,
                               float *v_conic, float *v_colors, float *v_extra, float *v_opacity,
                               int deep_tile_threshold, int accumulators_zeroed, gsr_stream_t stream);

/* ---- a whole view as ONE call (host-bound scenes and viewer frames) -----
 * The body of GaussianSplattingModel.get_outputs between the raw parameters and the images
 * (gs_toolkit/models/vanilla_gs.py:765-857: activations, project_gaussians, SH + 0.5 clamped,
 * tile lists, compositing -- with the depth image from the same pass when render_depth) and its
 * autograd backward, as two calls instead of ~18: the same exported functions in the same order,
 * run from C.  Nothing is allocated in here: every pointer is a caller-owned DEVICE buffer of the
 * size the per-stage function documents (n = num_points, P = H x W, T = tiles, 16 x 16 tiles).
 *   counts / cum      both NULL: lists without counts (gsr_bin_sorted_needs_counts() == 0 for
 *                     (n, capacity, grid, device_sized = 1)); else i32[n] each
 *   sort_ws / bin_ws  gsr_depth_order_workspace_bytes(n, 1) / gsr_bin_sorted_workspace_bytes(n, capacity, ..)
 *   count_out         i32[1], device or pinned host memory: entries the view needs (> capacity: cut)
 *   out_depth         [P], required iff render_depth;  out_alpha [P] nullable
 *   zero_ptr/bytes    the backward's accumulators (6 + 3 + render_depth floats per Gaussian), cleared
 *                     by the compositing launch; NULL: none */
typedef struct gsr_view_desc {
  int num_points, sh_degree, sh_degree_to_use, render_depth;
  int img_height, img_width;
  float fx, fy, cx, cy, glob_scale, clip_thresh;
  int capacity, deep_tile_threshold;
  const float *means, *log_scales, *raw_quats, *logits, *features_dc, *features_rest;
  const float *viewmat, *projmat, *campos, *background;
  float *scales, *quats, *opac, *dirs, *cov3d, *xys, *depths;
  int32_t *radii;
  float *conics, *comp;
  int32_t *tiles;
  float *colors;
  void *reach_records;
  int32_t *counts, *order, *cum, *ids, *tile_bins, *count_out;
  void *sort_ws;
  size_t sort_ws_bytes;
  void *bin_ws;
  size_t bin_ws_bytes;
  float *out_img, *out_depth, *final_Ts;
  int32_t *final_idx;
  float *out_alpha;
  void *zero_ptr;
  size_t zero_bytes;
  /* depth segments of the compositing, forward and backward (gsr_rasterize_forward_seg / _backward_seg):
   * segments < 2 = off; seg_ws: gsr_rasterize_forward_seg_workspace_bytes(...) bytes for gsr_view_forward,
   * gsr_rasterize_backward_seg_workspace_bytes(...) for gsr_view_backward (each call with its own `segments`) */
  int segments, segment_min_entries;
  void *seg_ws;
  size_t seg_ws_bytes;
} gsr_view_desc;

/* cotangents in, parameter gradients out.  accumulators: (9 + render_depth) n floats laid out
 * v_xy | v_conic | v_colors | v_opacity [| v_depths]; accumulators_zeroed != 0 when the forward's
 * launch cleared them (zero_ptr).  stats_first != NULL: also after_train's statistics
 * (gsr_densify_stats_dev).  tmp_*: scratch of [n,3] [n,6] [n,3] [n,4] floats.  v_dc == NULL: no SH backward --
 * the colour cotangents stay in the accumulators (before the clamp of the colours is applied to them) for a caller
 * that forms the SH gradient over several views (gsr_sh_backward_views). */
typedef struct gsr_view_grads {
  const float *v_img, *v_alpha, *v_depth; /* v_alpha / v_depth nullable (v_depth required iff render_depth) */
  float *accumulators;
  int accumulators_zeroed;
  const int32_t *stats_first;
  float stats_inv_size;
  float *xys_grad_norm;
  int32_t *vis_counts;
  float *max_2dsize;
  float *tmp_v_cov2d, *tmp_v_cov3d, *tmp_v_scales, *tmp_v_quats;
  float *v_means, *v_log_scales, *v_raw_quats, *v_logits, *v_dc, *v_rest;
} gsr_view_grads;

int gsr_view_forward(const gsr_view_desc *view, gsr_stream_t stream);

/* ---- the forward of rasterize_gaussians as ONE call ------------------------------------------
 * What `_RasterizeGaussians.forward` runs on the device (gs_toolkit/gs_components/rasterizer/rasterize.py:89-170:
 * bin_and_sort_gaussians, utils.py:128-182, then rasterize_forward) for 16 x 16 tiles and 3 colour channels: reach
 * records + depth order, device-sized tile lists, compositing (with alpha = 1 - T, the backward's cleared
 * accumulators, and optionally one extra channel, as gsr_rasterize_forward_ex / _rgbd).  The same exported functions
 * in the same order as the Python package would call them one by one; nothing is allocated in here.  The unchanged
 * models read `(num_tiles_hit > 0).any()` back right in front of this op (vanilla_gs.py:811): the GPU idles from
 * that read-back until the first launch below, so the host time in between is kept to one call.
 *   counts / cum   both NULL: lists without counts (gsr_bin_sorted_needs_counts() == 0); else i32[n] each
 *   order_ready    NULL, or the depth order already built for these depths / radii (gsr_depth_order(depths, radii,
 *                  NULL...), e.g. on another stream while the caller was busy): the sort is skipped and only the
 *                  records are written (needs counts == NULL); `order` / sort_ws may then be NULL
 *   reach_records  [n x gsr_reach_record_bytes()], 16-byte aligned;  order i32[n]
 *   sort_ws/bin_ws gsr_depth_order_workspace_bytes(n, 1) / gsr_bin_sorted_workspace_bytes(n, capacity, ..)
 *   ids i32[capacity], tile_bins i32[T,2], count_out i32[1] (device or pinned): entries needed (> capacity: cut)
 *   extra / out_extra  NULL, or [n] / [P]: one more channel composited with the same weights (the depths)
 *   out_img == NULL    the lists only: no compositing (colors / background / final_* / out_alpha / zero_ptr unused);
 *                      the caller composites later with gsr_rasterize_forward_ex -- how the Python package builds
 *                      a view's lists on a side stream while the models wait for their read-backs */
typedef struct gsr_raster_desc {
  int num_points, img_height, img_width, capacity, deep_tile_threshold;
  float extra_background;
  const float *xys, *depths;
  const int32_t *radii;
  const float *conics, *colors, *extra, *opac, *background;
  const int32_t *order_ready;
  void *reach_records;
  int32_t *counts, *order, *cum, *ids, *tile_bins, *count_out;
  void *sort_ws;
  size_t sort_ws_bytes;
  void *bin_ws;
  size_t bin_ws_bytes;
  float *out_img, *out_extra, *final_Ts;
  int32_t *final_idx;
  float *out_alpha;
  void *zero_ptr;
  size_t zero_bytes;
  /* depth segments of the compositing (gsr_rasterize_forward_seg): segments < 2 = off */
  int segments, segment_min_entries;
  void *seg_ws;
  size_t seg_ws_bytes;
  int deep_tile_threshold_backward; /* > 0 with GSR_DEEP_ORDERED | GSR_DEEP_SECOND: the coming backward's argument --
                                       its job order is written by the same launch as the forward's (the backward
                                       then passes it with GSR_DEEP_PREBUILT); 0: not built */
} gsr_raster_desc;
int gsr_rasterize_gaussians_forward(const gsr_raster_desc *desc, gsr_stream_t stream);
int gsr_view_backward(const gsr_view_desc *view, const gsr_view_grads *grads, gsr_stream_t stream);

/* ---- per-Gaussian activations (caller-side glue) ----------
 * exp(scales), quats / |quats|, sigmoid(opacities) and the normalised view
 * directions means - camera_position of GaussianSplattingModel.get_outputs
 * (gs_toolkit/models/vanilla_gs.py:765-826) in one launch, and their VJP in one
 * launch.  camera_position: 3 floats on the device (may be NULL together with
 * viewdirs).  Backward: v_scales / v_quats / v_opacities may each be NULL (= 0);
 * no gradient flows to the view directions (the reference detaches the means). */
int gsr_activate_forward(int num_points, const float *means,
                         const float *log_scales, const float *raw_quats,
                         const float *opacity_logits,
                         const float *camera_position, float *scales,
                         float *quats, float *opacities, float *viewdirs,
                         gsr_stream_t stream);
int gsr_activate_backward(int num_points, const float *raw_quats,
                          const float *scales, const float *quats,
                          const float *opacities, const float *v_scales,
                          const float *v_quats, const float *v_opacities,
                          float *v_log_scales, float *v_raw_quats,
                          float *v_logits, gsr_stream_t stream);

/* ---- densification statistics -----------------------------
 * GaussianSplattingModel.after_train (gs_toolkit/models/vanilla_gs.py:344-372) in
 * one launch: for every Gaussian with radii > 0,
 *   xys_grad_norm += |v_xys| (skipped if v_xys is NULL), vis_counts += 1,
 *   max_2dsize = max(max_2dsize, radii * inv_size),   inv_size = 1 / max(W, H).
 * first != 0 is the reference's first call after a refinement (:354-356,
 * `xys_grad_norm = grads; vis_counts = ones`): EVERY Gaussian gets count 1 and its
 * gradient norm (zero for the invisible ones), max_2dsize starts from 0; the three
 * arrays need not be initialised. */
int gsr_densify_stats(int num_points, const float *v_xys, const int32_t *radii,
                      float inv_size, int first, float *xys_grad_norm,
                      int32_t *vis_counts, float *max_2dsize,
                      gsr_stream_t stream);
/* the same with `first` read from device memory (one int32): the call can then sit in
 * a captured HIP graph that is replayed across refinement boundaries */
int gsr_densify_stats_dev(int num_points, const float *v_xys,
                          const int32_t *radii, float inv_size,
                          const int32_t *first, float *xys_grad_norm,
                          int32_t *vis_counts, float *max_2dsize,
                          gsr_stream_t stream);

/* ---- refinement: densify / split / duplicate / cull --------
 * GaussianSplattingModel.refinement_after (gs_toolkit/models/vanilla_gs.py:381-497)
 * with split_gaussians :540-592, dup_gaussians :594-603, cull_gaussians :499-538 and
 * the optimizer surgery dup_in_optim :303-337 / remove_from_optim :282-301, as a
 * plan (decisions + final positions) and ONE compaction launch over all tensors.
 * The host decides the branch from the step counter and fills the config:
 *   densify              step < stop_split_at and step % (reset_alpha_every *
 *                        refine_every) > num_train_data + refine_every  (:390-395);
 *                        0 = cull only (:459-463)
 *   split_by_screen_size step < stop_screen_size_at (:421)
 *   cull_big             step > refine_every * reset_alpha_every (:512)
 *   cull_by_screen_size  step < stop_screen_size_at (:518)
 *   half_max_dim         0.5 * max(W, H) of the last rendered view (:404-408)
 * gsr_refine_plan writes flags[n] (bit 0: the original survives, 1: its split
 * children survive, 2: its duplicate survives, 3: it was split), offsets[n,4] (i32,
 * 16-byte aligned: exclusive prefix counts of those four bits) and counts[4] =
 * {kept originals K0, kept split sources Ks, kept duplicates Kd, split sources}.
 * The output has K0 + n_split_samples * Ks + Kd rows: the caller reads counts back,
 * allocates, and calls gsr_refine_apply, which moves/creates the rows of up to
 * GSR_REFINE_MAX_TENSORS tensors ([n, width] fp32 each) in one launch:
 * originals in order, then split children sample by sample, then duplicates -- the
 * order of the reference's cat + cull.  kind says what new rows hold:
 *   COPY        the parent's row (quats, features_dc, features_rest, opacities)
 *   LOG_SCALES  log(exp(s) / 1.6) for children (and duplicates) of a split Gaussian
 *   MEANS       split children: mean + R(q/|q|) (exp(s) * z), z ~ N(0,1)
 *   MOMENT      zeros (exp_avg / exp_avg_sq of the new rows)
 * z comes from samples [n_split_samples * split sources, 3] (row j * sources +
 * rank, the reference's torch.randn layout) or, if samples is NULL, from
 * Philox4x32-10 with key = seed and counter = (Gaussian index, j, 0, 0) + Box-Muller.
 * vis_counts is int32 (gsr_densify_stats); max_2dsize may be NULL when no
 * screen-size rule is on. */
typedef struct {
  float densify_grad_thresh;
  float densify_size_thresh;
  float split_screen_size;
  float cull_alpha_thresh;
  float cull_scale_thresh;
  float cull_screen_size;
  float half_max_dim;
  int n_split_samples;
  int densify;
  int split_by_screen_size;
  int cull_big;
  int cull_by_screen_size;
} gsr_refine_config;
#define GSR_REFINE_COPY 0
#define GSR_REFINE_LOG_SCALES 1
#define GSR_REFINE_MEANS 2
#define GSR_REFINE_MOMENT 3
#define GSR_REFINE_MAX_TENSORS 24
typedef struct {
  const float *in; /* [n, width] */
  float *out;      /* [K0 + n_split_samples * Ks + Kd, width] */
  int width;
  int kind;
} gsr_refine_tensor;
size_t gsr_refine_workspace_bytes(int num_points);
int gsr_refine_plan(int num_points, const float *log_scales,
                    const float *opacity_logits, const float *xys_grad_norm,
                    const int32_t *vis_counts, const float *max_2dsize,
                    const gsr_refine_config *cfg, uint8_t *flags,
                    int32_t *offsets, int32_t *counts, void *workspace,
                    size_t workspace_bytes, gsr_stream_t stream);
int gsr_refine_apply(int num_points, int n_split_samples, const uint8_t *flags,
                     const int32_t *offsets, const int32_t *counts,
                     const float *log_scales, const float *raw_quats,
                     const float *samples, unsigned long long seed,
                     int num_tensors, const gsr_refine_tensor *tensors,
                     gsr_stream_t stream);

/* ---- optimiser step ------------------------------------
 * Adam over up to GSR_ADAM_MAX_TENSORS tensors in one launch; replaces the
 * per-group torch.optim.Adam objects the toolkit builds
 * (gs_toolkit/engine/optimizers.py:59-196, learning rates and eps = 1e-15 of
 * configs/method_configs.py:47-80).  Rule of torch.optim.Adam with
 * amsgrad = False, weight_decay = 0:  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
 * p -= lr / (1 - b1^step) * m / (sqrt(v) / sqrt(1 - b2^step) + eps).
 * `tensors` is a HOST array; its pointers are device pointers (fp32, n elements
 * each, updated in place: param, exp_avg, exp_avg_sq).  `step` counts from 1.
 * The betas are doubles because torch forms 1 - beta in double before rounding
 * to fp32 (1.f - 0.999f is 1.3e-5 off). */
#define GSR_ADAM_MAX_TENSORS 8
typedef struct {
  float *param;
  const float *grad;
  float *exp_avg;
  float *exp_avg_sq;
  long long n;
  float lr;
} gsr_adam_tensor;
int gsr_adam_step(int num_tensors, const gsr_adam_tensor *tensors,
                  double beta1, double beta2, double eps, long long step,
                  gsr_stream_t stream);

/* ---- deterministic compositing backward ------------------------------------------
 * gsr_rasterize_backward(_rgbd) sum the per-tile contributions to a Gaussian's gradient
 * with float atomics: the result depends on the order in which the tiles' waves retire
 * (differences of a few ulp from run to run).  This variant (16x16 tiles) writes each
 * (tile, list entry) contribution to its own row of a workspace and then sums the rows of
 * every Gaussian in a fixed order, so two runs on the same inputs give bit-identical
 * gradients ("deterministic backward").  It needs what the binning knew:
 * order / cum_sorted of gsr_depth_order (num_bands = gsr_tile_bands) and slot_of_entry of
 * gsr_bin_sorted(_dev); list_capacity = the length gaussian_ids_sorted / slot_of_entry were
 * sized for.  extra / v_output_extra / v_extra NULL: three channels; otherwise the RGB +
 * extra-channel pass of gsr_rasterize_backward_rgbd.  Every output element is written.
 * About 1.4x the time of the atomic version and 48 B of workspace per list entry. */
size_t gsr_rasterize_backward_det_workspace_bytes(int list_capacity);
int gsr_rasterize_backward_det(
    unsigned img_height, unsigned img_width, int num_points, int list_capacity,
    const int32_t *gaussian_ids_sorted, const int32_t *tile_bins,
    const float *xys, const float *conics, const float *colors,
    const float *extra, const float *opacities, const float *background,
    float extra_background, const float *final_Ts, const int32_t *final_idx,
    const float *v_output, const float *v_output_extra,
    const float *v_output_alpha, const int32_t *order,
    const int32_t *cum_sorted, int num_bands, const int32_t *slot_of_entry,
    void *workspace, size_t workspace_bytes, float *v_xy, float *v_conic,
    float *v_colors, float *v_extra, float *v_opacity, gsr_stream_t stream);

/* ---- measurement hook ---------------------------------------------------------
 * counters: two device uint64 (or NULL = off, the default).  While set, the 16x16
 * compositing kernels add the number of list entries they stage to counters[0]
 * (forward) / counters[1] (backward): tiles stop once every pixel is saturated, so
 * this is what a launch really reads of the lists (bench.py prices the roofline on
 * it).  Device-wide setting (synchronises the device); not for production loops. */
int gsr_debug_count_staged(unsigned long long *counters);

/* ---- measurement hook: who ran when ---------------------------------------------
 * records: device buffer of 2 * capacity_waves * 4 uint64 (or NULL = off, the
 * default).  While set, every wave of the 16x16 compositing kernels that runs its
 * tile to the end writes {100-MHz clock at entry, at exit, tile | sub-tile mask <<
 * 32, list length | hardware id << 32} to record blockIdx.x -- forward launches to
 * the first capacity_waves records, backward launches to the second half.  The
 * tail and the imbalance of a launch (tools/exp/wave_trace.py).  Device-wide
 * setting (synchronises the device); not for production loops. */
int gsr_debug_wave_trace(unsigned long long *records, unsigned capacity_waves);

/* ---- box calibration (bench.py `calibration`; not part of the path) ----------
 * Two fixed workloads, timed by the caller with events on `stream` next to the
 * bench's own steps, so that a move of the headline between two GPU leases can be
 * attributed to the box (clock state, memory) or to the code.
 * gsr_calibrate_valu: `workgroups` x 256 lanes each run `iters` rounds of 8
 *   independent non-packed fp32 fma chains; returns the number of VALU
 *   lane-operations launched (8 * iters * 256 * workgroups; -1 on error);
 *   scratch: at least `workgroups` floats (never written).
 * gsr_calibrate_copy: float4 grid-stride copy of `bytes` (multiple of 16). */
long long gsr_calibrate_valu(int iters, int workgroups, float *scratch,
                             gsr_stream_t stream);
int gsr_calibrate_copy(const void *src, void *dst, size_t bytes,
                       gsr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GSRASTER_H_ */
