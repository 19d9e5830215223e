// view.hip -- a whole view as ONE call into the library (forward) and one more (backward).
//
// gs_fused.render_gaussians chains the native calls of a view -- activations, projection, SH
// (+0.5, clamp), reach records, depth order, device-sized tile lists, compositing; and back --
// from Python: ~12 + ~6 ctypes calls of 8-10 us of host time each.  On a small scene (BASELINE
// config 1: 10 k Gaussians at 256 x 256) or a forward-only viewer frame the GPU work is shorter
// than that.  These two entry points run the same sequence of the same exported functions from C:
// the caller allocates every buffer (no allocation in here, as everywhere in this library), fills
// one descriptor and makes one call.  What they replace on the reference's side is the body of
// GaussianSplattingModel.get_outputs between the raw parameters and the images
// (gs_toolkit/models/vanilla_gs.py:765-857) and its autograd backward.
#include "gsr_common.h"

namespace {
#define GSR_TRY(expr)          \
  do {                         \
    const int rc_ = (expr);    \
    if (rc_ != GSR_OK) return rc_; \
  } while (0)
}  // namespace

GSR_EXPORT int gsr_view_forward(const gsr_view_desc *v, gsr_stream_t stream) {
  GSR_REQUIRE(v != nullptr, "view_forward: null descriptor");
  const int n = v->num_points;
  GSR_REQUIRE(n >= 0 && v->capacity >= 1, "view_forward: bad sizes");
  GSR_REQUIRE(v->sh_degree >= 0 && v->sh_degree <= 3 && v->sh_degree_to_use <= v->sh_degree, "view_forward: SH degree");
  const int tiles_x = (v->img_width + 15) / 16, tiles_y = (v->img_height + 15) / 16;
  GSR_TRY(gsr_activate_forward(n, v->means, v->log_scales, v->raw_quats, v->logits, v->campos, v->scales, v->quats,
                               v->opac, v->dirs, stream));
  GSR_TRY(gsr_project_forward(n, v->means, v->scales, v->glob_scale, v->quats, v->viewmat, v->projmat, v->fx, v->fy,
                              v->cx, v->cy, (unsigned)v->img_height, (unsigned)v->img_width, 16, v->clip_thresh,
                              v->cov3d, v->xys, v->depths, v->radii, v->conics, v->comp, v->tiles, stream));
  GSR_TRY(gsr_sh_forward_split((unsigned)n, (unsigned)v->sh_degree, (unsigned)v->sh_degree_to_use, v->dirs,
                               v->features_dc, v->features_rest, v->colors, 0.5f, 1, stream));
  // lists: with counts (short lists: single-pass scatter) or without (two-level partition), as the caller
  // decided with gsr_bin_sorted_needs_counts when it sized the buffers
  if (v->counts) {
    GSR_TRY(gsr_count_reach(n, v->xys, v->radii, v->conics, v->opac, tiles_x, tiles_y, 1, v->counts, v->reach_records,
                            stream));
    GSR_TRY(gsr_depth_order(n, v->depths, v->radii, v->counts, 1, v->order, v->cum, v->sort_ws, v->sort_ws_bytes,
                            stream));
  } else {
    GSR_TRY(gsr_reach_records_depth_order(n, v->xys, v->radii, v->conics, v->opac, v->depths, tiles_x, tiles_y,
                                          v->reach_records, v->order, v->sort_ws, v->sort_ws_bytes, stream));
  }
  GSR_TRY(gsr_bin_sorted_dev(n, v->capacity, v->order, v->counts ? v->cum : nullptr, v->xys, v->radii,
                             v->reach_records, tiles_x, tiles_y, 16, 1, v->ids, v->tile_bins, v->count_out, nullptr,
                             v->bin_ws, v->bin_ws_bytes, stream));
  GSR_REQUIRE(!v->render_depth || v->out_depth != nullptr, "view_forward: render_depth without out_depth");
  GSR_TRY(gsr_rasterize_forward_seg(tiles_x, tiles_y, (unsigned)v->img_width, (unsigned)v->img_height, v->ids,
                                    v->tile_bins, v->xys, v->conics, v->colors, v->render_depth ? v->depths : nullptr,
                                    v->opac, v->background, 0.f, v->out_img, v->render_depth ? v->out_depth : nullptr,
                                    v->final_Ts, v->final_idx, v->deep_tile_threshold, v->out_alpha, v->zero_ptr,
                                    v->zero_bytes, v->segments, v->segment_min_entries, v->seg_ws, v->seg_ws_bytes,
                                    stream));
  return GSR_OK;
}

// gsr_rasterize_gaussians_forward -- everything `_RasterizeGaussians.forward` does on the device
// (rasterizer/rasterize.py:89-170 of the reference: bin_and_sort_gaussians + rasterize_forward) as ONE call:
// reach records + depth order (or the records alone, around an order the caller already has), device-sized tile
// lists, compositing.  The unchanged models block the host right in front of this op (`assert (num_tiles_hit >
// 0).any()`, vanilla_gs.py:811): with one call instead of four, ~10 us of host time instead of ~40 stand between
// that read-back and a busy GPU.
GSR_EXPORT int gsr_rasterize_gaussians_forward(const gsr_raster_desc *v, gsr_stream_t stream) {
  GSR_REQUIRE(v != nullptr, "rasterize_gaussians_forward: null descriptor");
  const int n = v->num_points;
  GSR_REQUIRE(n >= 1 && v->capacity >= 1, "rasterize_gaussians_forward: bad sizes");
  const int tiles_x = (v->img_width + 15) / 16, tiles_y = (v->img_height + 15) / 16;
  const int32_t *order = v->order_ready;
  if (v->counts) {
    GSR_REQUIRE(order == nullptr, "rasterize_gaussians_forward: a ready-made depth order goes with lists without counts");
    GSR_TRY(gsr_count_reach(n, v->xys, v->radii, v->conics, v->opac, tiles_x, tiles_y, 1, v->counts, v->reach_records,
                            stream));
    GSR_TRY(gsr_depth_order(n, v->depths, v->radii, v->counts, 1, v->order, v->cum, v->sort_ws, v->sort_ws_bytes,
                            stream));
    order = v->order;
  } else if (order != nullptr) {
    GSR_TRY(gsr_count_reach(n, v->xys, v->radii, v->conics, v->opac, tiles_x, tiles_y, 1, nullptr, v->reach_records,
                            stream));
  } else {
    GSR_TRY(gsr_reach_records_depth_order(n, v->xys, v->radii, v->conics, v->opac, v->depths, tiles_x, tiles_y,
                                          v->reach_records, v->order, v->sort_ws, v->sort_ws_bytes, stream));
    order = v->order;
  }
  GSR_TRY(gsr_bin_sorted_dev(n, v->capacity, order, v->counts ? v->cum : nullptr, v->xys, v->radii, v->reach_records,
                             tiles_x, tiles_y, 16, 1, v->ids, v->tile_bins, v->count_out, nullptr, v->bin_ws,
                             v->bin_ws_bytes, stream));
  int deep = v->deep_tile_threshold;
  if (deep > 0 && (deep & GSR_DEEP_ORDERED) && !(deep & GSR_DEEP_PREBUILT) && v->deep_tile_threshold_backward > 0) {
    // both job orders in one launch, right behind the lists (also when only the lists were asked for: lists built
    // ahead of time on a side stream come with their orders, and the compositing call that takes them later -- on
    // the critical path behind the models' read-back -- launches nothing in front of its kernel)
    GSR_TRY(gsr_tile_jobs_build(tiles_x, tiles_y, v->tile_bins, deep, v->deep_tile_threshold_backward, stream));
    deep |= GSR_DEEP_PREBUILT;
  }
  if (v->out_img == nullptr) return GSR_OK;  // the lists only (built ahead of time, composited by a later call)
  GSR_REQUIRE(v->extra == nullptr || v->out_extra != nullptr, "rasterize_gaussians_forward: extra channel without out_extra");
  GSR_TRY(gsr_rasterize_forward_seg(tiles_x, tiles_y, (unsigned)v->img_width, (unsigned)v->img_height, v->ids,
                                    v->tile_bins, v->xys, v->conics, v->colors, v->extra, v->opac, v->background,
                                    v->extra_background, v->out_img, v->extra ? v->out_extra : nullptr, v->final_Ts,
                                    v->final_idx, deep, v->out_alpha, v->zero_ptr, v->zero_bytes,
                                    v->segments, v->segment_min_entries, v->seg_ws, v->seg_ws_bytes, stream));
  return GSR_OK;
}

GSR_EXPORT int gsr_view_backward(const gsr_view_desc *v, const gsr_view_grads *g, gsr_stream_t stream) {
  GSR_REQUIRE(v != nullptr && g != nullptr, "view_backward: null descriptor");
  const int n = v->num_points;
  const size_t N = (size_t)n;
  // accumulators laid out as the Python binding lays them out: v_xy | v_conic | v_colors | v_opacity [| v_extra]
  float *acc = g->accumulators;
  GSR_REQUIRE(acc != nullptr && g->v_img != nullptr, "view_backward: null pointer");
  float *v_xy = acc, *v_conic = acc + 2 * N, *v_colors = acc + 5 * N, *v_opac = acc + 8 * N, *v_extra = acc + 9 * N;
  GSR_REQUIRE(!v->render_depth || g->v_depth != nullptr, "view_backward: render_depth without its cotangent");
  GSR_TRY(gsr_rasterize_backward_seg((unsigned)v->img_height, (unsigned)v->img_width, n, v->ids, v->tile_bins, v->xys,
                                     v->conics, v->colors, v->render_depth ? v->depths : nullptr, v->opac,
                                     v->background, 0.f, v->final_Ts, v->final_idx, g->v_img,
                                     v->render_depth ? g->v_depth : nullptr, g->v_alpha, v_xy, v_conic, v_colors,
                                     v->render_depth ? v_extra : nullptr, v_opac, v->deep_tile_threshold,
                                     g->accumulators_zeroed, v->segments, v->segment_min_entries, v->seg_ws,
                                     v->seg_ws_bytes, stream));
  if (g->stats_first != nullptr)
    GSR_TRY(gsr_densify_stats_dev(n, v_xy, v->radii, g->stats_inv_size, g->stats_first, g->xys_grad_norm,
                                  g->vis_counts, g->max_2dsize, stream));
  // v_dc == NULL: the caller forms the SH gradient itself from the colour cotangents left in the accumulators (data
  // parallel: gathered over the ranks, gsr_sh_backward_views)
  if (g->v_dc != nullptr)
    GSR_TRY(gsr_sh_backward_split((unsigned)n, (unsigned)v->sh_degree, (unsigned)v->sh_degree_to_use, v->dirs,
                                  v_colors, v->colors, g->v_dc, g->v_rest, stream));
  GSR_TRY(gsr_project_backward(n, v->means, v->scales, v->glob_scale, v->quats, v->viewmat, v->projmat, v->fx, v->fy,
                               v->cx, v->cy, (unsigned)v->img_height, (unsigned)v->img_width, v->cov3d, v->radii,
                               v->conics, v->comp, v_xy, v->render_depth ? v_extra : nullptr, v_conic, nullptr,
                               g->tmp_v_cov2d, g->tmp_v_cov3d, g->v_means, g->tmp_v_scales, g->tmp_v_quats, stream));
  GSR_TRY(gsr_activate_backward(n, v->raw_quats, v->scales, v->quats, v->opac, g->tmp_v_scales, g->tmp_v_quats, v_opac,
                                g->v_log_scales, g->v_raw_quats, g->v_logits, stream));
  return GSR_OK;
}
