"""One render op: what ``GaussianSplattingModel.get_outputs`` does between the raw
parameters and the images, as a single autograd node.

The toolkit's models (gs_toolkit/models/vanilla_gs.py:765-857, depth_gs.py:225-363) run,
per view: exp / normalise / sigmoid, the view directions, ``torch.cat`` of the SH
features, ``project_gaussians``, ``spherical_harmonics``, ``clamp(+0.5)``,
``rasterize_gaussians`` (and a second one for depth) -- ~25 torch ops and four custom
autograd nodes, each with its own saved-tensor bookkeeping.  ``render_gaussians`` chains
the same native calls (include/gsraster.h) inside ONE ``torch.autograd.Function``:

  forward : gsr_activate_forward -> gsr_project_forward -> gsr_sh_forward_split (+0.5,
            clamp) -> count_reach / depth_order / bin_sorted_dev -> gsr_rasterize_forward
            (or _rgbd: RGB + depth from one compositing pass)
  backward: gsr_rasterize_backward(_rgbd) -> gsr_sh_backward_split -> gsr_project_backward
            -> gsr_activate_backward, and the densification statistics of ``after_train``
            (vanilla_gs.py:344-372) straight from the screen-space gradient.

Nothing in it reads device memory back: the tile lists are sized by ``capacity`` (the
caller's job: `ListCapacity` below keeps a running estimate and checks the real count one
view later), which makes the whole view -- forward, loss, backward -- capturable in a HIP
graph (`ViewGraph`).  Same values as the separate ops (tests/test_gpu_render.py): images
bit-identical, gradients equal up to the order of the float atomics.
fp32 CUDA tensors; 16x16 tiles; SH degree 0-3 storage (K = 1, 4, 9, 16).
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, Optional

import torch
from torch import Tensor
from torch.autograd import Function

import rasterizer.cuda as _C
from rasterizer.cuda import _call, _ptr, _stream

_f32, _i32 = torch.float32, torch.int32
BLOCK = 16


class _ViewDesc(C.Structure):  # gsr_view_desc (include/gsraster.h)
    _fields_ = ([(k, C.c_int) for k in ("num_points", "sh_degree", "sh_degree_to_use", "render_depth", "img_height",
                                        "img_width")]
                + [(k, C.c_float) for k in ("fx", "fy", "cx", "cy", "glob_scale", "clip_thresh")]
                + [(k, C.c_int) for k in ("capacity", "deep_tile_threshold")]
                + [(k, C.c_void_p) for k in ("means", "log_scales", "raw_quats", "logits", "features_dc", "features_rest",
                                             "viewmat", "projmat", "campos", "background", "scales", "quats", "opac",
                                             "dirs", "cov3d", "xys", "depths", "radii", "conics", "comp", "tiles",
                                             "colors", "reach_records", "counts", "order", "cum", "ids", "tile_bins",
                                             "count_out", "sort_ws")]
                + [("sort_ws_bytes", C.c_size_t), ("bin_ws", C.c_void_p), ("bin_ws_bytes", C.c_size_t)]
                + [(k, C.c_void_p) for k in ("out_img", "out_depth", "final_Ts", "final_idx", "out_alpha", "zero_ptr")]
                + [("zero_bytes", C.c_size_t), ("segments", C.c_int), ("segment_min_entries", C.c_int),
                   ("seg_ws", C.c_void_p), ("seg_ws_bytes", C.c_size_t)])


class _ViewGrads(C.Structure):  # gsr_view_grads
    _fields_ = ([(k, C.c_void_p) for k in ("v_img", "v_alpha", "v_depth", "accumulators")]
                + [("accumulators_zeroed", C.c_int), ("stats_first", C.c_void_p), ("stats_inv_size", C.c_float)]
                + [(k, C.c_void_p) for k in ("xys_grad_norm", "vis_counts", "max_2dsize", "tmp_v_cov2d", "tmp_v_cov3d",
                                             "tmp_v_scales", "tmp_v_quats", "v_means", "v_log_scales", "v_raw_quats",
                                             "v_logits", "v_dc", "v_rest")])


def _segments(spec, capacity, dev):
    """(segments, minimum entries, workspace tensor or None) of the view's compositing
    (rasterizer.cuda.depth_segments).  The workspace is scratch of ONE call; the caller keeps the tensor until the
    call has been issued."""
    tb = spec.tile_bounds
    return _C._forward_segments(capacity, tb[0] * tb[1], spec.height, spec.width, dev)


def _segments_backward(spec, capacity, dev):
    """The backward's runs (not capped by GSR_DEPTH_SEGMENTS_FWD) and its workspace of (runs - 1) maps per pixel."""
    tb = spec.tile_bounds
    segs, seg_min = _C.depth_segments(capacity, tb[0] * tb[1])
    if segs < 2:
        return 0, 0, None
    return segs, seg_min, torch.empty(((segs - 1) * spec.height * spec.width * 8,), dtype=torch.uint8, device=dev)


def _seg_fields(seg):
    segs, seg_min, ws = seg
    if ws is None:
        return 0, 0, None, 0
    return segs, seg_min, ws.data_ptr(), ws.numel()


_ws_bytes_cache = {}


def _workspace_bytes(n, capacity, tb):
    key = (n, capacity, tb[0], tb[1])
    v = _ws_bytes_cache.get(key)
    if v is None:
        lib = _C._lib()
        v = (int(lib.gsr_depth_order_workspace_bytes(C.c_int(n), C.c_int(1))),
             int(lib.gsr_bin_sorted_workspace_bytes(C.c_int(n), C.c_int(capacity), C.c_int(tb[0]), C.c_int(tb[1]))),
             int(lib.gsr_reach_record_bytes()),
             not _C.lists_need_counts(n, capacity, tb, device_sized=True))
        if len(_ws_bytes_cache) > 64:
            _ws_bytes_cache.clear()
        _ws_bytes_cache[key] = v
    return v


@dataclass(frozen=True)
class ViewSpec:
    """The per-view scalars (baked into a captured graph: one graph per distinct spec)."""
    height: int
    width: int
    fx: float
    fy: float
    cx: float
    cy: float
    sh_degree_to_use: int
    render_depth: bool = False
    clip_thresh: float = 0.01
    glob_scale: float = 1.0

    @property
    def tile_bounds(self):
        return ((self.width + BLOCK - 1) // BLOCK, (self.height + BLOCK - 1) // BLOCK, 1)


class DensifyStats:
    """``xys_grad_norm`` / ``vis_counts`` / ``max_2dsize`` of after_train (vanilla_gs.py:344-372),
    updated by the backward of `render_gaussians` itself (one launch, no retain_grad)."""

    def __init__(self, n: int, device, max_dim: int):
        self.xys_grad_norm = torch.zeros(n, device=device)
        self.vis_counts = torch.zeros(n, device=device, dtype=_i32)
        self.max_2dsize = torch.zeros(n, device=device)
        self.max_dim = int(max_dim)
        # 1 = the next update is the first after a refinement (vanilla_gs.py:354-356); kept on the
        # DEVICE so that a captured graph replays correctly across refinement boundaries
        self.first = torch.ones(1, device=device, dtype=_i32)
        self.enabled = True

    def as_tuple(self):
        return self.xys_grad_norm, self.vis_counts, self.max_2dsize

    def restart(self):
        self.first.fill_(1)


class _Render(Function):
    @staticmethod
    def forward(ctx, means, log_scales, raw_quats, logits, features_dc, features_rest, viewmat, projmat, campos,
                background, spec: ViewSpec, capacity: int, count_out: Tensor, stats: Optional[DensifyStats],
                sh_collector=None):
        n = means.shape[0]
        dev = means.device
        H, W = spec.height, spec.width
        tb = spec.tile_bounds
        degree = {1: 0, 4: 1, 9: 2, 16: 3}[features_rest.shape[1] + 1]
        with torch.cuda.device(dev):
            e = lambda shape, dt=_f32: torch.empty(shape, dtype=dt, device=dev)
            scales, quats, opac, dirs = e((n, 3)), e((n, 4)), torch.empty_like(logits), e((n, 3))
            cov3d, xys, depths, radii = e((n, 6)), e((n, 2)), e((n,)), e((n,), _i32)
            conics, comp, tiles, colors = e((n, 3)), e((n,)), e((n,), _i32), e((n, 3))
            sort_b, bin_b, rec_b, lean = _workspace_bytes(n, capacity, tb)
            recs = e((n, rec_b), torch.uint8)
            counts = None if lean else e((n,), _i32)
            cum = None if lean else e((n,), _i32)
            order, ids, bins = e((n,), _i32), e((capacity,), _i32), _C.alloc_tile_bins(tb, dev)
            sort_ws, bin_ws = e((max(sort_b, 1),), torch.uint8), e((max(bin_b, 1),), torch.uint8)
            img, Ts, idx, alpha = e((H, W, 3)), e((H, W)), e((H, W), _i32), e((H, W))
            dep = e((H, W)) if spec.render_depth else None
            # alpha = 1 - T and the backward's cleared accumulators come out of the compositing launch
            acc = _C.backward_accumulators(n, 4 if spec.render_depth else 3, dev) if any(ctx.needs_input_grad[:6]) else None
            p = lambda t: None if t is None else t.data_ptr()
            seg = _segments(spec, capacity, dev)
            desc = _ViewDesc(n, degree, spec.sh_degree_to_use, int(spec.render_depth), H, W, spec.fx, spec.fy, spec.cx,
                             spec.cy, spec.glob_scale, spec.clip_thresh, capacity,
                             _C.deep_arg(bins, capacity, tb[0] * tb[1], tile_bounds=tb),
                             p(means), p(log_scales), p(raw_quats), p(logits), p(features_dc), p(features_rest),
                             p(viewmat), p(projmat), p(campos), p(background), p(scales), p(quats), p(opac), p(dirs),
                             p(cov3d), p(xys), p(depths), p(radii), p(conics), p(comp), p(tiles), p(colors), p(recs),
                             p(counts), p(order), p(cum), p(ids), p(bins), p(count_out), p(sort_ws), sort_b,
                             p(bin_ws), bin_b, p(img), p(dep), p(Ts), p(idx), p(alpha), p(acc),
                             0 if acc is None else acc.numel() * 4, *_seg_fields(seg))
            _call("gsr_view_forward", C.byref(desc), _stream(dev))
            del seg
        ctx.spec, ctx.stats, ctx.degree = spec, stats, degree
        ctx.sh_collector = sh_collector
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(means, raw_quats, features_dc, features_rest, viewmat, projmat, background, scales, quats,
                              opac, dirs, cov3d, xys, depths, radii, conics, comp, colors, ids, bins, Ts, idx)
        ctx.mark_non_differentiable(radii)
        ctx.accumulators = acc
        ctx.capacity = capacity
        if dep is None:
            return img, alpha, radii
        return img, alpha, radii, dep

    @staticmethod
    def backward(ctx, v_img, v_alpha, _v_radii, v_dep=None):
        (means, raw_quats, features_dc, features_rest, viewmat, projmat, background, scales, quats, opac, dirs, cov3d,
         xys, depths, radii, conics, comp, colors, ids, bins, Ts, idx) = ctx.saved_tensors
        spec, stats = ctx.spec, ctx.stats
        n = means.shape[0]
        dev = means.device
        H, W = spec.height, spec.width
        tb = spec.tile_bounds
        with torch.cuda.device(dev):
            if v_img is None:
                v_img = torch.zeros(H, W, 3, device=dev)
            v_img = v_img.contiguous()
            v_a = None if v_alpha is None else v_alpha.contiguous()
            if spec.render_depth:
                v_dep = torch.zeros(H, W, device=dev) if v_dep is None else v_dep.contiguous()
            acc, ctx.accumulators = ctx.accumulators, None
            zeroed = acc is not None
            if acc is None:  # a second backward (retain_graph): the kernels clear their own
                acc = _C.backward_accumulators(n, 4 if spec.render_depth else 3, dev)
            e = lambda shape: torch.empty(shape, dtype=_f32, device=dev)
            t_cov2d, t_cov3d, t_vs, t_vq = e((n, 3)), e((n, 6)), e((n, 3)), e((n, 4))
            v_means, g_s, g_q, g_o = e((n, 3)), e((n, 3)), e((n, 4)), torch.empty_like(opac)
            collector = ctx.sh_collector
            if collector is None:
                v_dc, v_rest = torch.empty_like(features_dc), torch.empty_like(features_rest)
            else:  # data parallel: the SH gradient is formed from the ranks' gathered colour cotangents
                v_dc = v_rest = None
            use_stats = stats is not None and stats.enabled
            p = lambda t: None if t is None else t.data_ptr()
            seg = _segments_backward(spec, ctx.capacity, dev)
            desc = _ViewDesc(n, ctx.degree, spec.sh_degree_to_use, int(spec.render_depth), H, W, spec.fx, spec.fy,
                             spec.cx, spec.cy, spec.glob_scale, spec.clip_thresh, ctx.capacity,
                             _C.deep_arg(bins, ctx.capacity, tb[0] * tb[1], backward=True, tile_bounds=tb),
                             p(means), None, p(raw_quats), None, p(features_dc), p(features_rest), p(viewmat),
                             p(projmat), None, p(background), p(scales), p(quats), p(opac), p(dirs), p(cov3d), p(xys),
                             p(depths), p(radii), p(conics), p(comp), None, p(colors), None, None, None, None, p(ids),
                             p(bins), None, None, 0, None, 0, None, None, p(Ts), p(idx), None, None, 0,
                             *_seg_fields(seg))
            grads = _ViewGrads(p(v_img), p(v_a), p(v_dep) if spec.render_depth else None, p(acc), int(zeroed),
                               p(stats.first) if use_stats else None, (1.0 / stats.max_dim) if use_stats else 0.0,
                               p(stats.xys_grad_norm) if use_stats else None, p(stats.vis_counts) if use_stats else None,
                               p(stats.max_2dsize) if use_stats else None, p(t_cov2d), p(t_cov3d), p(t_vs), p(t_vq),
                               p(v_means), p(g_s), p(g_q), p(g_o), p(v_dc), p(v_rest))
            _call("gsr_view_backward", C.byref(desc), C.byref(grads), _stream(dev))
            del seg
            if use_stats:
                stats.first.zero_()
            if collector is not None:
                # the colour cotangents the compositing backward left in the accumulators, cut where the forward's
                # clamp cut the colour (marked -0.0), as gsr_sh_backward_split would have cut them
                v_col = acc[5 * n:8 * n].view(n, 3)
                cut = colors.view(torch.int32) == -2147483648
                collector.offer(torch.where(cut, torch.zeros((), dtype=_f32, device=dev), v_col).contiguous())
        return (v_means, g_s, g_q, g_o, v_dc, v_rest) + (None,) * 9


def render_gaussians(means: Tensor, log_scales: Tensor, raw_quats: Tensor, opacity_logits: Tensor,
                     features_dc: Tensor, features_rest: Tensor, viewmat: Tensor, projmat: Tensor, campos: Tensor,
                     background: Tensor, spec: ViewSpec, capacity: int, count_out: Optional[Tensor] = None,
                     stats: Optional[DensifyStats] = None, sh_collector=None) -> Dict[str, Optional[Tensor]]:
    """The raw parameters of a Gaussian model -> ``{"rgb" [H,W,3] (not clamped at 1), "alpha"
    [H,W], "depth" [H,W] or None (accumulated, not divided by alpha), "radii" [N] i32,
    "count" int32[1]}``.

    viewmat: the top 3x4 (or 4x4) world->camera matrix, projmat 4x4 ``P @ V``, campos [3],
    all on the device.  ``capacity``: entries the tile lists are sized for; ``count`` receives
    the entries the view really needs -- if it exceeds the capacity the lists were cut and
    the result is incomplete: render again with a larger capacity (`ListCapacity`).
    ``count_out`` may be a caller-owned int32[1] (device, or pinned host memory).    ``sh_collector`` (data parallel, `harness.parallel.GradientExchange.begin_sh_views`): the backward skips its SH
    backward, hands the colour cotangents [N,3] to ``sh_collector.offer`` and returns no gradient for
    ``features_dc`` / ``features_rest`` -- the exchange forms it from all ranks' cotangents.
    """
    n = means.shape[0]
    if features_dc.shape != (n, 3) or features_rest.dim() != 3 or features_rest.shape[1] + 1 not in (1, 4, 9, 16):
        raise ValueError("features_dc [N,3] and features_rest [N,K-1,3] with K in (1, 4, 9, 16) expected")
    if log_scales.shape != (n, 3) or raw_quats.shape != (n, 4) or opacity_logits.numel() != n:
        raise ValueError("expected scales [N,3], quats [N,4], opacities [N,1]")
    if capacity < 1:
        raise ValueError("capacity must be positive")
    if count_out is None:
        count_out = torch.empty(1, dtype=_i32, device=means.device)
    vm = viewmat[:3, :] if viewmat.shape[0] == 4 else viewmat
    out = _Render.apply(means.contiguous(), log_scales.contiguous(), raw_quats.contiguous(),
                        opacity_logits.contiguous(), features_dc.contiguous(), features_rest.contiguous(),
                        vm.contiguous(), projmat.contiguous(), campos.contiguous(), background.contiguous(), spec,
                        int(capacity), count_out, stats, sh_collector)
    return {"rgb": out[0], "alpha": out[1], "radii": out[2], "depth": out[3] if spec.render_depth else None,
            "count": count_out}


@dataclass
class ListCapacity:
    """Keeps the tile lists large enough without waiting for the GPU: ``capacity`` for the
    next view, and a check of a PREVIOUS view's real count (read from pinned memory
    once its event has fired).  ``grow`` is the head-room over the largest count seen."""
    capacity: int = 1 << 20
    grow: float = 1.5
    _pending: list = field(default_factory=list)

    def slot(self, device) -> Tensor:
        """A pinned int32[1] for `render_gaussians(count_out=...)`; call `submitted()` after
        the render has been enqueued."""
        return torch.zeros(1, dtype=_i32).pin_memory()

    def submitted(self, slot: Tensor, capacity_used: int, device) -> None:
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        self._pending.append((ev, slot, capacity_used))

    def overflowed(self, wait: bool = False) -> bool:
        """Did any finished view need more entries than it was given?  Grows `capacity`."""
        bad = False
        keep = []
        for ev, slot, cap in self._pending:
            if wait:
                ev.synchronize()
            if not ev.query():
                keep.append((ev, slot, cap))
                continue
            need = int(slot[0])
            if need > cap:
                bad = True
            want = int(self.grow * need) + 65536
            if want > self.capacity:
                self.capacity = (want + (1 << 20) - 1) & ~((1 << 20) - 1)
        self._pending = keep
        return bad


class ViewGraph:
    """One HIP graph per view: render -> loss -> backward of a Gaussian model captured once
    (``torch.cuda.CUDAGraph`` = hipGraph on ROCm) and replayed for every view with the same
    `ViewSpec`, number of Gaussians and list capacity.  What varies per view -- the camera
    matrices and position, the target image(s) -- is copied into static buffers before
    the replay; the parameter gradients land in static ``.grad`` tensors that the optimizer
    then reads as usual.  The graph holds ~45 kernel nodes that are otherwise ~45 launches
    from Python through ctypes / autograd.

    ``loss_fn(out, targets) -> scalar`` with ``out`` the dict of `render_gaussians` and
    ``targets`` the tuple of static target tensors.  After a replay, ``count_host[0]`` (pinned)
    holds the list entries the view needed: `fits()` tells whether they fitted the capacity
    the graph was captured with -- if not, the result of that view is incomplete and the
    owner re-captures with a larger capacity (`ListCapacity`)."""

    def __init__(self, params: Dict[str, Tensor], spec: ViewSpec, capacity: int, loss_fn, background: Tensor,
                 target_shapes, stats: Optional[DensifyStats] = None):
        self.params, self.spec, self.capacity, self.loss_fn, self.stats = params, spec, int(capacity), loss_fn, stats
        dev = params["means"].device
        self.device = dev
        self.background = background
        self.viewmat = torch.zeros(3, 4, device=dev)
        self.projmat = torch.zeros(4, 4, device=dev)
        self.campos = torch.zeros(3, device=dev)
        self.targets = tuple(torch.zeros(s, device=dev) for s in target_shapes)
        self.count_dev = torch.zeros(1, dtype=_i32, device=dev)
        self.count_host = torch.zeros(1, dtype=_i32).pin_memory()
        self.graph = None
        self.out = None
        self.loss = None

    def _set_inputs(self, viewmat, projmat, campos, targets):
        self.viewmat.copy_(viewmat[:3, :])
        self.projmat.copy_(projmat)
        self.campos.copy_(campos)
        for dst, src in zip(self.targets, targets):
            dst.copy_(src)

    def _step(self):
        p = self.params
        out = render_gaussians(p["means"], p["scales"], p["quats"], p["opacities"], p["features_dc"],
                               p["features_rest"], self.viewmat, self.projmat, self.campos, self.background, self.spec,
                               self.capacity, count_out=self.count_dev, stats=self.stats)
        loss = self.loss_fn(out, self.targets)
        loss.backward()
        _C.publish_int32(self.count_dev, self.count_host)
        return out, loss

    def capture(self, viewmat, projmat, campos, targets) -> None:
        """Warm up eagerly on a side stream (allocator, lazy initialisation), then capture."""
        self._set_inputs(viewmat, projmat, campos, targets)
        first_flag = None if self.stats is None else self.stats.first.clone()
        saved = None if self.stats is None else [t.clone() for t in self.stats.as_tuple()]
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(2):
                for t in self.params.values():
                    t.grad = None
                self._step()
        torch.cuda.current_stream(self.device).wait_stream(side)
        for t in self.params.values():
            t.grad = None
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out, self.loss = self._step()
        if self.stats is not None:  # the warm-up and the capture must not count as views
            self.stats.first.copy_(first_flag)
            for dst, src in zip(self.stats.as_tuple(), saved):
                dst.copy_(src)

    def replay(self, viewmat, projmat, campos, targets):
        """-> (loss, out): static tensors, overwritten by the next replay."""
        self._set_inputs(viewmat, projmat, campos, targets)
        self.graph.replay()
        return self.loss, self.out

    def fits(self) -> bool:
        """After a synchronisation point: did the last replayed view fit the capacity?"""
        return int(self.count_host[0]) <= self.capacity
