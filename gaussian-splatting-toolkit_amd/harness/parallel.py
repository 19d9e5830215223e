"""Per-view data parallelism for Gaussian training.

Every rank holds a full replica of the Gaussian parameters and renders a
different camera; after the backward the parameter gradients are summed across
ranks and (optionally) averaged.  The reference gets this from
``DistributedDataParallel`` (gs_toolkit/pipelines/base_pipeline.py:202-207),
which breaks as soon as densification replaces the ``nn.Parameter`` objects, so
the exchange is explicit here: 59 floats = 236 B per Gaussian at SH degree 3,
reduced either as one flat buffer or tensor by tensor in place (see
`allreduce_gradients`).  On a ROCm build ``backend="nccl"`` is RCCL over xGMI;
the messages are large (the SH block alone is 180 MB at 1 M Gaussians), which is
what lets RCCL spread a collective over all seven links of a GPU -- DDP's default
25 MB buckets would turn the same volume into ~10 smaller rings.
"""
from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


def flatten_grads(params: Sequence[torch.Tensor]) -> torch.Tensor:
    """One contiguous fp32 buffer holding every parameter's gradient (zeros
    for parameters that received none)."""
    total = sum(p.numel() for p in params)
    flat = torch.empty(total, dtype=torch.float32, device=params[0].device)
    off = 0
    for p in params:
        n = p.numel()
        if p.grad is None:
            flat[off:off + n].zero_()
        else:
            flat[off:off + n].copy_(p.grad.reshape(-1))
        off += n
    return flat


def unflatten_to_grads(flat: torch.Tensor, params: Sequence[torch.Tensor]) -> None:
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n


def allreduce_gradients(params: Sequence[torch.Tensor], average: bool = True, group=None,
                        flat: Optional[bool] = None) -> Optional[torch.Tensor]:
    """Sum (or average) the gradients of `params` over all ranks; the result
    replaces ``p.grad``.

    flat=True : pack everything into one fp32 buffer, ONE all-reduce, unpack (two
                extra passes over the 236 B/Gaussian, but a single message);
    flat=False: all-reduce each gradient tensor in place (no copies; six messages,
                the 180 B/Gaussian `features_rest` one dominating).
    Default: in place on RCCL (the copies cost more than five extra launches
    there), flat on gloo/CPU."""
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    if flat is None:
        flat = not (distributed and dist.get_backend(group) == "nccl")
    if not flat:
        ws = dist.get_world_size(group) if distributed else 1
        for p in params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            if distributed:
                _allreduce_inplace(p.grad, average, ws, group)
        return None
    buf = flatten_grads(params)
    if distributed:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        if average:
            buf.div_(dist.get_world_size(group))
    unflatten_to_grads(buf, params)
    return buf


_capabilities = {}


def collective_capabilities(group=None, device=None) -> dict:
    """What the process group's backend can do, found out ONCE, synchronously, on a 4-element message, before any
    gradient is on the wire -- never by catching an exception around a real (asynchronous) collective: an RCCL
    error on an async collective surfaces at `wait()`, after the communicator is already unusable.

    COLLECTIVE: every rank of `group` must call this at the same point (GradientExchange / ShardedAdam do, in
    their constructors).  -> {"avg": ReduceOp.AVG works for all_reduce and reduce_scatter_tensor,
    "gather_into_tensor": all_gather_into_tensor exists, "reduce_scatter_tensor": ...}.  Every rank runs the same
    library and probes the same calls, so every rank gets the same answers."""
    if not (dist.is_available() and dist.is_initialized()):
        return {"avg": False, "gather_into_tensor": False, "reduce_scatter_tensor": False}
    backend = dist.get_backend(group)
    key = (backend, id(group))
    caps = _capabilities.get(key)
    if caps is not None:
        return caps
    ws = dist.get_world_size(group)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")

    def works(fn):
        try:
            fn()
            if device.type == "cuda":
                torch.cuda.synchronize(device)
            return True
        except (RuntimeError, ValueError, AttributeError, NotImplementedError):
            return False

    one = torch.ones(4 * ws, dtype=torch.float32, device=device)
    shard = torch.empty(4, dtype=torch.float32, device=device)
    caps = {
        # gloo has no AVG: it raises at issue, on every rank alike (nothing was enqueued)
        "avg": backend == "nccl" and works(lambda: dist.all_reduce(one.clone(), op=dist.ReduceOp.AVG, group=group)),
        "gather_into_tensor": works(lambda: dist.all_gather_into_tensor(one.clone(), shard.zero_(), group=group)),
        "reduce_scatter_tensor": works(lambda: dist.reduce_scatter_tensor(shard, one.clone(), op=dist.ReduceOp.SUM,
                                                                          group=group)),
    }
    if caps["avg"] and caps["reduce_scatter_tensor"]:
        caps["avg"] = works(lambda: dist.reduce_scatter_tensor(shard, one.clone(), op=dist.ReduceOp.AVG, group=group))
    _capabilities[key] = caps
    return caps


def _allreduce_inplace(t: torch.Tensor, average: bool, ws: int, group) -> None:
    """RCCL averages inside the collective (ReduceOp.AVG): no extra pass over the
    192-MB SH gradient to divide it; SUM + div_ where `collective_capabilities` found no AVG."""
    if average and collective_capabilities(group, t.device)["avg"]:
        dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group)
        return
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    if average:
        t.div_(ws)


class _CaptureColorGrad(torch.autograd.Function):
    """Identity on the colours an SH evaluation returned; in the backward the colour cotangent goes to the
    exchange (`GradientExchange.offer`) instead of through the SH backward.  The SH parameters are inputs only so
    that the node sits in the graph; they receive no gradient here."""

    @staticmethod
    def forward(ctx, colors, collector, clamped, *params):
        ctx.collector, ctx.clamped = collector, clamped
        if clamped:
            ctx.save_for_backward(colors)
        return colors.view_as(colors)

    @staticmethod
    def backward(ctx, v):
        if ctx.clamped:  # the fused epilogue marks the channels its clamp cut with -0.0: no gradient there
            (colors,) = ctx.saved_tensors
            cut = colors.view(torch.int32) == -2147483648
            v = torch.where(cut, torch.zeros((), dtype=v.dtype, device=v.device), v)
        ctx.collector.offer(v.contiguous())
        return (None, None, None) + (None,) * len(ctx.collector._sh["params"])


class GradientExchange:
    """The per-view data-parallel exchange, overlapped with the backward and cut to what
    is non-zero.

    * A post-accumulate hook on every parameter starts its all-reduce (``async_op``) the
      moment autograd has produced that gradient: the SH block -- 3/4 of the bytes at SH
      degree 3, and the first to finish (its node was created after the projection's, so
      it runs before it) -- is on the wire while ``project_backward`` still computes.  On
      RCCL the collectives run on the process group's own stream, ordered after the
      producing kernel by an event; ``finish()`` makes the current stream wait for them.
    * ``active_rows[name] = k``: only ``grad[:, :k]`` of that parameter can be non-zero
      (SH bands above the warm-up degree, vanilla_gs.py:811-820: the kernels write exact
      zeros there on every rank), so only that slice is exchanged -- 0 rows: nothing.
      45 of 59 floats per Gaussian are SH bands 1-3: during the first
      ``sh_degree_interval`` iterations the exchange is 56 B instead of 236 B per Gaussian.
    * averaging happens inside the collective on RCCL (``ReduceOp.AVG``), by one in-place
      division elsewhere.

    * ``sh_views`` (`deferred_sh_colors`): the SH gradient of ONE view is rank one per Gaussian,
      ``v_coeffs[g, k, :] = B_k(dir_g) v_colors[g, :]``.  Instead of all-reducing its 12 K bytes per Gaussian (180 of the
      236 B at degree 3) the ranks all-gather the 12-byte colour cotangents (+ their camera positions, one message,
      started from the backward of the SH node, i.e. early) and every rank forms
      ``1/W sum_r B(normalize(means - campos_r)) (x) v_colors_r`` itself with one kernel (`gsr_sh_backward_views`),
      views in rank order: the same bits on every rank.  Wire bytes per Gaussian and rank at W ranks:
      2 (W-1)/W x 236 -> 2 (W-1)/W x 44 + (W-1) x 12  (W = 8: 413 -> 161; W = 2: 236 -> 56).  Not bit-identical to the
      all-reduce of the per-rank gradients (other summation order, directions re-formed from the camera positions:
      1e-6 relative), and not available under HIP-graph replay (no hooks).

    ``bytes_last`` holds the bytes exchanged by the last ``finish()``.
    """

    def __init__(self, named_params, average: bool = True, group=None, flat_small: bool = True,
                 force: bool = False):
        """`force`: run the exchange on a process group of ONE rank too (every collective is then an identity that
        still goes through the backend: how the RCCL code path is exercised on a single-GPU box)."""
        self.named = dict(named_params)
        self.average, self.group = average, group
        # the small tensors (rows of <= 4 floats: means, scales, quats, opacities, features_dc = 56 of
        # the 236 B per Gaussian) travel as ONE flat message, started when the last of them exists:
        # five collectives less per step (their latency, not their bytes, is what they cost), at the
        # price of one packing copy -- the reduced values are handed back as views of the flat buffer
        self.flat_small = flat_small
        self._flat_arrived = {}
        self._flat_started = False
        self.active_rows = {}
        self.pending = []
        self.bytes_last = 0
        self._bytes = 0
        self._handles = []
        self.use_hooks = True
        self._sh = None           # this step's deferred SH gradient (deferred_sh_colors), or None
        self._sh_work = None
        self._held = []           # collectives that came up before this step's SH gather was issued (`_hold`)
        self.sh_views_backward = None  # (degree, deg_use, means, campos [W,3], v_colors [W,N,3], scale) -> grads; default: native
        self.enabled = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force)
        self._ws = dist.get_world_size(group) if self.enabled else 1
        dev = next(iter(self.named.values())).device if self.named else None
        self._caps = collective_capabilities(group, dev) if self.enabled else {}
        self._avg_in_collective = self.enabled and average and self._caps.get("avg", False)

    def attach(self) -> "GradientExchange":
        """(Re-)register the hooks, e.g. after refinement replaced the parameter objects.
        With ``use_hooks = False`` (HIP-graph replay: a hook would issue its collective during the
        warm-up backwards and bake one into the captured graph) nothing is registered and the
        caller starts the exchange with `start_all()` after the replay."""
        self.detach()
        if self.enabled and self.use_hooks:
            for name, p in self.named.items():
                self._handles.append(p.register_post_accumulate_grad_hook(
                    lambda param, name=name: self._start(name, param)))
        return self

    def detach(self) -> None:
        for h in self._handles:
            h.remove()
        self._handles = []

    def rebind(self, named_params) -> "GradientExchange":
        self.named = dict(named_params)
        return self.attach()

    def _is_small(self, name) -> bool:
        p = self.named[name]
        return self.flat_small and p.dim() >= 1 and p.shape[0] > 0 and p.numel() // p.shape[0] <= 4 \
            and self.active_rows.get(name) is None

    def _deferred(self):
        return self._sh["names"] if self._sh is not None else ()

    def _small_names(self):
        return [k for k in self.named if self._is_small(k) and k not in self._deferred()]

    # ---- the SH gradient as gathered colour cotangents (class docstring, `sh_views`) ----
    def deferred_sh_colors(self, sh_fn, names, params, means3d, campos, degree, degrees_to_use, clamped=False):
        """Colours of this rank's view, `sh_fn()` evaluated WITHOUT an autograd graph, wired so that their cotangent is
        gathered over the ranks and the gradients of the SH parameters `params` (named `names` in this exchange:
        ("features_dc", "features_rest") or ("sh_coeffs",)) are formed by `finish()`.  `clamped`: `sh_fn` applied the
        models' `clamp(rgbs + 0.5, min=0)` epilogue and marked the cut channels with -0.0.  Returns None when the
        exchange is off or runs without hooks: the caller then evaluates SH the ordinary way."""
        if self.begin_sh_views(names, params, means3d, campos, degree, degrees_to_use) is None:
            return None
        with torch.no_grad():
            colors = sh_fn()
        return _CaptureColorGrad.apply(colors, self, bool(clamped), *params)

    def begin_sh_views(self, names, params, means3d, campos, degree, degrees_to_use):
        """Announce that this step's SH gradient arrives as colour cotangents: the caller's backward must call
        `offer(v_colors)` once (v_colors [N,3], the clamp of the colours already applied) and leave the `.grad` of
        `params` alone; `finish()` sets them.  -> self (the collector), or None when the exchange is off or runs
        without hooks.  (`deferred_sh_colors` is this + the capture node; `gs_fused.render_gaussians(...,
        sh_collector=)` calls `offer` from its one native backward.)"""
        if not (self.enabled and self.use_hooks):
            return None
        self._sh = {"names": tuple(names), "params": tuple(params), "means": means3d.detach(),
                    "campos": campos.detach().reshape(3).to(means3d.dtype), "degree": int(degree),
                    "deg_use": int(degrees_to_use)}
        return self

    # The collectives of a step pair across ranks by ISSUE ORDER.  With a deferred SH gradient announced
    # (`begin_sh_views`) the all-gather of the colour cotangents is the FIRST collective of the step on every rank, by
    # construction rather than by the luck of autograd's ordering: an all-reduce that comes up while the gather has not
    # been issued yet is HELD (`_hold`) and issued right behind it -- from `offer()` (the normal case: the colour
    # cotangent leaves the compositing backward before any parameter gradient exists, nothing is ever held) or, on a
    # rank whose colours received no cotangent at all, from the top of `finish()`, where that rank joins the gather
    # with zeros.  (the zeros used to be offered at the very END of finish(), behind the late all-reduces --
    # an all-gather paired with an all-reduce on the other rank.)
    def _gather_outstanding(self) -> bool:
        return self._sh is not None and self._sh_work is None

    def _hold(self, thunk) -> bool:
        if self._gather_outstanding():
            self._held.append(thunk)
            return True
        return False

    def _release_held(self) -> None:
        held, self._held = self._held, []
        for thunk in held:
            thunk()

    def offer(self, v_colors: torch.Tensor) -> None:
        """(from the backward of `deferred_sh_colors`) start the all-gather of [v_colors | campos], then whatever
        all-reduces were held back for it."""
        sh = self._sh
        if self._sh_work is not None:
            raise RuntimeError("GradientExchange.offer: this step's colour cotangents were already gathered")
        msg = torch.cat((v_colors.reshape(-1), sh["campos"].to(v_colors.device)))
        out = torch.empty((self._ws, msg.numel()), dtype=msg.dtype, device=msg.device)
        if self._caps.get("gather_into_tensor", False):
            work = dist.all_gather_into_tensor(out.view(-1), msg, group=self.group, async_op=True)
        else:
            work = dist.all_gather(list(out.unbind(0)), msg, group=self.group, async_op=True)
        sh["out"] = out
        self._sh_work = work
        self._bytes += out.numel() * out.element_size()
        self._release_held()

    def _finish_sh(self) -> None:
        sh = self._sh
        self._sh = None
        if sh is None:
            return
        self._sh_work.wait()  # (issued by offer(), or with zeros at the top of finish())
        self._sh_work = None
        out, n = sh["out"], sh["means"].shape[0]
        v_all, campos_all = out[:, :3 * n], out[:, 3 * n:]
        scale = 1.0 / self._ws if self.average else 1.0
        split = len(sh["names"]) == 2
        fn = self.sh_views_backward
        if fn is None:
            from gs_fused import sh_backward_views as fn
        grads = fn(sh["degree"], sh["deg_use"], sh["means"], campos_all, v_all, scale, split)
        grads = grads if split else (grads,)
        for k, g in zip(sh["names"], grads):
            self.named[k].grad = g.view_as(self.named[k])

    def _start_flat(self) -> None:
        if self._hold(self._start_flat):
            self._flat_started = True  # (decided; issued right behind the gather)
            return
        names = self._small_names()
        parts = []
        for k in names:
            p = self.named[k]
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            parts.append(p.grad.reshape(-1))
        flat = torch.cat(parts)
        op = dist.ReduceOp.AVG if self._avg_in_collective else dist.ReduceOp.SUM
        work = dist.all_reduce(flat, op=op, group=self.group, async_op=True)
        self._bytes += flat.numel() * flat.element_size()
        self.pending.append((work, flat, names, "flat", op))
        self._flat_started = True

    def _start(self, name, p) -> None:
        if name in self._deferred():
            return
        if self._is_small(name):
            self._flat_arrived[name] = True
            if not self._flat_started and all(self._flat_arrived.get(k) for k in self._small_names()):
                self._start_flat()
            return
        if self._hold(lambda: self._start(name, p)):
            return
        g = p.grad
        rows = self.active_rows.get(name)
        stage = None
        if rows is not None and g.dim() >= 2 and rows < g.shape[1]:
            # the kernels write B[k] * v with B[k] = 0 into the inactive bands: exact zeros unless a
            # cotangent was non-finite on this rank only -- zero them so the replicas cannot diverge
            g[:, max(rows, 0):].zero_()
            if rows <= 0:
                return  # zeros on every rank
            stage = g[:, :rows].contiguous()  # the active bands, packed
            buf = stage
        else:
            buf = g
        op = dist.ReduceOp.AVG if self._avg_in_collective else dist.ReduceOp.SUM
        work = dist.all_reduce(buf, op=op, group=self.group, async_op=True)
        self._bytes += buf.numel() * buf.element_size()
        self.pending.append((work, buf, g, rows if stage is not None else None, op))

    def start_all(self) -> None:
        """Start the collectives for every parameter now (gradients written by a replayed HIP
        graph: no autograd hooks fire)."""
        if self.enabled:
            for name, p in self.named.items():
                if p.grad is not None:
                    self._start(name, p)

    def finish(self) -> int:
        """Wait for the collectives started during this backward; parameters that received
        no gradient on this rank are exchanged as zeros.  Returns the bytes exchanged.

        Collectives match across ranks by ISSUE ORDER: the hook-started ones follow the order in
        which autograd produces the gradients, the late ones (below) the order of `named`.  Every
        rank must therefore build the same autograd graph over the same set of parameters -- true
        for per-view data parallelism (same model, same ops, another camera); a rank-dependent
        branch that drops a parameter from the graph on one rank only would pair different
        tensors' collectives."""
        if not self.enabled:
            return 0
        if self._gather_outstanding():
            # the colours never received a cotangent on this rank (their output was unused in its loss): join the
            # gather with zeros -- still this step's first collective here, the held all-reduces follow it -- and
            # contribute no gradient
            n_ = self._sh["means"].shape[0]
            self.offer(torch.zeros((n_, 3), dtype=self._sh["means"].dtype, device=self._sh["means"].device))
        seen = {id(g) for _, _, g, rows, _ in self.pending if rows != "flat"}
        for name, p in self.named.items():
            if self._is_small(name) or name in self._deferred():
                continue
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            if id(p.grad) not in seen and (self.active_rows.get(name) is None or self.active_rows[name] > 0):
                self._start(name, p)
        if not self._flat_started and self._small_names():
            self._start_flat()  # some small parameter received no gradient on this rank: zeros
        for work, buf, g, rows, op in self.pending:
            work.wait()
            if self.average and op != dist.ReduceOp.AVG:
                buf.div_(self._ws)
            if rows == "flat":
                off = 0
                for k in g:
                    q = self.named[k]
                    piece = buf[off:off + q.numel()].view_as(q)
                    if self.use_hooks:
                        q.grad = piece  # a view of the flat buffer: no unpacking copy
                    else:
                        # gradients written by a replayed HIP graph live in STATIC tensors the graph keeps writing to:
                        # re-pointing `.grad` would leave the next replay's gradients where nobody reads them (and
                        # `start_all` would re-reduce this step's values) -- copy the reduced values back
                        q.grad.copy_(piece)
                    off += q.numel()
            elif rows is not None:
                g[:, :rows].copy_(buf)
        self.pending = []
        self._flat_arrived, self._flat_started = {}, False
        self._finish_sh()
        self.bytes_last, self._bytes = self._bytes, 0
        return self.bytes_last


def single_process_vis_counts(vis_counts: torch.Tensor, first_visible: Optional[torch.Tensor], rank: int) -> None:
    """`after_train` starts `vis_counts` at ONE for every Gaussian on its first call after a refinement,
    visible or not, and adds 1 per visible view afterwards (vanilla_gs.py:354-359).  Every rank applies
    that quirk to its own first view, so the plain sum over ranks is  world + (visible later views),
    while ONE process seeing the same views -- rank 0's first, then the others' -- would hold
    1 + (visible later views) + (visible first views of ranks >= 1).  In place, before the sum: a rank
    >= 1 replaces its "1" by whether its first view really saw the Gaussian."""
    if rank > 0 and first_visible is not None:
        vis_counts += first_visible.to(vis_counts.dtype) - 1


def allreduce_densify_stats(xys_grad_norm: torch.Tensor, vis_counts: torch.Tensor,
                            max_2dsize: torch.Tensor, group=None, first_visible: Optional[torch.Tensor] = None,
                            force: bool = False) -> None:
    """Keep the densification statistics (vanilla_gs.py:351-372) identical on
    every rank AND equal to what one process seeing all ranks' views would hold: sum, sum (with
    `single_process_vis_counts`), max.  In place."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return
    single_process_vis_counts(vis_counts, first_visible, dist.get_rank(group))
    packed = torch.stack([xys_grad_norm, vis_counts.to(xys_grad_norm.dtype)])
    dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    xys_grad_norm.copy_(packed[0])
    vis_counts.copy_(packed[1].to(vis_counts.dtype))
    dist.all_reduce(max_2dsize, op=dist.ReduceOp.MAX, group=group)


def view_for_rank(step: int, rank: int, world_size: int, num_views: int) -> int:
    """Deterministic per-rank view schedule: ranks never render the same view in
    the same step (the reference relies on per-rank RNG seeds, train.py:54)."""
    return (step * world_size + rank) % num_views


class ShardedAdam:
    """The other way to spend the same wire bytes: instead of all-reducing
    236 B per Gaussian and running Adam over ALL rows on every rank,

        reduce-scatter the gradients  ->  Adam on this rank's 1/n of the rows  ->  all-gather the parameters

    2 (n-1)/n x 236 B per Gaussian on the wire either way; the Adam kernel and both moments shrink to
    1/n (0.30 -> 0.04 ms and 472 -> 59 B per Gaussian at n = 8).  Rows, not a flat buffer: a parameter
    is [N, ...] row-major, so rows [r q, (r+1) q) of every tensor are one contiguous slice and refinement
    (which replaces all six tensors every `refine_every` iterations) needs no re-packing -- the shard
    parameters are re-made as views of the new tensors.  q = N // n; the N - n q tail rows (< n) are
    all-reduced and updated redundantly by every rank.

    Adam is element-wise, and for n = 2 `(a + b) / 2` is the same number whichever collective forms it:
    the parameters stay bit-identical to the all-reduce path (tests/test_dp_gloo.py, tests/test_gpu_dp.py).

    `make_optimizer(param_groups)` builds the inner optimizer (gs_fused.FusedAdam on the GPU,
    torch.optim.Adam elsewhere); its parameters are VIEWS of the model's rows."""

    def __init__(self, named_params, lrs, make_optimizer, group=None, average: bool = True, force: bool = False):
        """`force`: go through the collectives on a process group of one rank too (see GradientExchange)."""
        self.group, self.average = group, average
        self.enabled = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force)
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.rank = dist.get_rank(group) if self.enabled else 0
        self.lrs = dict(lrs)
        self.make_optimizer = make_optimizer
        dev = next(iter(dict(named_params).values())).device
        self._avg = self.enabled and average and collective_capabilities(group, dev)["avg"]
        self.bytes_last = 0
        self.step_count = 0
        self.bind(named_params, None)

    # ---- layout
    def _split(self, n):
        q = n // self.world
        return q, q * self.world

    def bind(self, named_params, full_moments) -> None:
        """(Re-)create the shard views over `named_params`; `full_moments` = {name: (exp_avg,
        exp_avg_sq)} full-size tensors to cut this rank's state from (after refinement), or None."""
        self.named = dict(named_params)
        n = next(iter(self.named.values())).shape[0]
        q, body = self._split(n)
        self.q, self.body, self.n = q, body, n
        lo, hi = self.rank * q, (self.rank + 1) * q
        self.main, self.tail, self.gshard = {}, {}, {}
        groups = []
        for name, p in self.named.items():
            d = p.detach()
            self.main[name] = torch.nn.Parameter(d[lo:hi])          # a view: updating it updates the model
            self.tail[name] = torch.nn.Parameter(d[body:]) if body < n else None
            self.gshard[name] = torch.empty_like(d[lo:hi])
            ps = [self.main[name]] + ([self.tail[name]] if self.tail[name] is not None else [])
            groups.append({"params": ps, "lr": self.lrs[name], "name": name})
        old_step = self.step_count
        self.inner = self.make_optimizer(groups)
        if full_moments is not None:
            for name in self.named:
                if full_moments.get(name) is None:
                    continue
                m, v = full_moments[name]
                for prm, sl in ((self.main[name], slice(lo, hi)), (self.tail[name], slice(body, n))):
                    if prm is None:
                        continue
                    self.inner.state[prm] = {"step": self._step_value(old_step), "exp_avg": m[sl].clone(),
                                             "exp_avg_sq": v[sl].clone()}

    def _step_value(self, k):
        # torch.optim.Adam keeps `step` as a tensor, gs_fused.FusedAdam as an int
        return torch.tensor(float(k)) if isinstance(self.inner, torch.optim.Adam) else int(k)

    def set_lr(self, name, lr) -> None:
        for g in self.inner.param_groups:
            if g["name"] == name:
                g["lr"] = lr

    # ---- one update
    def step(self) -> int:
        """gradients in `p.grad` of the full parameters -> updated, replica-identical parameters.
        Returns the bytes this rank handed to collectives."""
        lo, hi = self.rank * self.q, (self.rank + 1) * self.q
        nbytes = 0
        for name, p in self.named.items():
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            g = g.contiguous()
            if self.enabled:
                op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
                if self.q > 0:
                    dist.reduce_scatter_tensor(self.gshard[name], g[:self.body], op=op, group=self.group)
                    nbytes += g[:self.body].numel() * 4
                if self.body < self.n:
                    dist.all_reduce(g[self.body:], op=op, group=self.group)
                    nbytes += g[self.body:].numel() * 4
                if self.average and not self._avg:
                    self.gshard[name].div_(self.world)
                    g[self.body:].div_(self.world)
            else:
                self.gshard[name].copy_(g[lo:hi])
            self.main[name].grad = self.gshard[name]
            if self.tail[name] is not None:
                self.tail[name].grad = g[self.body:]
        self.inner.step()
        self.step_count += 1
        if self.enabled and self.q > 0:
            for name, p in self.named.items():
                d = p.detach()
                dist.all_gather_into_tensor(d[:self.body], d[lo:hi], group=self.group)  # in place
                nbytes += d[lo:hi].numel() * 4
        self.bytes_last = nbytes
        return nbytes

    # ---- refinement / checkpoints need the whole state
    def full_moments(self):
        """{name: (exp_avg, exp_avg_sq)} at full size on every rank (one all-gather per moment)."""
        out = {}
        lo, hi = self.rank * self.q, (self.rank + 1) * self.q
        for name, p in self.named.items():
            st = self.inner.state.get(self.main[name])
            if not st or "exp_avg" not in st:
                continue
            pair = []
            for key in ("exp_avg", "exp_avg_sq"):
                full = torch.zeros_like(p.detach())
                if self.q > 0:
                    if self.enabled:
                        dist.all_gather_into_tensor(full[:self.body], st[key].contiguous(), group=self.group)
                    else:
                        full[lo:hi].copy_(st[key])
                if self.tail[name] is not None:
                    full[self.body:].copy_(self.inner.state[self.tail[name]][key])
                pair.append(full)
            out[name] = tuple(pair)
        return out
