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


_avg_supported = {"ok": True}


def _allreduce_inplace(t: torch.Tensor, average: bool, ws: int, group) -> None:
    """RCCL averages inside the collective (ReduceOp.AVG): no extra pass over the
    192-MB SH gradient to divide it.  Falls back to SUM + div_ once if the backend
    refuses AVG (every rank runs the same library, so every rank falls back)."""
    if average and _avg_supported["ok"]:
        try:
            dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group)
            return
        except (RuntimeError, ValueError):
            _avg_supported["ok"] = False
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    if average:
        t.div_(ws)


def allreduce_densify_stats(xys_grad_norm: torch.Tensor, vis_counts: torch.Tensor,
                            max_2dsize: torch.Tensor, group=None) -> None:
    """Keep the densification statistics (vanilla_gs.py:351-372) identical on
    every rank: sum, sum, max.  In place."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    packed = torch.stack([xys_grad_norm, vis_counts.to(xys_grad_norm.dtype)])
    dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    xys_grad_norm.copy_(packed[0])
    vis_counts.copy_(packed[1].to(vis_counts.dtype))
    dist.all_reduce(max_2dsize, op=dist.ReduceOp.MAX, group=group)


def view_for_rank(step: int, rank: int, world_size: int, num_views: int) -> int:
    """Deterministic per-rank view schedule: ranks never render the same view in
    the same step (the reference relies on per-rank RNG seeds, train.py:54)."""
    return (step * world_size + rank) % num_views
