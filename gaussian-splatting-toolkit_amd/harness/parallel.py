"""Per-view data parallelism for Gaussian training.

Every rank holds a full replica of the Gaussian parameters and renders a
different camera; after the backward the parameter gradients are summed across
ranks and (optionally) averaged.  The reference gets this from
``DistributedDataParallel`` (gs_toolkit/pipelines/base_pipeline.py:202-207),
which breaks as soon as densification replaces the ``nn.Parameter`` objects, so
the exchange is explicit here: all gradients are packed into ONE flat fp32
buffer (59 floats = 236 B per Gaussian at SH degree 3) and reduced with a single
collective -- on a ROCm build ``backend="nccl"`` is RCCL over xGMI, where one
large message is what keeps all seven links of a GPU busy; many small bucketed
ring all-reduces would be bound by a single link each.
"""
from typing import Iterable, List, Sequence

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


def allreduce_gradients(params: Sequence[torch.Tensor], average: bool = True,
                        group=None) -> torch.Tensor:
    """Sum (or average) the gradients of `params` over all ranks with a single
    all-reduce of the flat buffer; writes the result back into ``p.grad``."""
    flat = flatten_grads(params)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat.div_(dist.get_world_size(group))
    unflatten_to_grads(flat, params)
    return flat


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
