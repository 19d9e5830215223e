"""Adam over all parameter tensors of a Gaussian model in ONE HIP launch.

The toolkit gives every parameter group its own `torch.optim.Adam`
(gs_toolkit/engine/optimizers.py:59-196; learning rates and `eps=1e-15` from
configs/method_configs.py:47-80) and steps them one after the other.  `FusedAdam`
is a `torch.optim.Optimizer` with the same constructor arguments, the same
`param_groups` (per-group `lr`, so the toolkit's schedulers keep working) and the
same state keys (`step`, `exp_avg`, `exp_avg_sq`; `state_dict()` is
interchangeable with `torch.optim.Adam`'s), whose `step()` is one call of
`gsr_adam_step` (include/gsraster.h) per 8 tensors.  fp32 CUDA parameters only;
no CPU fallback.
"""
import ctypes as C
from typing import Iterable

import torch

from rasterizer.cuda import _call, _check, _stream

MAX_TENSORS = 8


class _AdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p),
                ("exp_avg_sq", C.c_void_p), ("n", C.c_longlong), ("lr", C.c_float)]


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # batches of tensors that share (betas, eps, step, device)
        batches = {}
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("FusedAdam does not support sparse gradients")
                _check(p, "param", torch.float32)
                g = _check(p.grad.contiguous(), "grad", torch.float32)
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                # state restored from a checkpoint (torch.optim.Adam's state_dict is interchangeable)
                # or handed over by refinement may be non-fp32 / non-contiguous / on another
                # device: re-materialise it rather than hand a wrong pointer to the kernel
                for key_ in ("exp_avg", "exp_avg_sq"):
                    t = st[key_]
                    if t.shape != p.shape:
                        raise RuntimeError(f"FusedAdam: {key_} has shape {tuple(t.shape)}, parameter {tuple(p.shape)}")
                    if t.dtype != torch.float32 or t.device != p.device or not t.is_contiguous():
                        st[key_] = t = t.to(device=p.device, dtype=torch.float32).contiguous()
                    _check(t, key_, torch.float32)
                st["step"] = int(st["step"]) + 1
                key = (p.device, float(b1), float(b2), float(group["eps"]), st["step"])
                batches.setdefault(key, []).append((p, g, st["exp_avg"], st["exp_avg_sq"], float(group["lr"])))
        for (dev, b1, b2, eps, step), items in batches.items():
            with torch.cuda.device(dev):
                for i in range(0, len(items), MAX_TENSORS):
                    chunk = items[i:i + MAX_TENSORS]
                    arr = (_AdamTensor * len(chunk))()
                    for j, (p, g, m, v, lr) in enumerate(chunk):
                        arr[j] = _AdamTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr)
                    _call("gsr_adam_step", C.c_int(len(chunk)), arr, C.c_double(b1), C.c_double(b2),
                          C.c_double(eps), C.c_longlong(step), _stream(dev))
        return loss
