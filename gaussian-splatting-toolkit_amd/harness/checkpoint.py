"""Checkpoints in the toolkit's layout, with its resize-on-load.

What the reference does (gs_toolkit/engine/trainer.py:404-476, models/vanilla_gs.py:236-258):
``save_checkpoint`` writes ``step-{step:09d}.ckpt`` = ``{"step", "pipeline": state_dict,
"optimizers": {group: optimizer.state_dict()}, "schedulers", "scalers"}`` with
``torch.save``; ``_load_checkpoint`` picks the latest step of a directory (or a file),
and ``GaussianSplattingModel.load_state_dict`` first RESIZES the six parameters to the
checkpoint's number of Gaussians (densification changed N since initialisation), also
accepting the old un-prefixed parameter names, before the values are copied in.  The
optimizers then load their own state dicts (one Adam per parameter group).

Here the optimizer may be the toolkit's six ``torch.optim.Adam`` objects or ONE
``gs_fused.FusedAdam`` with six parameter groups; either way the file holds one
``torch.optim.Adam``-format state dict per group name, so checkpoints are
interchangeable between the two (and with the reference's files for the model part).
Pure torch: host logic, no native code.
"""
import os
import re
from typing import Dict, Optional

import torch

PARAM_NAMES = ("means", "scales", "quats", "features_dc", "features_rest", "opacities")
_PREFIX = "_model.gauss_params."  # pipeline.state_dict() key prefix of the reference


def checkpoint_path(directory: str, step: int) -> str:
    return os.path.join(directory, f"step-{step:09d}.ckpt")


def latest_checkpoint(directory: str) -> str:
    """trainer.py:409-416: the largest step among ``step-*.ckpt``."""
    steps = sorted(int(m.group(1)) for m in (re.match(r"step-(\d+)\.ckpt$", f) for f in os.listdir(directory)) if m)
    if not steps:
        raise FileNotFoundError(f"no step-*.ckpt in {directory}")
    return checkpoint_path(directory, steps[-1])


def _group_of(optims: Dict[str, torch.optim.Optimizer], param) -> Optional[tuple]:
    for o in optims.values():
        for g in o.param_groups:
            if any(p is param for p in g["params"]):
                return o, g
    return None


def optimizer_state_dicts(model, optims: Dict[str, torch.optim.Optimizer]) -> Dict[str, dict]:
    """One ``torch.optim.Adam.state_dict()``-shaped dict per parameter group name."""
    out = {}
    for name in PARAM_NAMES:
        p = model.gauss[name]
        found = _group_of(optims, p)
        if found is None:
            continue
        o, g = found
        st = o.state.get(p, {})
        state = {}
        if st:
            state[0] = {"step": torch.tensor(float(st["step"])) if not torch.is_tensor(st["step"]) else st["step"],
                        "exp_avg": st["exp_avg"], "exp_avg_sq": st["exp_avg_sq"]}
        group = {k: v for k, v in g.items() if k != "params"}
        group["params"] = [0]
        out[name] = {"state": state, "param_groups": [group]}
    return out


def sharded_state_dicts(sharded) -> Dict[str, dict]:
    """The same per-group dicts from a `parallel.ShardedAdam`, whose moments live in row shards across the ranks:
    gathered to full size first (COLLECTIVE: every rank must call this; all of them return the full state)."""
    moments = sharded.full_moments()
    groups = {g["name"]: g for g in sharded.inner.param_groups}
    out = {}
    for name in sharded.named:
        g = groups[name]
        group = {k: v for k, v in g.items() if k not in ("params", "name")}
        group["params"] = [0]
        state = {}
        if name in moments:
            state[0] = {"step": torch.tensor(float(sharded.step_count)), "exp_avg": moments[name][0],
                        "exp_avg_sq": moments[name][1]}
        out[name] = {"state": state, "param_groups": [group]}
    return out


def save_checkpoint(directory: str, step: int, model, optims: Dict[str, torch.optim.Optimizer],
                    save_only_latest: bool = True, sharded=None, write: bool = True) -> Optional[str]:
    """trainer.py:444-476.  With `sharded` (a `parallel.ShardedAdam`) EVERY rank calls this -- the moments are
    gathered by a collective -- and only the rank with `write=True` touches the disk (the reference: rank 0,
    trainer.py:446 `@check_main_thread`)."""
    opt_state = sharded_state_dicts(sharded) if sharded is not None else optimizer_state_dicts(model, optims)
    if not write:
        return None
    os.makedirs(directory, exist_ok=True)
    path = checkpoint_path(directory, step)
    torch.save({
        "step": step,
        "pipeline": {_PREFIX + k: model.gauss[k].detach() for k in PARAM_NAMES},
        "optimizers": opt_state,
        "schedulers": {},
        # exactly what the reference writes for this method: mixed_precision=False -> a DISABLED
        # GradScaler, whose state_dict() is {} and whose load_state_dict is a no-op (trainer.py:126,425,467)
        "scalers": {},
    }, path)
    if save_only_latest:
        for f in os.listdir(directory):
            if f != os.path.basename(path) and re.match(r"step-\d+\.ckpt$", f):
                os.unlink(os.path.join(directory, f))
    return path


def load_model_state(model, state: Dict[str, torch.Tensor]) -> int:
    """``GaussianSplattingModel.load_state_dict`` (vanilla_gs.py:236-258): resize every
    parameter to the checkpoint's number of Gaussians, then copy.  Accepts the keys of a
    pipeline state dict (``_model.gauss_params.*``), of a model state dict
    (``gauss_params.*``) and the old bare names.  Returns the new N."""
    def find(name):
        for key in (_PREFIX + name, "gauss_params." + name, name):
            if key in state:
                return state[key]
        raise KeyError(f"checkpoint has no parameter '{name}'")

    newp = find("means").shape[0]
    for name in PARAM_NAMES:
        old = model.gauss[name]
        src = find(name)
        if src.shape[1:] != old.shape[1:]:
            raise ValueError(f"{name}: checkpoint rows are {tuple(src.shape[1:])}, model rows {tuple(old.shape[1:])}")
        new = torch.zeros((newp,) + tuple(old.shape[1:]), device=old.device, dtype=old.dtype)
        new.copy_(src)
        model.gauss[name] = torch.nn.Parameter(new)
    return newp


def load_checkpoint(path: str, model, optims: Dict[str, torch.optim.Optimizer], trust_pickle: bool = False,
                    sharded=None) -> int:
    """trainer.py:404-443 for one file or a directory (latest step).  The model is resized,
    every optimizer is pointed at the new parameter objects and gets the saved Adam state
    (``step``, ``exp_avg``, ``exp_avg_sq``) and learning rate.  Returns the step to resume
    at (``loaded step + 1``).  With `sharded` (a `parallel.ShardedAdam`; every rank reads the same file) the
    full-size moments are cut into this rank's rows instead."""
    if os.path.isdir(path):
        path = latest_checkpoint(path)
    # the payload is tensors, dicts, numbers and strings: the safe loader reads it.  A third-party
    # .ckpt may carry arbitrary pickled objects -- those only on the caller's explicit say-so.
    loaded = torch.load(path, map_location="cpu", weights_only=not trust_pickle)
    old = {k: model.gauss[k] for k in PARAM_NAMES}
    load_model_state(model, loaded["pipeline"])
    if sharded is not None:
        moments, steps = {}, []
        for name in PARAM_NAMES:
            sd = loaded.get("optimizers", {}).get(name)
            if not sd:
                continue
            if "lr" in sd["param_groups"][0]:
                sharded.lrs[name] = sd["param_groups"][0]["lr"]
            st = sd["state"].get(0)
            if st:
                dev = model.gauss[name].device
                if st["exp_avg"].shape != model.gauss[name].shape:
                    raise ValueError(f"optimizer state of {name} has {st['exp_avg'].shape[0]} rows, the model "
                                     f"{model.gauss[name].shape[0]}")
                moments[name] = (st["exp_avg"].to(device=dev, dtype=torch.float32).contiguous(),
                                 st["exp_avg_sq"].to(device=dev, dtype=torch.float32).contiguous())
                steps.append(int(float(st["step"])))
        sharded.step_count = max(steps) if steps else 0
        sharded.bind({k: model.gauss[k] for k in PARAM_NAMES}, moments)
        return int(loaded["step"]) + 1
    for name in PARAM_NAMES:
        found = _group_of(optims, old[name])
        if found is None:
            continue
        o, g = found
        new = model.gauss[name]
        o.state.pop(old[name], None)
        g["params"] = [new if p is old[name] else p for p in g["params"]]
        sd = loaded.get("optimizers", {}).get(name)
        if not sd:
            continue
        for k, v in sd["param_groups"][0].items():
            if k != "params" and k in g:
                g[k] = v
        st = sd["state"].get(0)
        if st:
            if st["exp_avg"].shape != new.shape:
                raise ValueError(f"optimizer state of {name} has {st['exp_avg'].shape[0]} rows, the model {new.shape[0]}")
            step = st["step"]
            keep_tensor_step = isinstance(o, torch.optim.Adam)  # torch's Adam keeps `step` as a tensor
            # ... a float32 one, on the parameter's device when the group is fused or capturable
            step_dev = new.device if (g.get("fused") or g.get("capturable")) else "cpu"
            o.state[new] = {
                "step": (torch.tensor(float(step), dtype=torch.float32, device=step_dev) if keep_tensor_step
                         else int(float(step))),
                "exp_avg": st["exp_avg"].to(device=new.device, dtype=torch.float32).contiguous(),
                "exp_avg_sq": st["exp_avg_sq"].to(device=new.device, dtype=torch.float32).contiguous(),
            }
    return int(loaded["step"]) + 1
