"""Checkpoint save / resume in the toolkit's layout with resize-on-load
(gs_toolkit/engine/trainer.py:404-476, models/vanilla_gs.py:236-258).  Host logic: CPU."""
import os

import numpy as np
import pytest
import torch


def _model(n, seed, K=4):
    from harness.train import GaussianParams, blob_scene

    return GaussianParams(blob_scene(n, seed=seed, sh_degree={1: 0, 4: 1, 9: 2, 16: 3}[K]), torch.device("cpu"))


def _optims(model, fused_layout):
    from harness.train import LRS

    if fused_layout:  # one optimizer, six groups (the layout gs_fused.FusedAdam is used with)
        return {"all": torch.optim.Adam([{"params": [model.gauss[k]], "lr": lr} for k, lr in LRS.items()], eps=1e-15)}
    return {k: torch.optim.Adam([model.gauss[k]], lr=lr, eps=1e-15) for k, lr in LRS.items()}


def _step(model, optims, seed):
    g = torch.Generator().manual_seed(seed)
    for p in model.param_list():
        p.grad = torch.randn(p.shape, generator=g)
    for o in optims.values():
        o.step()


@pytest.mark.parametrize("save_fused,load_fused", [(False, False), (True, False), (False, True), (True, True)])
def test_checkpoint_roundtrip_resizes_model_and_optimizer(tmp_path, save_fused, load_fused):
    from harness import checkpoint as CK

    a = _model(500, seed=1)
    oa = _optims(a, save_fused)
    _step(a, oa, 1)
    _step(a, oa, 2)
    path = CK.save_checkpoint(str(tmp_path), 1200, a, oa)
    assert os.path.basename(path) == "step-000001200.ckpt"
    blob = torch.load(path, map_location="cpu", weights_only=False)
    # the reference's layout: trainer.py:452-466 / pipeline state-dict keys of the model
    assert set(blob) == {"step", "pipeline", "optimizers", "schedulers", "scalers"}
    assert set(blob["optimizers"]) == set(CK.PARAM_NAMES)
    assert "_model.gauss_params.means" in blob["pipeline"]
    # a model initialised with ANOTHER number of Gaussians resumes from it (densification grew N)
    b = _model(120, seed=9)
    ob = _optims(b, load_fused)
    _step(b, ob, 7)
    start = CK.load_checkpoint(str(tmp_path), b, ob)  # directory: latest step
    assert start == 1201 and b.num_points == 500
    for k in CK.PARAM_NAMES:
        assert torch.equal(b.gauss[k], a.gauss[k]) and b.gauss[k].requires_grad
    # the optimizers now drive the new parameter objects with the saved moments
    _step(a, oa, 3)
    _step(b, ob, 3)
    for k in CK.PARAM_NAMES:
        assert torch.allclose(b.gauss[k], a.gauss[k], rtol=0, atol=0), k
    # only the latest checkpoint is kept (save_only_latest_checkpoint, trainer.py:470-475)
    CK.save_checkpoint(str(tmp_path), 1300, a, oa)
    assert sorted(os.listdir(tmp_path)) == ["step-000001300.ckpt"]
    assert CK.latest_checkpoint(str(tmp_path)).endswith("step-000001300.ckpt")


def test_load_accepts_the_reference_s_key_styles(tmp_path):
    from harness import checkpoint as CK

    a = _model(64, seed=2)
    for prefix in ("", "gauss_params.", "_model.gauss_params."):
        b = _model(10, seed=3)
        n = CK.load_model_state(b, {prefix + k: a.gauss[k].detach() for k in CK.PARAM_NAMES})
        assert n == 64 and all(torch.equal(b.gauss[k], a.gauss[k]) for k in CK.PARAM_NAMES)
    with pytest.raises(KeyError):
        CK.load_model_state(_model(10, seed=3), {"means": a.gauss["means"].detach()})
