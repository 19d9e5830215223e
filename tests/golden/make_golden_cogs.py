#!/usr/bin/env python3
"""Generate tests/golden/cogs_losses.npz from the REFERENCE's own code for co-gs's optional loss terms.

Runs ONLY in the build container (it reads /root/reference).  `gs_toolkit.utils.losses` imports cv2 and open3d at its
top and `gs_toolkit.models.depth_gs` half of the toolkit, so neither can be imported here; as in
make_golden_refine.py the pieces are lifted out with `ast` AT GENERATION TIME and executed unmodified on torch CPU
tensors:
  * utils/losses.py: the functions `pearson_depth_loss`, `local_pearson_loss`, `tv_Loss` (`device="cuda"` in the
    source is served by a torch proxy that creates on the CPU; the patch corners `torch.randint` drew are recorded as
    inputs of the case -- torch's generator stream is not something another implementation can reproduce);
  * models/depth_gs.py, `DepthGSModel.get_loss_dict`: the statement blocks guarded by `use_scale_regularization`,
    `use_sparse_loss` and `"mono_depth_scale" in batch` (scale regularisation, sparse term, edge-aware scaled
    log-depth), run with a bare `self` / `batch` / locals carrying the case's tensors.
Nothing of the reference's source is stored: the committed .npz holds inputs and the values that code produced.

    python tests/golden/make_golden_cogs.py
"""
import ast
import math
import os
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF_LOSSES = "/root/reference/gs_toolkit/utils/losses.py"
REF_MODEL = "/root/reference/gs_toolkit/models/depth_gs.py"


class _TorchProxy(types.ModuleType):
    """torch, except that tensors asked for on "cuda" are made on the CPU and randint's draws are kept."""

    def __init__(self):
        super().__init__("torch")
        self.draws = []

    def __getattr__(self, k):
        return getattr(torch, k)

    def tensor(self, *a, **k):
        k.pop("device", None)
        return torch.tensor(*a, **k)

    def randint(self, *a, **k):
        k.pop("device", None)
        out = torch.randint(*a, **k)
        self.draws.append(out.clone())
        return out


def _functions(path, names, proxy):
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert {n.name for n in body} == set(names), set(names) - {n.name for n in body}
    mod = ast.Module(body=body, type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = {"torch": proxy, "math": math, "np": np}
    exec(compile(mod, path, "exec"), ns)
    return ns


def _guarded_blocks(path, func, needles):
    """{needle: [statements]} -- the bodies of the `if` statements inside `func` whose test mentions `needle`."""
    src = open(path).read()
    tree = ast.parse(src)
    fn = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == func)
    out = {}
    for node in ast.walk(fn):
        if isinstance(node, ast.If):
            test = ast.get_source_segment(src, node.test) or ""
            for nd in needles:
                if nd in test and nd not in out:
                    out[nd] = node.body
    assert set(out) == set(needles), set(needles) - set(out)
    return out


def _run(stmts, path, ns):
    mod = ast.Module(body=stmts, type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, path, "exec"), ns)
    return ns


def main():
    proxy = _TorchProxy()
    L = _functions(REF_LOSSES, ("pearson_depth_loss", "local_pearson_loss", "tv_Loss"), proxy)
    blocks = _guarded_blocks(REF_MODEL, "get_loss_dict", ("use_scale_regularization", "use_sparse_loss", '"mono_depth_scale" in batch'))
    out = {}
    rng = np.random.default_rng(20240929)
    cases = []
    for ci, (h, w, box, p_corr) in enumerate(((40, 56, 8, 0.5), (64, 48, 16, 0.5), (33, 47, 8, 1.0))):
        gt = rng.uniform(0.5, 4.0, (h, w)).astype(np.float32)
        pred = (0.6 * gt + 0.4 * rng.uniform(0.5, 4.0, (h, w))).astype(np.float32)
        img = rng.uniform(0.0, 1.0, (h, w, 3)).astype(np.float32)
        tp, tg, ti = torch.from_numpy(pred), torch.from_numpy(gt), torch.from_numpy(img)
        k = f"c{ci}_"
        out[k + "pred"], out[k + "gt"], out[k + "img"] = pred, gt, img
        out[k + "box_pcorr"] = np.array([box, p_corr], np.float64)
        out[k + "pearson"] = np.float64(L["pearson_depth_loss"](tp.reshape(-1), tg.reshape(-1)).item())
        torch.manual_seed(100 + ci)
        proxy.draws.clear()
        val = L["local_pearson_loss"](tp, tg, box, p_corr)
        assert len(proxy.draws) == 2
        out[k + "patch_rows"], out[k + "patch_cols"] = proxy.draws[0].numpy(), proxy.draws[1].numpy()
        out[k + "local_pearson"] = np.float64(val.item())
        out[k + "tv"] = np.float64(L["tv_Loss"](tp).item())
        # the scaled log-depth block (depth_gs.py: `if "mono_depth_scale" in batch:`)
        scale, shift = float(rng.uniform(0.7, 1.3)), float(rng.uniform(-0.2, 0.2))
        ns = {"torch": torch, "batch": {"mono_depth_scale": scale, "mono_depth_shift": shift}, "pred_depth": tp,
              "gt_depth": tg, "gt_img": ti, "loss_dict": {}}
        _run(blocks['"mono_depth_scale" in batch'], REF_MODEL, ns)
        out[k + "scale_shift"] = np.array([scale, shift], np.float64)
        out[k + "log_depth"] = np.float64(ns["loss_dict"]["log_depth"].item())
        cases.append(k)
    # scale regularisation / sparse term: a bare `self`
    for ci, n in enumerate((200, 1000)):
        log_scales = rng.normal(-3.0, 1.2, (n, 3)).astype(np.float32)
        opac = rng.uniform(0.02, 0.98, (n, 1)).astype(np.float32)  # (inside (0, 1): outside the source's logs are nan)
        cfg = types.SimpleNamespace(max_gauss_ratio=10.0, sparse_lambda=0.1)
        self_ = types.SimpleNamespace(config=cfg, scales=torch.from_numpy(log_scales),
                                      gauss_params={"opacities": torch.from_numpy(opac)})
        ns = {"torch": torch, "self": self_, "loss_dict": {}}
        _run(blocks["use_scale_regularization"], REF_MODEL, ns)
        _run(blocks["use_sparse_loss"], REF_MODEL, ns)
        k = f"s{ci}_"
        out[k + "log_scales"], out[k + "opacities"] = log_scales, opac
        out[k + "ratio_lambda"] = np.array([cfg.max_gauss_ratio, cfg.sparse_lambda], np.float64)
        out[k + "scale_reg"] = np.float64(ns["scale_reg"].item())
        out[k + "sparse_loss"] = np.float64(ns["loss_dict"]["sparse_loss"].item())
    np.savez_compressed(os.path.join(HERE, "cogs_losses.npz"), **out)
    print("wrote", os.path.join(HERE, "cogs_losses.npz"), len(out), "arrays;", cases)


if __name__ == "__main__":
    main()
