#!/usr/bin/env python3
"""BASELINE configs 3/4: iterations/s of a full training step (render, L1+SSIM,
backward, [gradient all-reduce], 6 Adam groups) on a synthetic scene.

    python tools/train_bench.py --gaussians 1000000 --width 1920 --height 1080 --iters 200
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/train_bench.py ...

Prints one JSON line on rank 0."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from harness.train import TrainConfig, train  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--views", type=int, default=16)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--densify", action="store_true", help="refinement_after every refine_every iterations (config 3/5)")
    ap.add_argument("--init-gaussians", type=int, default=None, help="the model starts from this many coarser Gaussians")
    ap.add_argument("--sh-interval", type=int, default=None, help="SH degree warm-up interval (reference: 1000)")
    ap.add_argument("--grad-thresh", type=float, default=None, help="densify_grad_thresh (reference: 0.0002)")
    ap.add_argument("--log-every", type=int, default=0)
    ap.add_argument("--fused-render", action="store_true", help="gs_fused.render_gaussians: one autograd node per view")
    ap.add_argument("--graph", action="store_true", help="render -> loss -> backward replayed as one HIP graph per view")
    ap.add_argument("--scene", default="ball", choices=["ball", "shell", "objects"])
    ap.add_argument("--tex-cell", type=float, default=0.04)
    ap.add_argument("--objects", type=float, nargs=3, default=[48, 0.18, 0.45], help="spheres: count, r_lo, r_hi")
    ap.add_argument("--cam-radius", type=float, default=6.0)
    ap.add_argument("--extent", type=float, default=1.5)
    ap.add_argument("--scene-scale", type=float, nargs=2, default=[0.01, 0.06])
    ap.add_argument("--init", default="perturbed", choices=["perturbed", "sfm", "random"],
                    help="model start: the perturbed truth, or the reference's own initialisations (harness.train.seed_model)")
    ap.add_argument("--means-lr-schedule", action="store_true", help="exponential decay of the means' learning rate")
    ap.add_argument("--phase-every", type=int, default=0)
    ap.add_argument("--torch-activations", action="store_true", help="A/B: torch ops for exp/normalise/sigmoid/viewdirs")
    ap.add_argument("--cat-sh", action="store_true", help="A/B: torch.cat + spherical_harmonics instead of the split op")
    ap.add_argument("--torch-fused-adam", action="store_true", help="A/B: torch's fused Adam instead of gs_fused.FusedAdam")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    idx = local % torch.cuda.device_count()
    torch.cuda.set_device(idx)
    dev = torch.device("cuda", idx)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    rcfg = None
    if args.densify:
        from gs_fused import RefineConfig

        rcfg = RefineConfig()
        if args.grad_thresh is not None:
            rcfg.densify_grad_thresh = args.grad_thresh
    cfg = TrainConfig(num_gaussians=args.gaussians, width=args.width, height=args.height,
                      num_views=args.views, iters=args.iters,
                      sh_degree_interval=args.sh_interval or max(1, args.iters // 4),
                      torch_fused_adam=args.torch_fused_adam, split_sh=not args.cat_sh,
                      fused_activations=not args.torch_activations, densify=args.densify,
                      init_gaussians=args.init_gaussians, refine=rcfg, log_every=args.log_every,
                      fused_render=args.fused_render, use_graph=args.graph,
                      scene=args.scene, scene_scale=tuple(args.scene_scale), init=args.init, tex_cell=args.tex_cell, scene_objects=tuple(args.objects), cam_radius=args.cam_radius, scene_extent=args.extent,
                      means_lr_schedule=args.means_lr_schedule, phase_every=args.phase_every)
    res = train(cfg, dev, rank, world)
    if world > 1:
        cs = torch.tensor([res["param_checksum"]], dtype=torch.float64, device=dev)
        lo, hi = cs.clone(), cs.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        res["replicas_identical"] = bool(lo.item() == hi.item())
    if rank == 0:
        losses = res.pop("losses", None)
        if losses:
            res["loss_first_last"] = [losses[0], losses[-1]]
        res.update(metric="train iters/s", n_gpus=world, gaussians=args.gaussians,
                   resolution=f"{args.width}x{args.height}")
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
