"""CPU, world_size 2, gloo: the trainer's data-parallel logic end to end
(`harness.train.train(world=2)`: per-rank views, gradient all-reduce, Adam,
all-reduced densification statistics, refinement with counter-based split samples,
optimizer-state surgery) on CPU stand-ins of the native ops (tests/cpu_standins.py,
backed by the oracle).  Both replicas must end bit-identical, with the same N.
"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GRAD_THRESH = 0.001  # 64x48 images: the per-pixel gradients are far above those at 1080p


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(rank, world, port, q, sharded=False, views=4, sh_exchange="dense", extra=None):
    for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(2 if world <= 2 else 1)
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import cpu_standins as SI
    import harness.pipeline as HP
    import harness.train as HT
    from gs_fused import RefineConfig
    from oracle import oracle as O

    O.set_threads(2 if world <= 2 else 1)
    HP.project_gaussians, HP.spherical_harmonics, HP.rasterize_gaussians = (
        SI.project_gaussians, SI.spherical_harmonics, SI.rasterize_gaussians)
    HT._refine = lambda params, moments, stats, rcfg, step, ntd, max_dim, seed: SI.refine_gaussians(
        params, moments, stats, rcfg, step, ntd, max_dim, seed=seed)
    # compressed schedule: densify at steps 20 and 50 (step % 30 > num_views + 10), opacity reset at
    # 10 and 40, cull only from step 55 on
    rcfg = RefineConfig(warmup_length=9, refine_every=10, reset_alpha_every=3, stop_screen_size_at=30,
                        stop_split_at=55, densify_grad_thresh=GRAD_THRESH, cull_alpha_thresh=0.05)
    cfg = HT.TrainConfig(num_gaussians=600, init_gaussians=200, width=64, height=48, num_views=views, iters=62,
                         sh_degree=1, sh_degree_interval=10, eval_views=2, densify=True, refine=rcfg,
                         scene_scale=(0.03, 0.15), sharded_adam=sharded, sh_exchange=sh_exchange)
    for k, v in (extra or {}).items():
        setattr(cfg, k, v)
    res = HT.train(cfg, torch.device("cpu"), rank, world)
    q.put((rank, res["param_checksum"], res["num_gaussians_start"], res["num_gaussians_end"], res["refinements"],
           res["psnr_start"], res["psnr_end"], res["allreduce_bytes"], res["update"]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_training_with_refinement_keeps_replicas_identical():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=500) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = results
    assert a[2] == b[2] == 200
    assert a[4] == b[4] and len(a[4]) >= 2, (a[4], b[4])       # same refinement history, N changed
    assert a[3] == b[3] and a[3] != 200                          # same final N on both ranks
    assert a[1] == b[1], (a[1], b[1])                            # bit-identical parameters (checksum in double)
    import math

    assert math.isfinite(a[1]) and math.isfinite(a[6])


def _launch(world, sharded=False, views=4, sh_exchange="dense", extra=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, world, port, q, sharded, views, sh_exchange, extra))
             for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=800) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return results


@pytest.mark.timeout(900)
def test_sharded_update_gives_the_same_replicas_as_the_all_reduce():
    """reduce-scatter -> Adam on this rank's rows -> all-gather (parallel.ShardedAdam) against all-reduce +
    Adam over every row: at two ranks (a + b) / 2 is the same number whichever collective forms it and
    Adam is element-wise, so the parameters must be BIT-identical -- through refinements that change N
    (odd N included: the tail rows are all-reduced) and an opacity reset."""
    plain = _launch(2, sharded=False)
    shard = _launch(2, sharded=True)
    assert shard[0][8].startswith("reduce-scatter") and plain[0][8].startswith("all-reduce")
    assert shard[0][1] == shard[1][1]                      # replicas identical
    assert shard[0][4] == plain[0][4] and len(shard[0][4]) >= 2   # same refinement history
    assert shard[0][1] == plain[0][1], (shard[0][1], plain[0][1])  # same parameters, bit for bit


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("sharded", [False, True])
def test_eight_rank_training_stand_in(sharded):
    """BASELINE config 4's shape (8 ranks, per-view data parallel) on the gloo stand-in: eight replicas
    stay bit-identical through refinement, take the same N history, and report the bytes they exchange
    per step in every phase (SH warm-up, after each refinement)."""
    res = _launch(8, sharded=sharded, views=8)
    sums = {r[1] for r in res}
    assert len(sums) == 1, sums
    hist = res[0][4]
    assert all(r[4] == hist for r in res) and len(hist) >= 2
    assert all(r[3] == res[0][3] for r in res) and res[0][3] != 200
    by = res[0][7]
    assert by and all(b > 0 for _, b in by)
    if not sharded:
        # degree 0 (56 B per Gaussian) -> degree 1 (+36 B), then new N after every refinement
        assert by[0][1] == 200 * 56 and by[1][0] == 10 and by[1][1] == 200 * 92, by
    assert len(by) >= 3, by


@pytest.mark.timeout(900)
def test_gathered_colour_cotangents_give_the_all_reduced_sh_gradient():
    """`GradientExchange` `sh_views` (the trainer's default): the ranks all-gather their 12-byte colour cotangents and
    camera positions and each forms the SH gradient of all views itself, instead of all-reducing it.  Replicas stay
    BIT-identical (every rank adds the views in rank order), the refinement history is the all-reduce run's, the
    parameters agree with it to rounding, and fewer bytes travel."""
    dense = _launch(2, sh_exchange="dense")
    views = _launch(2, sh_exchange="views")
    assert views[0][8].startswith("all-reduce (geometry) + all-gathered") and dense[0][8] == "all-reduce + Adam"
    assert views[0][1] == views[1][1]                                   # replicas identical, bit for bit
    assert views[0][4] == dense[0][4] and len(views[0][4]) >= 2         # same refinement history
    assert abs(views[0][1] - dense[0][1]) <= 1e-4 * abs(dense[0][1]), (views[0][1], dense[0][1])
    assert abs(views[0][6] - dense[0][6]) < 0.05                        # same PSNR at the end
    # degree 0, N = 200: 44 B per Gaussian all-reduced (means, scales, quats, opacities) + 2 x (3 N + 3) floats gathered
    assert views[0][7][0][1] == 200 * 44 + 2 * (3 * 200 + 3) * 4, views[0][7][:2]
    assert dense[0][7][0][1] == 200 * 56


@pytest.mark.timeout(1200)
def test_eight_ranks_with_gathered_colour_cotangents_stay_identical():
    res = _launch(8, views=8, sh_exchange="views")
    assert len({r[1] for r in res}) == 1
    assert all(r[4] == res[0][4] for r in res) and len(res[0][4]) >= 2
    by = res[0][7]
    # the bytes do not grow with the SH degree (step 10: degree 1): they change with N only, at the first refinement
    assert by[0][1] == 200 * 44 + 8 * (3 * 200 + 3) * 4 and by[1][0] > 10, by[:3]


@pytest.mark.timeout(900)
def test_sharded_adam_checkpoint_and_resume(tmp_path):
    """Config 4 with `sharded_adam` can save and resume (round-3 advice): the moments live in row shards across the
    ranks, `save_checkpoint` gathers them (every rank calls, rank 0 writes a file in the ordinary layout),
    `load_checkpoint` cuts this rank's rows out again.  A run resumed from the step-30 file (right behind a
    refinement, where the statistics restart anyway) ends bit-identical to the run that was never interrupted."""
    d = str(tmp_path / "ckpt")
    full = _launch(2, sharded=True, extra={"iters": 55, "save_every": 30, "checkpoint_dir": d})
    assert os.path.exists(os.path.join(d, "step-000000030.ckpt"))
    ck = torch.load(os.path.join(d, "step-000000030.ckpt"), map_location="cpu", weights_only=True)
    n30 = ck["pipeline"]["_model.gauss_params.means"].shape[0]
    for name in ("means", "features_rest", "opacities"):
        st = ck["optimizers"][name]["state"][0]
        assert st["exp_avg"].shape[0] == n30 and st["exp_avg_sq"].shape[0] == n30  # full size, not one rank's rows
        assert float(st["step"]) == 31.0
    resumed = _launch(2, sharded=True, extra={"iters": 55, "resume_from": d})
    assert resumed[0][1] == resumed[1][1]
    assert resumed[0][1] == full[0][1], (resumed[0][1], full[0][1])
    assert resumed[0][3] == full[0][3]
    # ... and the same file resumes the all-reduce path (one Adam over all rows) to the same parameters
    plain = _launch(2, sharded=False, extra={"iters": 55, "resume_from": d})
    assert plain[0][1] == full[0][1]
