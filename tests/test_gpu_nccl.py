"""The RCCL code path on ONE GPU: a world-size-1 `nccl` process group with the exchange forced on.

Every multi-rank test of this repository runs on gloo (CPU tests here, two ranks sharing one GPU in
tests/test_gpu_dp.py); the `nccl` branches of harness/parallel.py -- `ReduceOp.AVG` inside async all-reduces,
`all_gather_into_tensor` (async, and in place), `reduce_scatter_tensor`, the flat small-tensor message, the stream
ordering of `finish()` -- would otherwise run for the first time when the driver launches `bench.py --gpus 8`.
With one rank every collective is an identity that still goes through RCCL and its own stream: the results must
equal the exchange-off run (bit for bit where the arithmetic is the same, to 1e-5 where the SH gradient is
re-formed from gathered colour cotangents).  Reference mechanism replaced: DistributedDataParallel over NCCL
(gs_toolkit/pipelines/base_pipeline.py:202-207, scripts/train.py:97-103).
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import json, os, sys
ROOT = sys.argv[1]
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")]
import numpy as np, torch, torch.distributed as dist
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from harness import scene as S
from harness.parallel import GradientExchange, ShardedAdam, collective_capabilities, allreduce_densify_stats
from harness.pipeline import CameraTensors, render_view
from gs_fused import FusedAdam

out = {"caps": collective_capabilities(None, dev)}
cam = S.make_camera(640, 360, yaw=0.05)
n = 120_000
sc = S.make_scene(n, cam, sh_degree=3, seed=3, scale_lo=0.004, scale_hi=0.04)
ct = CameraTensors.from_numpy(cam, dev)
bg = torch.tensor(S.BACKGROUND, device=dev)
v_img, v_alpha = (torch.from_numpy(a).to(dev) for a in S.make_cotangents(cam))
names = ("means3d", "scales", "quats", "opacities", "sh_coeffs")


def grads(mode, deg_use=3):
    p = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in sc.items()}
    ex = None
    if mode != "off":
        ex = GradientExchange({k: p[k] for k in names}, average=True, force=True).attach()
        ex.active_rows["sh_coeffs"] = (deg_use + 1) ** 2
        assert ex.enabled and ex._avg_in_collective == out["caps"]["avg"]
    o = render_view(p["means3d"], p["scales"], p["quats"], p["opacities"], p["sh_coeffs"], ct, bg, deg_use,
                    clamp_rgb=False,
                    sh_exchange=(ex, ("sh_coeffs",), (p["sh_coeffs"],)) if mode == "views" else None)
    torch.autograd.backward([o["rgb"], o["alpha"]], [v_img, v_alpha[..., None]])
    nbytes = ex.finish() if ex is not None else 0
    torch.cuda.synchronize()
    if ex is not None:
        ex.detach()
    return [p[k].grad.clone() for k in names], nbytes


import rasterizer.rasterize as R
R.set_deterministic(True)   # atomics-free compositing backward: two runs of the same view are comparable bit for bit
ref, _ = grads("off")
dense, b_dense = grads("dense")
views, b_views = grads("views")
out["dense_bitwise"] = all(torch.equal(a, b) for a, b in zip(ref, dense))
out["dense_bytes"] = b_dense
out["views_geometry_bitwise"] = all(torch.equal(a, b) for a, b in zip(ref[:4], views[:4]))
out["views_sh_rel"] = float((ref[4] - views[4]).abs().max() / ref[4].abs().max())
out["views_bytes"] = b_views
warm, b_warm = grads("dense", deg_use=1)   # SH warm-up: only the active bands travel
ref1, _ = grads("off", deg_use=1)
out["warmup_bitwise"] = all(torch.equal(a, b) for a, b in zip(ref1, warm))
out["warmup_bytes"] = b_warm

# sharded Adam through reduce_scatter_tensor + in-place all_gather_into_tensor against FusedAdam
g = torch.Generator(device=dev).manual_seed(1)
shapes = {"means": (n, 3), "opacities": (n, 1), "features_rest": (n, 15, 3)}
lrs = {"means": 1.6e-4, "opacities": 0.05, "features_rest": 0.000125}
pa = {k: torch.randn(s, device=dev, generator=g).requires_grad_(True) for k, s in shapes.items()}
pb = {k: v.detach().clone().requires_grad_(True) for k, v in pa.items()}
plain = FusedAdam([{"params": [pa[k]], "lr": lrs[k]} for k in shapes], eps=1e-15)
shard = ShardedAdam(pb, lrs, lambda groups: FusedAdam(groups, eps=1e-15), force=True)
assert shard.enabled
for it in range(3):
    for k in shapes:
        gr = torch.randn(shapes[k], device=dev, generator=g)
        pa[k].grad, pb[k].grad = gr.clone(), gr.clone()
    plain.step()
    out["sharded_bytes"] = shard.step()
torch.cuda.synchronize()
out["sharded_bitwise"] = all(torch.equal(pa[k], pb[k]) for k in shapes)
m = shard.full_moments()
out["moments_bitwise"] = all(torch.equal(plain.state[pa[k]]["exp_avg"], m[k][0]) and
                             torch.equal(plain.state[pa[k]]["exp_avg_sq"], m[k][1]) for k in shapes)
# the densification statistics' all-reduce (sum, sum, max)
a, c, mx = torch.rand(n, device=dev), torch.randint(0, 5, (n,), device=dev, dtype=torch.int32), torch.rand(n, device=dev)
a0, c0, m0 = a.clone(), c.clone(), mx.clone()
allreduce_densify_stats(a, c, mx, force=True)
out["stats_identity"] = bool(torch.equal(a, a0) and torch.equal(c, c0) and torch.equal(mx, m0))

# the trainer end to end on the forced exchange: hooks + sh_views, then sharded Adam, against the plain run
from harness.train import TrainConfig, train
from gs_fused import RefineConfig
def run(**kw):
    rc = RefineConfig(warmup_length=20, refine_every=10, reset_alpha_every=4, stop_screen_size_at=100)
    cfg = TrainConfig(num_gaussians=80_000, init_gaussians=30_000, width=320, height=180, num_views=6, iters=45,
                      sh_degree=2, sh_degree_interval=15, densify=True, refine=rc, scene_scale=(0.01, 0.04), **kw)
    r = train(cfg, dev, 0, 1)
    return r["param_checksum"], r["num_gaussians_end"], r["update"], r["allreduce_bytes"]
base = run()
forced_dense = run(force_exchange=True, sh_exchange="dense")
forced_sharded = run(force_exchange=True, sharded_adam=True)
forced_views = run(force_exchange=True, sh_exchange="views")
R.set_deterministic(False)
out["train"] = {"base": base[:3], "dense": forced_dense[:3], "sharded": forced_sharded[:3], "views": forced_views[:3],
                "dense_bytes": forced_dense[3][:2], "sharded_bytes": forced_sharded[3][:2]}
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(900)
def test_every_collective_of_the_data_parallel_path_runs_on_rccl_with_one_rank(tmp_path):
    script = tmp_path / "nccl_worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=800, env=env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
    assert out.returncode == 0 and lines, out.stdout[-3000:] + out.stderr[-3000:]
    r = json.loads(lines[-1][7:])
    caps = r["caps"]
    assert caps["avg"] and caps["gather_into_tensor"] and caps["reduce_scatter_tensor"], caps  # RCCL has all three
    n, K = 120_000, 16
    assert r["dense_bitwise"] and r["dense_bytes"] == 4 * n * (3 + 3 + 4 + 1 + 3 * K)
    assert r["warmup_bitwise"] and r["warmup_bytes"] == 4 * n * (3 + 3 + 4 + 1 + 3 * 4)
    assert r["views_geometry_bitwise"] and r["views_sh_rel"] < 1e-5
    assert r["views_bytes"] == 4 * n * (3 + 3 + 4 + 1) + 4 * (3 * n + 3)   # geometry all-reduced, colour cotangents gathered
    assert r["sharded_bitwise"] and r["moments_bitwise"] and r["stats_identity"]
    assert r["sharded_bytes"] == 2 * 4 * n * (3 + 1 + 45)                    # reduce-scatter in, all-gather out
    t = r["train"]
    assert t["dense"][2].startswith("all-reduce") and t["sharded"][2].startswith("reduce-scatter")
    assert t["views"][2].startswith("all-reduce (geometry)")
    assert t["base"][1] == t["dense"][1] == t["sharded"][1]
    assert t["dense"][0] == t["base"][0], (t["dense"], t["base"])           # same parameters, bit for bit
    assert t["sharded"][0] == t["base"][0], (t["sharded"], t["base"])
    assert abs(t["views"][0] - t["base"][0]) <= 1e-4 * abs(t["base"][0])     # SH gradient re-formed: close, not equal
    assert t["dense_bytes"] and t["sharded_bytes"]


@pytest.mark.timeout(900)
def test_bench_with_one_gpu_on_the_rccl_backend_and_the_exchange_forced():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--backend", "nccl",
                          "--force-exchange", "--gaussians", "200000", "--steps", "5", "--warmup", "3",
                          "--no-cpu-baseline", "--no-pmc", "--train-iters", "0", "--no-synced-regions"],
                         capture_output=True, text=True, timeout=800, env=env, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and lines, out.stdout[-2000:] + out.stderr[-3000:]
    line = json.loads(lines[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    # geometry all-reduced as one flat message, colour cotangents all-gathered (+ the camera position)
    assert line["allreduce_bytes"] == 4 * 200_000 * 11 + 4 * (3 * 200_000 + 3)
    assert line["allreduce_ms"] is not None
