"""GPU: the bench.py contract the driver depends on -- one JSON line with the agreed fields, at N = 1 and, self-
spawned from a plain `python bench.py --gpus 2`, at N = 2 (gloo here: both ranks share cuda:0)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--train-iters", "0", "--gaussians", "60000", "--width", "640", "--height", "360", "--steps", "4", "--warmup", "2", "--no-pmc"]


def _run(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + extra, capture_output=True,
                         text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields():
    d = _run([])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["unit"] == "Mpix/s" and d["value"] > 0 and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert abs(d["value"] - 640 * 360 / d["ms_per_step"] / 1e3) < 0.01 * d["value"]
    assert "workload" in d["config"] and "60000" in d["config"]["workload"] and "60000" in d["metric"].replace(" ", "").replace(",", "") or "60k" in d["metric"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # what lets a reader tell box from code: a calibration of the box next to the timed region,
    # the normalised value, the fixed-duration warm-up, the per-step spread -- inside `config`, which the driver keeps
    cal, tim = d["config"]["calibration"], d["config"]["timing"]
    assert 20 < cal["valu_Tops"] < 90 and 500 < cal["copy_GBps"] < 8000, cal
    assert abs(d["value_normalised"] - d["value"] * cal["reference_valu_Tops"] / cal["valu_Tops"]) < 0.01 * d["value"]
    assert tim["fixed_warmup_s"] >= 0.3 and tim["fixed_warmup_steps"] > 0 and tim["ms_per_step_median"] > 0
    assert d["end_to_end_built_GBps"] < 8000
    # the rank was bound to cores of its GPU's NUMA node (or says why not): round 6, bench.bind_cpus
    assert "cpu_bind" in d["config"] and (d["config"]["cpu_bind"] is None or isinstance(d["config"]["cpu_bind"], str))
    assert d["config"]["ms_per_step_median"] > 0 and d["config"]["calib_valu_Tops"] > 0 and d["config"]["parity_meets"] is True
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1


def test_plain_invocation_with_two_gpus_spawns_its_ranks():
    d = _run(["--gpus", "2", "--backend", "gloo", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    # whole-job value: both ranks' pixels over the slowest rank's time
    assert abs(d["value"] - 2 * 640 * 360 / d["ms_per_step"] / 1e3) < 0.01 * d["value"]
    assert d["allreduce_bytes"] and d["allreduce_bytes"] > 0


@pytest.mark.timeout(1500)
def test_eight_ranks_as_the_driver_will_launch_them():
    """`python bench.py --gpus 8 ...` end to end once: the self-spawn of eight ranks, the port
    selection, eight process groups, the timed region with every collective of the exchange, the other ranks leaving,
    and the training leg's SECOND spawn of eight ranks (config 3 / 4's code on a scene that trains in seconds; the
    co-gs leg rides in the two-rank test below -- eight ranks on one GPU took this test 11 of the suite's 17 minutes
    with it) -- all eight sharing cuda:0 through gloo, which is not a measurement but is the exact command path of
    the driver's N = 8 run."""
    # (the smallest workload that still exercises all of that -- VERDICT r5 item 6: 12 k Gaussians at 320 x 180, two timed
    #  steps, twelve training iterations; eight processes importing torch and opening the one GPU are what is left)
    base = ["--gaussians", "12000", "--width", "320", "--height", "180", "--steps", "2", "--warmup", "1", "--no-pmc"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + base +
                         ["--gpus", "8", "--backend", "gloo", "--train-small", "--train-iters", "12", "--no-cogs",
                          "--no-cpu-baseline", "--train-timeout", "1200"], capture_output=True, text=True, timeout=1400, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["value"] > 0
    # (whole-job value: eight ranks' pixels over the slowest rank's time; the line rounds it to 0.01 Mpix/s)
    assert abs(d["value"] - 8 * 320 * 180 / d["ms_per_step"] / 1e3) < 0.01 * d["value"] + 0.006
    assert d["allreduce_bytes"] > 0 and "dp8" in d["config"]["parallelism"]
    t = d["train"]
    assert t and "error" not in t, t
    assert t["n_gpus"] == 8 and t["iters"] == 12 and t["replicas_identical"] is True
    assert t["views_per_s"] > 7.9 * t["iters_per_s"] and t["allreduce_bytes_step_bytes"]


def test_the_drivers_torchrun_form():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    --gpus N ...`: the ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL + [
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0


def test_the_training_record_rides_in_the_same_line_at_one_and_two_ranks():
    """The second half of BASELINE's metric: `train` in the ONE JSON line, produced by a fresh process (its own
    process group) after the raster timing -- at N = 1, and at N = 2 self-spawned twice over (the bench's ranks,
    then the training leg's).  `--train-small` swaps config 3's scene for one that trains in seconds; the code
    path is the same."""
    base = [a for a in SMALL if a not in ("--train-iters", "0")]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    for extra in ([], ["--gpus", "2", "--backend", "gloo"]):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + base + extra +
                             ["--train-small", "--train-iters", "160", "--no-cpu-baseline"], capture_output=True,
                             text=True, timeout=900, env=env, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
        assert len(lines) == 1, out.stdout[-2000:]
        t = json.loads(lines[0])["train"]
        assert t and "error" not in t, t
        assert t["iters"] == 160 and t["iters_per_s"] > 0 and t["n_gpus"] == (2 if extra else 1)
        assert t["gaussians"]["start"] == 8000 and t["refinements"] >= 2
        assert t["psnr"]["end"] > t["psnr"]["start"] + 1.0, t   # (half of the 160 iterations run at reduced resolution)
        assert t["schedule"]["num_downscales"] == 2 and t["schedule"]["background_color"] == "random"
        assert len(t["phase_ms_median_by_resolution"]) == 3, t["phase_ms_median_by_resolution"]
        assert t["iters_per_s_with_caller_syncs"] > 0 and t["list_overflow_views"] == 0
        # BASELINE config 5's loop (co-gs) rides in the same record ...
        c = t["cogs_3m_4k"]
        assert "error" not in c and c["iters"] == 120 and c["iters_per_s"] > 0, c
        assert c["depth"]["loss_from_step"] == 41 and c["depth"]["one_compositing_pass"] is True
        assert set(c["phase_ms_median_by_depth_loss"]) == {"depth_loss_off", "depth_loss_on"}
        assert c["peak_memory_GB"] > 0 and c["list_overflow_views"] == 0
        if extra:
            assert t["replicas_identical"] is True and t["allreduce_bytes_step_bytes"], t
        else:
            # ... and at N = 1 the raster bench on the model the config-3 leg has just trained
            r = t["trained_raster"]
            assert "error" not in r and r["ms"] > 0 and r["gaussians"] == t["gaussians"]["end"], r
            assert r["raster_fwd_ms"] > 0 and r["raster_bwd_ms"] > 0 and r["tile_list_length"]["max"] > 0
            dg = json.loads(lines[0])["config"]["train_digest"]
            assert dg["config3_iters_per_s"] == t["iters_per_s"] and dg["cogs_3m_4k"]["iters_per_s"] == c["iters_per_s"]
            assert dg["trained_raster"]["ms"] == r["ms"]
            # what the DRIVER's record keeps: scalars at the top level of `config`
            cf = json.loads(lines[0])["config"]
            assert cf["config3_iters_per_s"] == t["iters_per_s"] and cf["cogs_3m_4k_iters_per_s"] == c["iters_per_s"]
            assert cf["config3_with_caller_syncs_iters_per_s"] == t["iters_per_s_with_caller_syncs"]
            assert cf["trained_raster_ms"] == r["ms"] and cf["calib_valu_Tops"] > 0 and cf["ms_per_step_median"] > 0
            assert any(k.startswith("render_") and k.endswith("_ms") for k in cf), sorted(cf)
            assert cf["render_480_slow_share"] is None or 0.0 <= cf["render_480_slow_share"] <= 1.0
            # ... and the trained model's own results against the CPU oracle, in the same line
            pv = r["parity_vs_oracle"]
            assert pv and "error" not in pv and pv["meets"]["image_1e-4_abs"] and pv["meets"]["gradients_1e-3_rel"], pv
            assert cf["trained_parity_meets"] is True and cf["trained_parity_image_max_abs"] < 1e-4
