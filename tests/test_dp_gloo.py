"""CPU, world_size 2, gloo: the per-view data-parallel gradient exchange
(harness/parallel.py) equals single-process accumulation over both views."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _plain(obj):
    """Tensors -> numpy arrays, recursively, for the result queue: a torch tensor crosses a multiprocessing queue as a
    shared-memory handle that the parent can only open while the worker still lives (a worker that had already left
    reset the connection once the host got faster than the workers' barrier); a numpy array travels by value."""
    if isinstance(obj, torch.Tensor):
        return ("__tensor__", obj.detach().cpu().numpy())
    if isinstance(obj, (list, tuple)):
        return type(obj)(_plain(o) for o in obj)
    if isinstance(obj, dict):
        return {k: _plain(v) for k, v in obj.items()}
    return obj


def _tensors(obj):
    if isinstance(obj, tuple) and len(obj) == 2 and isinstance(obj[0], str) and obj[0] == "__tensor__":
        return torch.from_numpy(obj[1])
    if isinstance(obj, (list, tuple)):
        return type(obj)(_tensors(o) for o in obj)
    if isinstance(obj, dict):
        return {k: _tensors(v) for k, v in obj.items()}
    return obj


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _view_grads(view):
    """Per-view 'gradients' from the CPU oracle (the checker stands in for the
    GPU kernels here: this test is about the exchange, not the rasterizer)."""
    for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from harness import scene as S
    from oracle import oracle as O

    cam = S.make_camera(48, 32, yaw=0.05 * view)
    sc = S.make_scene(300, S.make_camera(48, 32), sh_degree=0, seed=9, scale_lo=0.03, scale_hi=0.2)
    n = 300
    colors = np.random.default_rng(1).uniform(0, 1, (n, 3)).astype(np.float32)
    bg = np.zeros(3, np.float32)
    r = O.render_forward(sc["means3d"], sc["scales"], 1.0, sc["quats"], cam.viewmat[:3], cam.projmat, cam.fx,
                         cam.fy, cam.cx, cam.cy, 32, 48, 16, colors, sc["opacities"], bg)
    v_img, v_alpha = S.make_cotangents(cam, seed=100 + view)
    vxy, vconic, vcol, vop = O.rasterize_backward(32, 48, 16, r["gaussian_ids_sorted"], r["tile_bins"], r["xys"],
                                                  r["conics"], colors, sc["opacities"], bg, r["final_Ts"],
                                                  r["final_idx"], v_img, v_alpha)
    z = np.zeros(n, np.float32)
    _, _, vmean, vscale, vquat = O.project_gaussians_backward(
        n, sc["means3d"], sc["scales"], 1.0, sc["quats"], cam.viewmat[:3], cam.projmat, cam.fx, cam.fy, cam.cx,
        cam.cy, 32, 48, r["cov3d"], r["radii"], r["conics"], r["compensation"], vxy, z, vconic, z)
    grads = [vmean, vscale, vquat, vop, vcol]
    radii = r["radii"]
    return [torch.from_numpy(g.copy()) for g in grads], torch.from_numpy(radii.copy())


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-toolkit_amd"))
    from harness.parallel import allreduce_densify_stats, allreduce_gradients, view_for_rank

    view = view_for_rank(step=0, rank=rank, world_size=world, num_views=8)
    grads, radii = _view_grads(view)
    params = [torch.zeros_like(g).requires_grad_(True) for g in grads]
    for p, g in zip(params, grads):
        p.grad = g.clone()
    params[3].grad = None  # a parameter without gradient is reduced as zeros
    flat = allreduce_gradients(params, average=False)
    # densification statistics: sum / sum / max
    gn = torch.full((5,), float(rank + 1))
    vc = torch.full((5,), rank + 1, dtype=torch.int32)
    mx = torch.tensor([float(rank), 3.0 - rank])
    allreduce_densify_stats(gn, vc, mx)
    summed = [p.grad.clone() for p in params]
    # the in-place, tensor-by-tensor variant (the default on RCCL) gives the same sums
    for p, g in zip(params, grads):
        p.grad = g.clone()
    params[3].grad = None
    assert allreduce_gradients(params, average=True, flat=False) is None
    inplace_ok = all(torch.allclose(p.grad * world, s_, rtol=1e-6, atol=1e-7) for p, s_ in zip(params, summed))
    for p, s_ in zip(params, summed):
        p.grad = s_
    q.put(_plain((rank, view, [p.grad.clone() for p in params], flat.numel(), gn, vc, mx, inplace_ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_allreduce_equals_accumulation():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([_tensors(q.get(timeout=240)) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    views = [r[1] for r in results]
    assert views[0] != views[1]
    g0, _ = _view_grads(views[0])
    g1, _ = _view_grads(views[1])
    expect = [a + b for a, b in zip(g0, g1)]
    expect[3] = g1[3] * 0  # both ranks had grad=None for that parameter
    for rank_res in results:
        for got, exp in zip(rank_res[2], expect):
            assert torch.allclose(got, exp, rtol=1e-6, atol=1e-7)
        assert rank_res[3] == sum(e.numel() for e in expect)
        assert torch.equal(rank_res[4], torch.full((5,), 3.0))
        assert torch.equal(rank_res[5], torch.full((5,), 3, dtype=torch.int32))
        assert torch.equal(rank_res[6], torch.tensor([1.0, 3.0]))
        assert rank_res[7]
    # both ranks hold bit-identical reduced gradients (replicas stay in sync)
    for a, b in zip(results[0][2], results[1][2]):
        assert torch.equal(a, b)


def test_flatten_roundtrip_single_process():
    sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-toolkit_amd"))
    from harness.parallel import allreduce_gradients, flatten_grads, unflatten_to_grads

    ps = [torch.randn(7, 3, requires_grad=True), torch.randn(7, 16, 3, requires_grad=True)]
    for p in ps:
        p.grad = torch.randn_like(p)
    before = [p.grad.clone() for p in ps]
    flat = flatten_grads(ps)
    assert flat.numel() == 7 * 3 + 7 * 48
    unflatten_to_grads(flat * 2, ps)
    for p, b in zip(ps, before):
        assert torch.equal(p.grad, b * 2)
    allreduce_gradients(ps)  # no process group: identity
    for p, b in zip(ps, before):
        assert torch.equal(p.grad, b * 2)


def _exchange_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-toolkit_amd"))
    from harness.parallel import GradientExchange

    g = torch.Generator().manual_seed(5)
    n = 40
    named = {"means": torch.randn(n, 3, generator=g).requires_grad_(True),
             "features_rest": torch.randn(n, 15, 3, generator=g).requires_grad_(True),
             "unused": torch.randn(n, 2, generator=g).requires_grad_(True)}
    ex = GradientExchange(named, average=True).attach()
    out = {}
    for active in (0, 3, 15):  # SH warm-up: degree 0, 1, 3
        ex.active_rows["features_rest"] = active
        for p in named.values():
            p.grad = None
        w = torch.full((n, 15, 3), float(rank + 1))
        w[:, active:] = 0  # the SH kernels write exact zeros for the inactive bands
        loss = (named["means"] * (rank + 1)).sum() + (named["features_rest"] * w).sum()
        loss.backward()  # the hooks start the collectives here
        nbytes = ex.finish()
        out[active] = (nbytes, named["means"].grad.clone(), named["features_rest"].grad.clone(),
                       named["unused"].grad.clone())
    q.put(_plain((rank, out)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_hooked_exchange_skips_inactive_sh_bands():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([_tensors(q.get(timeout=240)) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = 40
    for active in (0, 3, 15):
        (b0, m0, f0, u0), (b1, m1, f1, u1) = results[0][1][active], results[1][1][active]
        assert b0 == b1 == 4 * (n * 3 + n * active * 3 + n * 2)  # the bytes shrink with the active bands
        assert torch.equal(m0, m1) and torch.equal(f0, f1)      # replicas agree bit for bit
        assert torch.allclose(m0, torch.full((n, 3), 1.5))       # mean of 1 and 2
        exp = torch.zeros(n, 15, 3)
        exp[:, :active] = 1.5
        assert torch.equal(f0, exp)
        assert torch.equal(u0, torch.zeros(n, 2))                 # no gradient anywhere: zeros, exchanged


def test_vis_counts_merge_equals_one_process_seeing_all_views():
    """`after_train`'s "starts at 1" quirk under data parallelism: the per-rank counts, adjusted by
    `single_process_vis_counts` and summed, equal the counts of ONE process that saw rank 0's views,
    then rank 1's, ... (vanilla_gs.py:354-359)."""
    sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-toolkit_amd"))
    from harness.parallel import single_process_vis_counts

    g = torch.Generator().manual_seed(3)
    n, world, views_per_rank = 50, 3, 4
    visible = torch.rand(world, views_per_rank, n, generator=g) > 0.4
    per_rank, firsts = [], []
    for r in range(world):
        c = torch.ones(n, dtype=torch.int32)               # first call: 1 for everybody
        for v in range(1, views_per_rank):
            c += visible[r, v].to(torch.int32)             # later calls: +1 where visible
        per_rank.append(c)
        firsts.append(visible[r, 0].to(torch.int32))
    single = torch.ones(n, dtype=torch.int32)
    for r in range(world):
        for v in range(views_per_rank):
            if r == 0 and v == 0:
                continue
            single += visible[r, v].to(torch.int32)
    total = torch.zeros(n, dtype=torch.int32)
    for r in range(world):
        c = per_rank[r].clone()
        single_process_vis_counts(c, firsts[r], r)
        total += c
    assert torch.equal(total, single)
    assert not torch.equal(sum(per_rank), single)  # the plain sum is not


def _views_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    for p in (ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import cpu_standins as SI
    import harness.pipeline as HP
    import harness.train as HT
    from harness.parallel import GradientExchange

    HP.spherical_harmonics = SI.spherical_harmonics
    n, K, deg_use = 257, 9, 1
    g = torch.Generator().manual_seed(3)
    means = torch.randn(n, 3, generator=g)
    dc = torch.randn(n, 3, generator=g).requires_grad_(True)
    rest = torch.randn(n, K - 1, 3, generator=g).requires_grad_(True)
    other = torch.randn(n, 3, generator=g).requires_grad_(True)       # a parameter that is all-reduced as before
    campos = torch.tensor([4.0 + rank, 0.5 * rank, -1.0])
    weight = torch.randn(n, 3, generator=torch.Generator().manual_seed(10 + rank))   # this rank's colour cotangent
    dirs = means - campos
    dirs = dirs / dirs.norm(dim=-1, keepdim=True)

    def loss_of(colors):
        return (torch.clamp(colors + 0.5, min=0.0) * weight).sum() + (other * (rank + 1)).sum()

    # (a) every gradient all-reduced
    ex = GradientExchange({"features_dc": dc, "features_rest": rest, "other": other}, average=True).attach()
    loss_of(SI.spherical_harmonics(deg_use, dirs, torch.cat((dc[:, None, :], rest), 1))).backward()
    ex.finish()
    dense = [t.grad.clone() for t in (dc, rest, other)]
    ex.detach()
    for t in (dc, rest, other):
        t.grad = None
    # (b) the SH gradient from the gathered colour cotangents
    ex = GradientExchange({"features_dc": dc, "features_rest": rest, "other": other}, average=True).attach()
    ex.sh_views_backward = HT._sh_views_backward_autograd()
    colors = ex.deferred_sh_colors(lambda: SI.spherical_harmonics(deg_use, dirs, torch.cat((dc[:, None, :], rest), 1)),
                                   ("features_dc", "features_rest"), (dc, rest), means, campos, 2, deg_use)
    loss_of(colors).backward()
    assert dc.grad is None and rest.grad is None      # nothing went through the SH backward
    nbytes = ex.finish()
    views = [t.grad.clone() for t in (dc, rest, other)]
    q.put(_plain((rank, dense, views, nbytes)))
    dist.barrier()
    dist.destroy_process_group()


def test_sh_gradient_from_gathered_colour_cotangents_equals_the_all_reduced_one():
    """`GradientExchange.deferred_sh_colors`: the SH parameters get no gradient from autograd; `finish()` forms
    1/W sum_r B(dir_r) (x) v_colors_r from the all-gathered cotangents (here through the autograd stand-in of
    gsr_sh_backward_views) -- equal to the all-reduced gradients, bands above the warm-up degree zero, identical on
    both ranks, and the third parameter still travels by all-reduce."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_views_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([_tensors(q.get(timeout=120)) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        for d, v in zip(r[1], r[2]):
            assert torch.allclose(d, v, rtol=1e-5, atol=1e-6)
        assert not r[2][1][:, 3:, :].any()             # degree 1 in use: bands 4.. are exact zeros
        assert r[3] == 257 * 3 * 4 + world * (3 * 257 + 3) * 4   # `other` all-reduced + the gathered message
    for a, b in zip(res[0][2], res[1][2]):
        assert torch.equal(a, b)                        # the same bits on both ranks


def _static_grad_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-toolkit_amd"))
    from harness.parallel import GradientExchange

    n = 16
    named = {"means": torch.zeros(n, 3, requires_grad=True), "opacities": torch.zeros(n, 1, requires_grad=True),
             "features_rest": torch.zeros(n, 15, 3, requires_grad=True)}
    # what a captured HIP graph leaves behind: STATIC .grad tensors it writes into on every replay
    static = {k: torch.zeros_like(p) for k, p in named.items()}
    for k, p in named.items():
        p.grad = static[k]
    ex = GradientExchange(named, average=True)
    ex.use_hooks = False
    ex.attach()
    seen = []
    for step in range(1, 4):
        for k in named:  # "replay": this rank's gradient of this step, written in place
            static[k].fill_(float(step * (rank + 1)))
        ex.start_all()
        ex.finish()
        seen.append({k: (float(named[k].grad.mean()), named[k].grad.data_ptr() == static[k].data_ptr())
                     for k in named})
    q.put((rank, seen))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_exchange_without_hooks_keeps_the_static_gradients_of_a_replayed_graph():
    """Round-3 advice (high): under HIP-graph replay the exchange is started by `start_all()`; the gradients live
    in static tensors the graph writes into.  The flat small-tensor message must hand its result back INTO those
    tensors: re-pointing `.grad` at the flat buffer left the next replay's gradients unread (steps 2 and 3 then
    re-reduced step 1's values: 1.5, 1.5, 1.5 instead of 1.5, 3.0, 4.5)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_static_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=240) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, seen in results:
        for step, rec in enumerate(seen, start=1):
            for name, (mean, same_storage) in rec.items():
                assert mean == 1.5 * step, (rank, step, name, mean)   # mean of step and 2 step
                assert same_storage, (rank, step, name)


def _unoffered_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-toolkit_amd"))
    from harness.parallel import GradientExchange

    n = 12
    named = {"means": torch.randn(n, 3).requires_grad_(True), "sh_coeffs": torch.zeros(n, 4, 3, requires_grad=True)}
    ex = GradientExchange(named, average=False).attach()
    got = {}
    ex.sh_views_backward = lambda degree, deg_use, means, campos_all, v_all, scale, split: (
        got.setdefault("v", v_all.clone()), torch.zeros(n, 4, 3))[1]
    colors = ex.deferred_sh_colors(lambda: torch.ones(n, 3), ("sh_coeffs",), (named["sh_coeffs"],), named["means"],
                                   torch.zeros(3), 1, 1)
    # rank 1's loss does not use the colours at all: no cotangent reaches them there
    loss = named["means"].sum() + (colors.sum() * (rank + 2) if rank == 0 else 0.0)
    loss.backward()
    ex.finish()
    q.put((rank, got["v"].reshape(world, n, 3)[:, 0, 0].tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_a_rank_without_colour_cotangent_joins_the_gather_with_zeros():
    """Round-3 advice: `_finish_sh` used to raise on the rank whose backward never reached the colours while the
    others were already inside the all-gather (a hang, not an error).  It now contributes zeros."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_unoffered_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=240) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[0][1] == results[1][1] == [2.0, 0.0]


def _ordering_worker(rank, world, port, q, colours_first):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-toolkit_amd"))
    from harness.parallel import GradientExchange

    n = 10
    named = {"means": torch.randn(n, 3).requires_grad_(True), "big": torch.zeros(n, 15, 3, requires_grad=True),
             "sh_coeffs": torch.zeros(n, 4, 3, requires_grad=True)}
    ex = GradientExchange(named, average=False).attach()
    issued = []
    real_gather, real_reduce = dist.all_gather_into_tensor, dist.all_reduce
    dist.all_gather_into_tensor = lambda *a, **k: (issued.append("gather"), real_gather(*a, **k))[1]
    dist.all_reduce = lambda t, *a, **k: (issued.append("reduce%d" % t.numel()), real_reduce(t, *a, **k))[1]
    got = {}
    ex.sh_views_backward = lambda degree, deg_use, means, campos_all, v_all, scale, split: (
        got.setdefault("v", v_all.clone()), torch.zeros(n, 4, 3))[1]
    mk = lambda: ex.deferred_sh_colors(lambda: torch.ones(n, 3), ("sh_coeffs",), (named["sh_coeffs"],), named["means"],
                                       torch.zeros(3), 1, 1)
    # autograd runs the node created LAST first: `colours_first` decides whether rank 0's colour cotangent is offered
    # before or after the other parameters' hooks fire
    if colours_first:
        rest = named["big"].sum() * (rank + 1) + (named["means"] * 2).sum()
        colors = mk()
    else:
        colors = mk()
        rest = named["big"].sum() * (rank + 1) + (named["means"] * 2).sum()
    loss = rest + (colors.sum() * 5 if rank == 0 else 0.0)  # rank 1's loss does not touch the colours
    loss.backward()
    ex.finish()
    dist.all_gather_into_tensor, dist.all_reduce = real_gather, real_reduce
    q.put((rank, issued, got["v"].reshape(world, n, 3)[:, 0, 0].tolist(), float(named["big"].grad[0, 0, 0]),
           float(named["means"].grad[0, 0])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("colours_first", [True, False])
def test_the_colour_gather_is_the_first_collective_of_the_step_on_every_rank(colours_first):
    """Round-4 advice: a rank whose colours got no cotangent issued its (zero) all-gather at the END of finish(), behind
    all-reduces its hooks had started during the backward, while the other rank had issued the gather first:
    collectives pair by issue order, so an all-gather met an all-reduce.  Now every all-reduce that comes up before the
    step's gather is held and issued behind it -- whichever way autograd orders the nodes."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ordering_worker, args=(r, 2, port, q, colours_first)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=240) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, order0, v0, big0, m0), (_, order1, v1, big1, m1) = results
    assert order0 == order1 and order0[0] == "gather" and len(order0) == 3, (order0, order1)
    assert v0 == v1 == [5.0, 0.0]          # rank 0's cotangent, zeros from rank 1
    assert big0 == big1 == 3.0 and m0 == m1 == 4.0  # (1 + 2) and (2 + 2): summed over the ranks


def _caps_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-toolkit_amd"))
    from harness import parallel as P

    caps = P.collective_capabilities(None, torch.device("cpu"))
    again = P.collective_capabilities(None, torch.device("cpu"))
    # the exchange built on these answers still averages correctly (SUM + one division where AVG is missing)
    p = torch.zeros(8, 3, requires_grad=True)
    ex = P.GradientExchange({"means": p}, average=True).attach()
    (p * float(rank + 1)).sum().backward()
    ex.finish()
    q.put((rank, caps, again is caps, ex._avg_in_collective, float(p.grad.mean())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_capabilities_are_probed_once_and_agree_across_ranks():
    """`collective_capabilities` replaces the try/except around asynchronous collectives: one synchronous probe on a
    16-byte message when the exchange is constructed.  gloo: no ReduceOp.AVG (so the exchange sums and divides); the
    answer is cached and the same on every rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_caps_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=240) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, c0, cached0, avg0, g0), (_, c1, cached1, avg1, g1) = results
    assert c0 == c1 and cached0 and cached1
    assert c0["avg"] is False and avg0 is False and avg1 is False
    assert set(c0) == {"avg", "gather_into_tensor", "reduce_scatter_tensor"}
    assert g0 == g1 == 1.5
