import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'gaussian-splatting-toolkit_amd'), os.path.join(ROOT, 'tests')]
import torch
import test_gpu_render as T
from gs_fused import DensifyStats, ViewSpec, l1_ssim_loss, render_gaussians
from gs_fused.render import ViewGraph
from harness import scene as S
DEV='cuda:0'
W, H, n, K, deg = 256, 144, 20_000, 16, 2
bg = torch.tensor(S.BACKGROUND, device=DEV)
cams = [T._camera(W, H, i) for i in range(4)]
spec = ViewSpec(H, W, cams[0].fx, cams[0].fy, cams[0].cx, cams[0].cy, deg)
g = torch.Generator(device=DEV).manual_seed(1)
gts = [torch.rand(H, W, 3, device=DEV, generator=g) for _ in cams]
for name, loss_fn, use_stats in (("sum+stats", lambda out, t: (out["rgb"] * t[0]).sum(), True),):
    p = T._model(n, K, seed=6)
    stats = DensifyStats(n, DEV, max(W, H)) if use_stats else None
    vg = ViewGraph(p, spec, 3_000_000, loss_fn, bg, [(H, W, 3)], stats=stats)
    vg.capture(cams[0].viewmat, cams[0].projmat, cams[0].campos, (gts[0],))
    q = T._model(n, K, seed=6)
    for i in range(1):
        cam, gt = cams[i], gts[i]
        loss, out = vg.replay(cam.viewmat, cam.projmat, cam.campos, (gt,))
        for t in q.values(): t.grad = None
        oe = render_gaussians(q["means"], q["scales"], q["quats"], q["opacities"], q["features_dc"], q["features_rest"], cam.viewmat, cam.projmat, cam.campos, bg, spec, 3_000_000)
        le = loss_fn(oe, (gt,)); le.backward(); torch.cuda.synchronize()
        print(name, i, float(loss), float(le), {k: (float(p[k].grad.abs().max()), float(q[k].grad.abs().max())) for k in p})
