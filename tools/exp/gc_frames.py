#!/usr/bin/env python3
"""How much work does Python's cyclic garbage collector get per forward-only frame?  python tools/exp/gc_frames.py <root>"""
import collections
import gc
import os
import sys
import time

ROOT = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")]
import torch

from harness import scene as S
from harness.pipeline import CameraTensors, render_view

dev = torch.device("cuda:0")
cams = [S.make_camera(1920, 1080, yaw=0.01 * k) for k in range(8)]
sc = S.make_scene(1_000_000, cams[0], sh_degree=3, seed=42, scale_lo=0.0025, scale_hi=0.025)
p = {k: torch.from_numpy(v).to(dev) for k, v in sc.items()}
camt = [CameraTensors.from_numpy(c, dev) for c in cams]
bg = torch.tensor(S.BACKGROUND, device=dev)


def frame(k):
    with torch.no_grad():
        return render_view(p["means3d"], p["scales"], p["quats"], p["opacities"], p["sh_coeffs"], camt[k % 8], bg, 3)


for k in range(20):
    out = frame(k)
torch.cuda.synchronize()
gc.collect()
s0 = [dict(d) for d in gc.get_stats()]
c0 = gc.get_count()
t0 = time.perf_counter()
for k in range(200):
    out = frame(k)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 200 * 1e3
s1 = gc.get_stats()
print(ROOT, "%.4f ms/frame" % dt, "collections per generation over 200 frames:",
      [b["collections"] - a["collections"] for a, b in zip(s0, s1)], "collected:", [b["collected"] - a["collected"] for a, b in zip(s0, s1)],
      "tracked objects:", len(gc.get_objects()))
# what is cyclic garbage made of?
gc.collect()
gc.set_debug(gc.DEBUG_SAVEALL)
for k in range(5):
    out = frame(k)
torch.cuda.synchronize()
gc.collect()
kinds = collections.Counter(type(o).__name__ for o in gc.garbage)
print("cyclic garbage of 5 frames:", kinds.most_common(12))
fn = [o for o in gc.garbage if type(o).__name__ == "function"]
print("functions in cycles:", sorted({getattr(f, "__qualname__", "?") for f in fn})[:20])
gc.set_debug(0)
