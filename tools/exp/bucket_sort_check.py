"""Depth ordering, bucket pass + in-LDS pass (sort_bucket.hip) against a stable argsort, over distributions that
stress the bucket map (narrow ranges, clusters, heavy single buckets), and its time next to the four LSD passes
(GSR_DEPTH_SORT=radix in a second process)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gaussian-splatting-toolkit_amd"))
import rasterizer.cuda as C

dev = torch.device("cuda", 0)
def dists(n, rng):
    yield "uniform_2.5_7.5", rng.uniform(2.5, 7.5, n)
    yield "uniform_0.01_1000", rng.uniform(0.01, 1000.0, n)
    yield "equal", np.full(n, 3.25)
    yield "two", rng.choice([2.0, 7.5], n)
    yield "sorted", np.linspace(0.5, 50.0, n)
    yield "reversed", np.linspace(50.0, 0.5, n)
    yield "narrow", np.float32(4.0) + rng.integers(0, 200, n).astype(np.float32) * np.float32(4.7683716e-07)
    yield "normal", np.abs(rng.normal(5.0, 0.7, n)) + 0.2
    d = rng.uniform(1.0, 100.0, n); k = n // 2
    d[:k] = np.float32(1.0) + rng.integers(0, 400_000, k).astype(np.float32) * np.float32(1.1920929e-07)
    yield "heavy_bucket", d                       # half of the keys inside one bucket, distinct
    d = rng.uniform(1.0, 100.0, n); d[: n // 2] = 1.5
    yield "heavy_equal", d                        # half of the keys equal
    yield "lognormal", np.exp(rng.normal(1.0, 1.0, n))
    yield "tiny_range", np.float32(1.0) + rng.integers(0, 3, n).astype(np.float32) * np.float32(1.1920929e-07)
    yield "huge_range", np.exp(rng.uniform(-80, 80, n))

def check(n, seed=0, cull=0.1):
    rng = np.random.default_rng(seed)
    bad = 0
    for name, d in dists(n, rng):
        d = d.astype(np.float32)
        radii = np.ones(n, np.int32)
        if cull:
            radii[rng.integers(0, n, int(n * cull))] = 0
        key = np.where(radii > 0, d, 0).astype(np.float32)
        ref = np.argsort(key.view(np.uint32), kind="stable").astype(np.int32)
        dd, rr = torch.from_numpy(d).to(dev), torch.from_numpy(radii).to(dev)
        order, _ = C.depth_order(dd, rr, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            order, _ = C.depth_order(dd, rr, None)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        ok = np.array_equal(order.cpu().numpy(), ref)
        bad += not ok
        print(f"n={n:8d} {name:18s} {'ok ' if ok else 'BAD'} {ms*1e3:8.1f} us", flush=True)
    return bad

if __name__ == "__main__":
    print("GSR_DEPTH_SORT =", os.environ.get("GSR_DEPTH_SORT", "(bucket)"))
    bad = 0
    for n in [int(a) for a in sys.argv[1:]] or [65_537, 100_000, 1_000_000, 1_000_003, 1_600_000, 3_000_000, 4_194_304]:
        bad += check(n)
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)
