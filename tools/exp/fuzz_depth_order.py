"""Fuzz of gsr_depth_order (purpose-built sort with fused keys / count gather + decoupled look-back scan, and the
rocPRIM path outside 64 k .. 4 M): random sizes, 1-16 count rows, key distributions with many ties, culled
Gaussians; order and prefix against numpy.  python tools/exp/fuzz_depth_order.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")]
import numpy as np
import torch
import rasterizer.cuda as C


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for k in range(cases):
        n = int(rng.choice([1, 2, 4095, 4096, 4097, 65_536, 65_537, 100_003, 1_000_000, 2_500_001, 4_194_304, 4_194_305]))
        rows = int(rng.choice([1, 1, 2, 4, 7, 16]))
        if n * rows > 40_000_000:
            rows = 1
        dist = str(rng.choice(["random", "ties", "narrow", "sorted", "reversed"]))
        if dist == "random":
            depths = rng.uniform(0.01, 100.0, n)
        elif dist == "ties":
            depths = rng.integers(1, 40, n).astype(np.float64) * 0.25
        elif dist == "narrow":
            depths = 5.0 + rng.uniform(0, 1e-4, n)
        else:
            depths = np.sort(rng.uniform(0.01, 100.0, n))
            if dist == "reversed":
                depths = depths[::-1]
        depths = np.ascontiguousarray(depths.astype(np.float32))
        radii = (rng.random(n) < float(rng.choice([1.0, 0.7, 0.05]))).astype(np.int32) * rng.integers(1, 50, n).astype(np.int32)
        counts = rng.integers(0, int(rng.choice([2, 50, 2000])), (rows, n)).astype(np.int32)
        if counts.astype(np.int64).sum() >= 2 ** 31:
            counts //= 64
        order, cum = C.depth_order(torch.from_numpy(depths).cuda(), torch.from_numpy(radii).cuda(),
                                   torch.from_numpy(counts.reshape(-1)).cuda())
        key = np.where(radii > 0, depths.view(np.uint32), 0).astype(np.uint32)
        ref = np.argsort(key, kind="stable").astype(np.int32)
        ok = np.array_equal(order.cpu().numpy(), ref)
        want = np.cumsum(counts[:, ref].reshape(-1).astype(np.int64)).astype(np.int32)
        ok = ok and np.array_equal(cum.cpu().numpy(), want)
        order2, none = C.depth_order(torch.from_numpy(depths).cuda(), torch.from_numpy(radii).cuda(), None)
        ok = ok and none is None and np.array_equal(order2.cpu().numpy(), ref)
        print(f"case {k}: n={n} rows={rows} {dist} {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += 0 if ok else 1
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
