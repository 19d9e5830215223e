#!/usr/bin/env python3
"""Host-visible latency of the read-backs the unchanged models make (`if (radii).sum() == 0`,
`assert (num_tiles_hit > 0).any()`, `camera.cx.item()`) on an otherwise idle GPU, and after a 20-us kernel.
Run under different environments (HSA_ENABLE_SDMA=0, HIP_FORCE_DEV_KERNARG=1 ...) to see what a user can win
without touching the models."""
import os
import time

import torch

dev = torch.device("cuda:0")
radii = torch.randint(0, 30, (1_000_000,), dtype=torch.int32, device=dev)
tiles = torch.randint(0, 9, (1_000_000,), dtype=torch.int32, device=dev)
sc = torch.rand(6, device=dev)
a = torch.rand(1_000_000, 3, device=dev)


def timeit(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


res = {
    "env": {k: os.environ.get(k) for k in ("HSA_ENABLE_SDMA", "HIP_FORCE_DEV_KERNARG", "GPU_MAX_HW_QUEUES")},
    "radii_sum_eq0_us": timeit(lambda: bool(radii.sum() == 0)),
    "tiles_gt0_any_us": timeit(lambda: bool((tiles > 0).any())),
    "scalar_item_us": timeit(lambda: sc[2].item()),
    "eight_items_us": timeit(lambda: [sc[i % 6].item() for i in range(8)]),
    "sync_only_us": timeit(lambda: torch.cuda.current_stream().synchronize()),
    "launch_one_kernel_then_sync_us": timeit(lambda: (a.mul_(1.0), torch.cuda.current_stream().synchronize())),
    "launch_one_kernel_no_sync_us": timeit(lambda: a.mul_(1.0), n=2000),
}
import json

print(json.dumps(res))
