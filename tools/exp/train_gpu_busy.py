"""How busy is the GPU while config 3 trains?  Run under rocprofv3 --kernel-trace, then summarise the trace:
   rocprofv3 --kernel-trace --output-format csv -d <dir> -- python tools/exp/train_gpu_busy.py run
   python tools/exp/train_gpu_busy.py summarize <dir>
Iterations are delimited by the compositing forward launches (one per iteration)."""
import csv, glob, json, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gaussian-splatting-toolkit_amd")]


def run():
    import torch
    import bench
    from harness.train import train
    cfg = bench.config3(7000)
    cfg.phase_every = cfg.log_every = 0
    r = train(cfg, torch.device("cuda", 0))
    print(json.dumps({"iters_per_s": r["iters_per_s"], "seconds": r["seconds"], "N_end": r["num_gaussians_end"]}))


def summarize(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = []
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    fwd = [i for i, r in enumerate(rows) if "raster_fwd_tile16" in r[2]]
    # the training run = the LAST 7000 forward launches before the evaluation renders; take the longest run of
    # launches whose iteration time is below 20 ms
    print("kernels", len(rows), "compositing forward launches", len(fwd))
    its = fwd[-7000 - 64:]  # evaluation renders (<= 64) follow the loop
    best = None
    for off in range(0, len(its) - 7000 + 1):
        span = rows[its[off + 6999]][0] - rows[its[off]][0]
        if best is None or span < best[0]:
            best = (span, off)
    off = best[1]
    its = its[off:off + 7000]
    out = {}
    for name, a, b in (("480x270 (steps 0-499, before refinement)", 0, 500), ("480x270 (steps 0-1999)", 0, 2000), ("960x540 (2000-3999)", 2000, 4000),
                       ("1920x1080 (4000-6999)", 4000, 6999), ("all", 0, 6999)):
        i0, i1 = its[a], its[b]
        t0, t1 = rows[i0][0], rows[i1][0]
        busy = 0
        last_end = t0
        for s, e, _ in rows[i0:i1]:
            s = max(s, last_end)  # (overlapping streams: count the union)
            if e > s:
                busy += e - s
                last_end = e
        per = {}
        for s_, e_, k_ in rows[i0:i1]:
            k_ = k_.replace("(anonymous namespace)::", "").split("(")[0].split("<")[0].replace("void ", "").strip()[-48:]
            per[k_] = per.get(k_, 0) + (e_ - s_)
        top = sorted(per.items(), key=lambda kv: -kv[1])[:18]
        out[name + " us/iter by kernel"] = {k_: round(v_ / (b - a) / 1e3, 1) for k_, v_ in top}
        out[name] = {"ms_per_iter": round((t1 - t0) / (b - a) / 1e6, 4), "gpu_busy": round(busy / (t1 - t0), 3),
                     "kernels_per_iter": round((i1 - i0) / (b - a), 1)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else summarize(sys.argv[2])
