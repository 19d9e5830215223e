#!/bin/bash
# All bench lines of DESIGN.md section 5 in one GPU lease:  tools/collect_benches.sh <tag>
# -> gpurun_out/bench_<tag>_*.json (copy the ones to keep into profiles/).  The default line
# runs first and in full (PMC passes + CPU baseline); the others skip both.
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd "$ROOT"
run() {  # name, flags...
  local name=$1; shift
  timeout 600 python bench.py "$@" > "$OUT/bench_${TAG}_${name}.json" 2> "$OUT/bench_${TAG}_${name}.err"
  python - "$OUT/bench_${TAG}_${name}.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = {n: round(v if isinstance(v, (int, float)) else v.get("ms", 0), 3) for n, v in (d.get("kernels") or {}).items()}
    print(f"{sys.argv[2]:22s} {d['ms_per_step']:8.4f} ms  {d['value']:8.1f} {d['unit']}  {k}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
Q="--no-pmc --no-cpu-baseline --train-iters 0"
P="--no-cpu-baseline --train-iters 0"   # with the counter passes (roofline.traffic, per-stage traffic)
DENSE="--scale-lo 0.005 --scale-hi 0.05"
run default
run dense $Q $DENSE
run config1 $Q $DENSE --gaussians 10000 --width 256 --height 256 --sh-degree 0
run config2 $Q $DENSE --gaussians 200000
run config5 $P $DENSE --gaussians 3000000 --width 3840 --height 2160 --render-depth --steps 20 --warmup 5
run config5_fused_depth $P $DENSE --gaussians 3000000 --width 3840 --height 2160 --render-depth --fused-depth --steps 20 --warmup 5
run longtail $Q --scene longtail
run deterministic $Q --deterministic
# a TRAINED distribution: the config-3 model after its 7 000 iterations, exported as `gs-export gaussian-splat`
# writes it, benched forward + backward from a training view (counter passes on: per-kernel traffic)
timeout 900 python - /tmp/config3_trained.ply <<'PY'
import os, sys
sys.path[:0] = [os.environ.get("GRAFT_REPO_ROOT", "."), os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gaussian-splatting-toolkit_amd")]
import torch, bench
from harness.train import train
cfg = bench.config3(7000)
cfg.export_ply, cfg.phase_every, cfg.log_every = sys.argv[1], 0, 0
r = train(cfg, torch.device("cuda", 0))
print("exported", sys.argv[1], "N", r["num_gaussians_end"], "psnr", r["psnr_end"], "it/s", r["iters_per_s"])
PY
run trained $P --scene "ply:/tmp/config3_trained.ply" --ply-cam-radius 5.0 --ply-view 3
