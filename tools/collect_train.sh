#!/bin/bash
# The trainer lines of DESIGN.md section 5 in one GPU lease: tools/collect_train.sh <tag> -> gpurun_out/train_<tag>_*.json
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd "$ROOT"
python tools/train_bench.py --gaussians 100000 --iters 50 > /dev/null 2>&1  # page the image in
python tools/train_bench.py --gaussians 1000000 --init-gaussians 300000 --iters 7000 --densify --views 48 --sh-interval 1000 --scene shell --scene-scale 0.004 0.02 --grad-thresh 0.00002 --log-every 500 2>/dev/null | tail -1 > "$OUT/train_${TAG}_config3.json"
python tools/train_bench.py --gaussians 1000000 --iters 400 2>/dev/null | tail -1 > "$OUT/train_${TAG}_1m_fixed.json"
python tools/train_bench.py --gaussians 1000000 --iters 400 --fused-render 2>/dev/null | tail -1 > "$OUT/train_${TAG}_1m_fused.json"
for f in config3 1m_fixed 1m_fused; do
  python -c "
import json
d = json.loads(open('$OUT/train_${TAG}_$f.json').read().strip().splitlines()[-1])
print('$f', round(d['iters_per_s'], 1), 'it/s  psnr', round(d['psnr_start'], 2), '->', round(d['psnr_end'], 2), ' N', d['num_gaussians_start'], '->', d['num_gaussians_end'])
"
done
