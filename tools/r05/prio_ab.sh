#!/bin/bash
# Issue priority by estimated work (-DGSR_WAVE_PRIO=1 build, loaded through GSR_LIBRARY) against the default build.
out=gpurun_out/prio; mkdir -p $out
R=${GRAFT_REPO_ROOT:-/root/repo}
PRIO=${PRIO_LIB:-$R/gaussian-splatting-toolkit_amd/rasterizer/cuda/libgsraster_prio.so}
ply=/tmp/config3_trained.ply
python tools/exp/config3_rate.py 7000 $ply > $out/train_default.json 2> $out/train.err || { tail -5 $out/train.err; exit 1; }
tail -1 $out/train_default.json | cut -c1-120
run() {  # label, args...
  local label=$1; shift
  python bench.py "$@" --steps 100 --warmup 20 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'fwd', k['raster_fwd']['ms'], 'bwd', k['raster_bwd']['ms'])"
}
ab() {
  local label=$1; shift
  run "$label  default" "$@"
  GSR_LIBRARY=$PRIO run "$label  prio" "$@"
}
{
ab "trained 1080p" --scene ply:$ply
ab "uniform 1080p"
ab "longtail 1080p" --scene longtail
ab "ball 1080p" --scene ball
ab "trained 960x540" --scene ply:$ply --width 960 --height 540
ab "trained 480x270" --scene ply:$ply --width 480 --height 270
ab "trained 4K" --scene ply:$ply --width 3840 --height 2160
ab "config5-3M 4K" --gaussians 3000000 --width 3840 --height 2160 --render-depth --fused-depth
ab "uniform-200k 1080p" --gaussians 200000 --scale-lo 0.005 --scale-hi 0.05
} | tee $out/prio_ab.txt
for v in "GSR_NOTHING=1" "GSR_LIBRARY=$PRIO" "GSR_NOTHING=2" "GSR_LIBRARY=$PRIO"; do
  echo "config3 $v: $(env $v python tools/exp/config3_rate.py 7000 2>/dev/null | tail -1 | cut -c1-330)"
done | tee $out/config3_prio.txt
GSR_LIBRARY=$PRIO timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "raster or job_order or alike or segment or compositing" 2>&1 | tail -3
