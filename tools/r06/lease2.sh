#!/bin/bash
# Lease 2 of round 6: (1) parity of the one-walk forward depth segments; (2) the same forward against round 5's library
# on 510-tile grids; (3) the two modes of the 480x270 render phase: counters, CPU / NUMA placement, speculation always on.
out=$PWD/gpurun_out/lease2; mkdir -p $out
R=$PWD
echo "== parity" | tee $out/parity.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "segment or compositing_kernels_on_random or deep_tiles or rasterize_forward or job_order" 2>&1 | tail -15 | tee -a $out/parity.txt
timeout 600 python -m pytest tests/test_gpu_api.py tests/test_gpu_render.py -x -q 2>&1 | tail -5 | tee -a $out/parity.txt
echo "== forward A/B (GSR_LIBRARY)" | tee $out/fwd_ab.txt
ply=/tmp/config3_trained.ply; young=/tmp/config3_young.ply
python tools/exp/config3_rate.py 7000 $ply > $out/train_7000.json 2> $out/train.err || tail -5 $out/train.err
python tools/exp/config3_rate.py 1500 $young > $out/train_1500.json 2>> $out/train.err || tail -5 $out/train.err
run() {  # label, args...
  local label=$1; shift
  python bench.py "$@" --steps 100 --warmup 20 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'fwd', k['raster_fwd']['ms'], 'bwd', k['raster_bwd']['ms'], 'mean', d['config']['tile_list_length']['mean'], 'par', d.get('parity_vs_oracle'))"
}
r05=$R/gaussian-splatting-toolkit_amd/rasterizer/cuda/libgsraster_r05.so
for scene in "ply:$young" "ply:$ply" "uniform" "ball" "longtail"; do
  for n in 300000; do
    GSR_LIBRARY=$r05 run "r05 $scene n=$n 480x270" --scene $scene --gaussians $n --width 480 --height 270
    run "r06 $scene n=$n 480x270" --scene $scene --gaussians $n --width 480 --height 270
  done
done 2>&1 | tee -a $out/fwd_ab.txt
GSR_LIBRARY=$r05 GSR_DEPTH_SEGMENTS_FWD=8 run "r05-8runs uniform n=300000 480x270" --scene uniform --gaussians 300000 --width 480 --height 270 | tee -a $out/fwd_ab.txt
GSR_LIBRARY=$r05 OPAQUE=1 run "r05 dense uniform n=1000000 480x270" --scene uniform --gaussians 1000000 --width 480 --height 270 | tee -a $out/fwd_ab.txt
run "r06 dense uniform n=1000000 480x270" --scene uniform --gaussians 1000000 --width 480 --height 270 | tee -a $out/fwd_ab.txt
echo "== modes" | tee $out/modes.txt
run2() { # tag, env..., -- args
  tag=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  for i in $(seq 1 $REPS); do
    env "${envs[@]}" timeout 200 $PFX python tools/r06/mode480.py --tag $tag "$@" 2>>$out/err.log | grep '^{' >> $out/runs.jsonl
  done
}
nproc; lscpu | grep -i "numa\|socket\|model name" | tee -a $out/modes.txt
REPS=8 PFX="" run2 default_syncs X=1 -- --syncs 1
REPS=6 PFX="" run2 lists_syncs GSR_SPECULATE=lists -- --syncs 1
REPS=4 PFX="" run2 lists_nosync GSR_SPECULATE=lists -- --syncs 0
gnode=$(cat /sys/class/drm/card0/device/numa_node 2>/dev/null); echo "gpu numa node $gnode" | tee -a $out/modes.txt
for node in $(ls -d /sys/devices/system/node/node* | sed 's/.*node//'); do
  cpus=$(cat /sys/devices/system/node/node$node/cpulist)
  REPS=3 PFX="taskset -c $cpus" run2 node${node}_syncs X=1 -- --syncs 1
done
python - <<PY | tee -a $out/modes.txt
import json, collections
rows=[json.loads(l) for l in open("$out/runs.jsonl")]
for r in rows:
    print(r["tag"], r["iters_per_s"], "render", r.get("render",{}).get("p50"), "bwd", r.get("backward",{}).get("p50"), "cpu", r.get("cpu_start_end"), "aff", r.get("affinity"), "gpu_node", r.get("gpu_numa_node"), r.get("counters"))
PY
