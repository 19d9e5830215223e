#!/bin/bash
# Lease 7: the held-out / trained-model parity tests with their full output; the training record with and without the
# CPU binding; kernel-time regret of the depth-segment rows on split-all grids (wall time is host-bound there).
out=$PWD/gpurun_out/lease7; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_heldout.py -q -x 2>&1 | tail -60 > $out/heldout.txt; tail -5 $out/heldout.txt
timeout 900 python -m pytest tests/test_gpu_heldout.py -q 2>&1 | grep -E "^(FAILED|PASSED|E  |tests/)|passed|failed|AssertionError|assert " | head -80 > $out/heldout_all.txt
for bind in auto off auto off; do
  timeout 600 python bench.py --train-only --train-iters 7000 --no-cogs --cpu-bind $bind 2>$out/train_$bind.err | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('cpu-bind $bind:', d.get('cpu_bind'), '| config3', d['iters_per_s'], 'it/s; with read-backs', d['iters_per_s_with_caller_syncs'], '; render by res (read-backs)', {k: v['render'] for k, v in d['with_caller_syncs']['phase_ms_median_by_resolution'].items()}, 'slow share', d['with_caller_syncs'].get('render_slow_share_lowest_resolution'), '; unchanged caller', d.get('iters_per_s_unchanged_caller'), '; fixed 1M', d['fixed_1m']['iters_per_s'], '; one-op', d['one_op_path']['iters_per_s'])" | tee -a $out/cpu_bind_ab.txt
done
run() {  # label, env, args...
  local label=$1; shift
  python bench.py "$@" --steps 60 --warmup 10 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'fwd', k['raster_fwd']['ms'], 'bwd', k['raster_bwd']['ms'], 'sum', round(k['raster_fwd']['ms'] + k['raster_bwd']['ms'], 4), 'wall', d['ms_per_step'])"
}
for scene in room floaters needles; do
  for res in "480 270 300000" "400 300 150000"; do
    set -- $res
    for t in '{}' '{"depth_segments": 1}' '{"depth_segments": 8}' '{"depth_segments_fwd": 8}' '{"depth_segments_fwd": 4}'; do
      GSR_TUNE="$t" run "$scene $1x$2 $t" --scene $scene --gaussians $3 --width $1 --height $2
    done
  done
done 2>&1 | tee $out/regret_small_kernel_time.txt
# the wave trace's finding: the 480x270 launches end on ONE unsegmented walk (a 459-entry tile, below depth_segments_min = 512)
young=/tmp/config3_young.ply; ply=/tmp/config3_trained.ply
python tools/exp/config3_rate.py 1500 $young > $out/train_1500.json 2> $out/train.err
python tools/exp/config3_rate.py 7000 $ply > $out/train_7000.json 2>> $out/train.err
for scene in "ply:$young" "ply:$ply" uniform room; do
  for m in 512 256 128 64; do
    GSR_TUNE="{\"depth_segments_min\": $m}" run "$scene 480x270 depth_segments_min=$m" --scene $scene --gaussians 300000 --width 480 --height 270
  done
done 2>&1 | tee $out/segments_min.txt
for m in 512 128; do
  GSR_TUNE="{\"depth_segments_min\": $m}" python tools/exp/config3_rate.py 7000 2>/dev/null | tail -1 | cut -c1-400 | sed "s/^/config3 depth_segments_min=$m: /" | tee -a $out/segments_min.txt
  GSR_TUNE="{\"depth_segments_min\": $m}" python tools/exp/config3_rate.py 7000 2>/dev/null | tail -1 | cut -c1-400 | sed "s/^/config3 depth_segments_min=$m: /" | tee -a $out/segments_min.txt
done
GSR_TUNE='{"depth_segments_min": 128}' timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_api.py -q -k "segment or compositing or oracle" 2>&1 | tail -3 | tee -a $out/segments_min.txt
# are the short kernels of the 480x270 phase clock-bound?  (a run wave lives 17.8 us for 64 entries, the launch is paced at ~520 workgroups/us)
( rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk\|fclk" | head -8
  run "young 480x270 default power policy" --scene ply:$young --width 480 --height 270
  rocm-smi --setperflevel high 2>&1 | tail -2
  rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk\|fclk" | head -8
  run "young 480x270 perflevel high" --scene ply:$young --width 480 --height 270
  run "uniform 1M 1080p perflevel high" --scene uniform
  rocm-smi --setperflevel auto 2>&1 | tail -1 ) 2>&1 | tee $out/clocks.txt
tools/exp/dispatch_bench 2>&1 | tee $out/dispatch_bench.txt
