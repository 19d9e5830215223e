#!/bin/bash
# Lease 5: the -m gpu suite on the final kernels (saturation marks in the forward's runs); the forward against round 5's
# library and the regret cells on small grids again; the render mode against the process's own Python speed / CPU pinning.
out=$PWD/gpurun_out/lease5; mkdir -p $out
R=$PWD
( timeout 1700 python -m pytest tests -m gpu -q --durations=12 2>&1 | tail -40 ) > $out/gpu_suite.txt
tail -4 $out/gpu_suite.txt
ply=/tmp/config3_trained.ply; young=/tmp/config3_young.ply
python tools/exp/config3_rate.py 7000 $ply > $out/train_7000.json 2> $out/train.err || tail -5 $out/train.err
python tools/exp/config3_rate.py 1500 $young > $out/train_1500.json 2>> $out/train.err || tail -5 $out/train.err
cat $out/train_7000.json | cut -c1-600
run() {  # label, args...
  local label=$1; shift
  python bench.py "$@" --steps 100 --warmup 20 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'fwd', k['raster_fwd']['ms'], 'bwd', k['raster_bwd']['ms'], 'mean', d['config']['tile_list_length']['mean'])"
}
r05=$R/gaussian-splatting-toolkit_amd/rasterizer/cuda/libgsraster_r05.so
for scene in "ply:$young" "ply:$ply" "uniform" "ball" "longtail" "needles" "floaters" "room"; do
  GSR_LIBRARY=$r05 run "r05 $scene 300k 480x270" --scene $scene --gaussians 300000 --width 480 --height 270
  run "r06 $scene 300k 480x270" --scene $scene --gaussians 300000 --width 480 --height 270
done 2>&1 | tee $out/fwd_ab.txt
GSR_TUNE='{"depth_segments": 1}' run "r06-noseg needles 300k 480x270" --scene needles --gaussians 300000 --width 480 --height 270 | tee -a $out/fwd_ab.txt
timeout 300 python tools/exp/wave_trace.py --scene ply:$young --width 480 --height 270 > $out/wave_trace_young_480.txt 2>&1
REGRET_SMALL=1 timeout 900 python tools/r06/regret.py $out/regret_small.txt > $out/regret_stdout.txt 2>&1; head -12 $out/regret_small.txt
run2() { # tag, prefix..., -- args
  tag=$1; shift
  pfx=(); while [ "$1" != "--" ]; do pfx+=("$1"); shift; done; shift
  for i in $(seq 1 $REPS); do
    timeout 200 "${pfx[@]}" python tools/r06/mode480.py --tag $tag "$@" 2>>$out/err.log | grep '^{' >> $out/runs.jsonl
  done
}
REPS=8 run2 default env X=1 -- --syncs 1
REPS=6 run2 pinned8 taskset -c 8-15 -- --syncs 1
REPS=4 run2 pinned1 taskset -c 8 -- --syncs 1
python - <<PY | tee $out/modes.txt
import json
for l in open("$out/runs.jsonl"):
    r=json.loads(l)
    print(r["tag"], r["iters_per_s"], "render p50", r.get("render",{}).get("p50"), "bwd", r.get("backward",{}).get("p50"), "python loop us", r.get("python_200k_loop_us_before_after"), "round trip p50", [x["p50"] for x in r.get("round_trip_us_before_after",[])], "cpu", r.get("cpu_start_end"))
PY
