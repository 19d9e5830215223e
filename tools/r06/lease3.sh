#!/bin/bash
# Lease 3: (1) parity of the parallel re-walk; (2) forward A/B vs round 5's library on 510-tile grids; (3) the two
# render modes under a kernel-only trace (stream timing per process) and under runtime wait / queue settings.
out=$PWD/gpurun_out/lease3; mkdir -p $out
R=$PWD
echo "== parity" | tee $out/parity.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "segment or compositing_kernels_on_random or deep_tiles or rasterize_forward or job_order" 2>&1 | tail -15 | tee -a $out/parity.txt
timeout 600 python -m pytest tests/test_gpu_api.py tests/test_gpu_render.py tests/test_gpu_cogs.py -q 2>&1 | tail -8 | tee -a $out/parity.txt
echo "== forward A/B (GSR_LIBRARY)" | tee $out/fwd_ab.txt
ply=/tmp/config3_trained.ply; young=/tmp/config3_young.ply
python tools/exp/config3_rate.py 7000 $ply > $out/train_7000.json 2> $out/train.err || tail -5 $out/train.err
python tools/exp/config3_rate.py 1500 $young > $out/train_1500.json 2>> $out/train.err || tail -5 $out/train.err
run() {  # label, args...
  local label=$1; shift
  python bench.py "$@" --steps 100 --warmup 20 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'], 'fwd', k['raster_fwd']['ms'], 'bwd', k['raster_bwd']['ms'], 'mean', d['config']['tile_list_length']['mean'])"
}
r05=$R/gaussian-splatting-toolkit_amd/rasterizer/cuda/libgsraster_r05.so
for scene in "ply:$young" "ply:$ply" "uniform" "ball" "longtail"; do
  GSR_LIBRARY=$r05 run "r05 $scene 300k 480x270" --scene $scene --gaussians 300000 --width 480 --height 270
  run "r06 $scene 300k 480x270" --scene $scene --gaussians 300000 --width 480 --height 270
done 2>&1 | tee -a $out/fwd_ab.txt
GSR_LIBRARY=$r05 run "r05 dense uniform 1M 480x270" --scene uniform --gaussians 1000000 --width 480 --height 270 | tee -a $out/fwd_ab.txt
run "r06 dense uniform 1M 480x270" --scene uniform --gaussians 1000000 --width 480 --height 270 | tee -a $out/fwd_ab.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_y; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_y -- python $R/bench.py --scene ply:$young --width 480 --height 270 --steps 50 --warmup 10 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions --event-every 0 > $out/bench_young_traced.log 2>&1
python $R/tools/summarize_prof.py /tmp/prof_y $out/kernel_trace_young.json > $out/kernel_trace_young.txt 2>&1
python $R/tools/step_seq.py /tmp/prof_y $out/step_sequence_young.txt
cd $R
echo "== modes under a kernel-only trace" | tee $out/modes.txt
for i in 1 2 3 4 5 6 7 8; do
  rm -rf /tmp/prof_k
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_k -- python $R/tools/r06/mode480.py --iters 500 --syncs 1 --tag ktrace$i 2>>$out/err.log | grep '^{' > $out/ktrace$i.json)
  python - <<PY | tee -a $out/modes.txt
import json
r=json.load(open("$out/ktrace$i.json")); print("ktrace$i it/s", r["iters_per_s"], "render p50", r["render"]["p50"], "bwd", r["backward"]["p50"])
PY
  python tools/r06/side_timing.py /tmp/prof_k ktrace$i 2>>$out/err.log | tee -a $out/modes.txt
done
echo "== runtime settings (no trace)" | tee -a $out/modes.txt
run2() { # tag, env..., -- args
  tag=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  for i in $(seq 1 $REPS); do
    env "${envs[@]}" timeout 200 python tools/r06/mode480.py --tag $tag "$@" 2>>$out/err.log | grep '^{' >> $out/runs.jsonl
  done
}
REPS=6 run2 default X=1 -- --syncs 1
REPS=6 run2 nointerrupt HSA_ENABLE_INTERRUPT=0 -- --syncs 1
REPS=6 run2 activewait ROC_ACTIVE_WAIT_TIMEOUT=2000 -- --syncs 1
REPS=4 run2 hwq8 GPU_MAX_HW_QUEUES=8 -- --syncs 1
REPS=4 run2 hwq2 GPU_MAX_HW_QUEUES=2 -- --syncs 1
REPS=4 run2 nodirect AMD_DIRECT_DISPATCH=0 -- --syncs 1
python - <<PY | tee -a $out/modes.txt
import json
for l in open("$out/runs.jsonl"):
    r=json.loads(l)
    print(r["tag"], r["iters_per_s"], "render", r.get("render",{}).get("p50"), "p10", r.get("render",{}).get("p10"), "bwd", r.get("backward",{}).get("p50"), "slow", r.get("render_slow_share"))
PY
