#!/bin/bash
# Lease 11: the forward's segment launches as block queues -- parity, then the A/B against the library of the closing records.
out=$PWD/gpurun_out/lease11; mkdir -p $out
R=$PWD
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_api.py tests/test_gpu_render.py tests/test_gpu_cogs.py tests/test_gpu_heldout.py -q -x 2>&1 | tail -6 | tee $out/parity.txt
young=/tmp/config3_young.ply; ply=/tmp/config3_trained.ply
python tools/exp/config3_rate.py 1500 $young > $out/train_1500.json 2> $out/train.err
python tools/exp/config3_rate.py 7000 $ply > $out/train_7000.json 2>> $out/train.err
run() {  # label, args...
  local label=$1; shift
  python bench.py "$@" --steps 100 --warmup 20 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'ms', d['ms_per_step'], 'fwd', k['raster_fwd']['ms'], 'bwd', k['raster_bwd']['ms'])"
}
prev=$R/gaussian-splatting-toolkit_amd/rasterizer/cuda/libgsraster_prev.so
for scene in "ply:$young" "ply:$ply" uniform ball room needles floaters; do
  GSR_LIBRARY=$prev run "grid  $scene 480x270" --scene $scene --gaussians 300000 --width 480 --height 270
  run "queue $scene 480x270" --scene $scene --gaussians 300000 --width 480 --height 270
done 2>&1 | tee $out/queue_ab.txt
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_y
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_y -- python $R/bench.py --scene ply:$young --width 480 --height 270 --steps 50 --warmup 10 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions --event-every 0 > $out/bench_young_traced.log 2>&1
python $R/tools/summarize_prof.py /tmp/prof_y $out/kernel_trace_young.json | head -12 | tee $out/kernel_trace_young.txt
cd $R
for i in 1 2; do python tools/exp/config3_rate.py 7000 2>/dev/null | tail -1 | cut -c1-330 | tee -a $out/config3.txt; done
for i in 1 2; do GSR_LIBRARY=$prev python tools/exp/config3_rate.py 7000 2>/dev/null | tail -1 | cut -c1-330 | sed 's/^/prev: /' | tee -a $out/config3.txt; done
