#!/bin/bash
out=$PWD/gpurun_out/lease12; mkdir -p $out
R=$PWD
young=/tmp/config3_young.ply
python tools/exp/config3_rate.py 1500 $young > $out/train_1500.json 2> $out/train.err
run() {  # label, args...
  local label=$1; shift
  python bench.py "$@" --steps 100 --warmup 20 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'ms', d['ms_per_step'], 'fwd', k['raster_fwd']['ms'], 'bwd', k['raster_bwd']['ms'])"
}
D=$R/gaussian-splatting-toolkit_amd/rasterizer/cuda
for scene in "ply:$young" uniform room needles; do
  for lib in prev q4 q8 q16; do
    GSR_LIBRARY=$D/libgsraster_$lib.so run "$lib $scene 480x270" --scene $scene --gaussians 300000 --width 480 --height 270
  done
done 2>&1 | tee $out/queue_batch_ab.txt
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_y
GSR_LIBRARY=$D/libgsraster_q8.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_y -- python $R/bench.py --scene ply:$young --width 480 --height 270 --steps 50 --warmup 10 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions --event-every 0 > $out/bench_young_traced.log 2>&1
python $R/tools/summarize_prof.py /tmp/prof_y $out/kernel_trace_young.json | head -10 | tee $out/kernel_trace_young_q8.txt
