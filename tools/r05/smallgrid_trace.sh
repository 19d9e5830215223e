#!/bin/bash
# Kernel trace of the raster step at 480x270 and 960x540 on the model config 3 ends with: which launches the small
# grids' render / backward time is made of.
out=$PWD/${1:-gpurun_out/smallgrid_trace}; mkdir -p $out
R=$PWD
ply=/tmp/config3_trained.ply
python bench.py --train-only --train-iters 7000 --no-cogs --train-export-ply $ply > $out/train.json 2> $out/train.err || exit 1
cd /tmp && export TMPDIR=/tmp
for wh in "480 270" "960 540"; do
  set -- $wh
  rm -rf /tmp/prof_sg
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sg -- python $R/bench.py --scene ply:$ply --width $1 --height $2 --steps 50 --warmup 10 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions --event-every 0 > $out/bench_$1.log 2>&1
  python $R/tools/summarize_prof.py /tmp/prof_sg $out/kernel_trace_$1.json > $out/kernel_trace_$1.txt
  python $R/tools/step_seq.py /tmp/prof_sg $out/step_sequence_$1.txt
done
