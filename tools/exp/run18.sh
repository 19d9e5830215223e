cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt2
GSR_TWO_ROUND=auto rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -- python $R/bench.py --no-cpu-baseline --no-pmc --train-iters 0 --scale-lo 0.005 --scale-hi 0.05 --gaussians 3000000 --width 3840 --height 2160 --render-depth --fused-depth --steps 20 --warmup 6 --event-every 0 > /tmp/kt2.log 2>&1
python $R/tools/summarize_prof.py /tmp/kt2 /tmp/kt2.json > $R/gpurun_out/r03_two_round_trace.txt
python $R/tools/step_seq.py /tmp/kt2 $R/gpurun_out/r03_two_round_step.txt
