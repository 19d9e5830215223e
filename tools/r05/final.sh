#!/bin/bash
# Round-5 record: the driver's command, the bench on the other workloads, rocprof stats, the whole GPU suite.
out=${1:-gpurun_out/final}; mkdir -p $out
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_form.json 2> $out/bench_driver_form.err
python bench.py > $out/bench_default.json 2> $out/bench_default.err
python bench.py --gaussians 3000000 --width 3840 --height 2160 --render-depth --fused-depth --train-iters 0 --no-cpu-baseline > $out/bench_config5_fused_depth.json 2>/dev/null
python bench.py --scene longtail --train-iters 0 --no-cpu-baseline > $out/bench_longtail.json 2>/dev/null
python bench.py --gaussians 200000 --scale-lo 0.005 --scale-hi 0.05 --train-iters 0 > $out/bench_config2.json 2>/dev/null
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_f; rocprofv3 --kernel-trace --stats -d /tmp/prof_f --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions --event-every 0 > /dev/null 2>&1
cp $(find /tmp/prof_f -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$out/kernel_stats.csv
cd $GRAFT_REPO_ROOT
python -c "
import json
for f in ('bench_driver_form', 'bench_default'):
    d = json.load(open('$out/' + f + '.json'))
    print(f, d['value'], d['value_normalised'], d['ms_per_step'], d['config']['timing'], d['config']['calibration']['valu_Tops'], d['config']['calibration']['copy_GBps'])
    print(' roofline', {k: d['roofline'].get(k) for k in ('kernel', 'frac', 'kernel_ms', 'valu_busy', 'valu_pipe_busy', 'traffic', 'algorithmic_bytes')}, d['roofline'].get('valu'))
    print(' e2e', d['end_to_end_algorithmic_GBps'], d['end_to_end_built_GBps'], 'syncs', d['ms_per_step_with_caller_syncs'], d['ms_per_step_with_caller_and_camera_syncs'])
    print(' digest', d['config'].get('train_digest'))
    print(' parity', d.get('parity_vs_oracle'))
"
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8) | tee $out/tests.log
