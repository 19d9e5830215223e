#!/bin/bash
# Lease 6: lists-ahead launched from a helper thread (tuning row speculate_thread) -- parity tests of everything that
# touches it, then the A/B in fresh processes; the round's rocprofv3 evidence; the driver-form line on the final code.
out=$PWD/gpurun_out/lease6; mkdir -p $out
R=$PWD
( timeout 1200 python -m pytest tests/test_gpu_api.py tests/test_gpu_render.py tests/test_gpu_train.py tests/test_gpu_dp.py tests/test_gpu_nccl.py tests/test_gpu_cogs.py tests/test_gpu_kernels.py -q -x --durations=8 2>&1 | tail -25 ) > $out/gpu_tests.txt
tail -3 $out/gpu_tests.txt
run2() { # tag, env..., -- args
  tag=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 200 python tools/r06/mode480.py --tag $tag "$@" 2>>$out/err.log | grep '^{' >> $out/runs.jsonl
}
for i in 1 2 3 4 5 6; do
  run2 thread1_syncs X=1 -- --syncs 1
  run2 thread0_syncs GSR_TUNE='{"speculate_thread": 0}' -- --syncs 1
done
for i in 1 2 3; do
  run2 thread1_nosync X=1 -- --syncs 0
  run2 thread0_nosync GSR_TUNE='{"speculate_thread": 0}' -- --syncs 0
done
python - <<PY | tee $out/thread_ab.txt
import json, collections
by=collections.OrderedDict()
for l in open("$out/runs.jsonl"):
    r=json.loads(l); by.setdefault(r["tag"],[]).append(r)
for t,rs in by.items():
    print("%-16s it/s %-44s render p50 %s" % (t, " ".join("%.0f"%r["iters_per_s"] for r in rs), " ".join("%.3f"%r["render"]["p50"] for r in rs)))
PY
for v in 1 0; do
  GSR_TUNE="{\"speculate_thread\": $v}" python bench.py --steps 100 --warmup 20 --train-iters 0 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('bench default 1M 1080p, speculate_thread=$v: ms', d['ms_per_step'], 'with caller syncs', d['ms_per_step_with_caller_syncs'], 'with camera syncs too', d['ms_per_step_with_caller_and_camera_syncs'])" | tee -a $out/thread_ab.txt
done
bash tools/collect_profiles.sh r06 > $out/collect_profiles.log 2>&1; tail -2 $out/collect_profiles.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_form.json 2> $out/bench_driver_form.err; python - <<PY
import json
d=json.loads([l for l in open("$out/bench_driver_form.json") if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], "roofline", {k:v for k,v in d["roofline"].items() if not isinstance(v,(dict,list,str))})
print({k:v for k,v in d["config"].items() if not isinstance(v,(dict,list)) and k != "workload"})
PY
