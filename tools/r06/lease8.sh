#!/bin/bash
# Lease 8: the round's closing records -- the whole -m gpu suite and the driver-form line on the final code.
out=$PWD/gpurun_out/${LEASE:-lease8}; mkdir -p $out
( timeout 1700 python -m pytest tests -m gpu -q --durations=12 2>&1 | tail -32 ) > $out/gpu_suite.txt
tail -4 $out/gpu_suite.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_form.json 2> $out/bench_driver_form.err; python - <<PY
import json
d=json.loads([l for l in open("$out/bench_driver_form.json") if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], "roofline", {k:v for k,v in d["roofline"].items() if not isinstance(v,(dict,list,str))})
print({k:v for k,v in d["config"].items() if not isinstance(v,(dict,list)) and k != "workload"})
PY
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --train-iters 0 > $out/bench_driver_form_2.json 2>/dev/null; python -c "
import json
d=json.loads([l for l in open('$out/bench_driver_form_2.json') if l.startswith('{')][-1]); print('second raster line: value', d['value'], 'ms', d['ms_per_step'], 'median', d['ms_per_step_median'])"
