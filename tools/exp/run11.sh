cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_render.py tests/test_gpu_train.py tests/test_gpu_dp.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r03_tests6.log
python tools/exp/c1_bench.py > gpurun_out/r03_c1_bench.json 2>/dev/null
for d in "" "--depth"; do python tools/render_bench.py --fused $d 2>/dev/null | tail -1; done > gpurun_out/r03_render_bench_view.txt
python tools/train_bench.py --gaussians 1000000 --iters 300 --fused-render 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('1M fused-render it/s', round(d['iters_per_s'],1))" >> gpurun_out/r03_render_bench_view.txt
python tools/train_bench.py --gaussians 50000 --width 640 --height 360 --iters 600 --fused-render 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('50k 640x360 fused-render it/s', round(d['iters_per_s'],1))" >> gpurun_out/r03_render_bench_view.txt
python tools/train_bench.py --gaussians 50000 --width 640 --height 360 --iters 600 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('50k 640x360 separate ops it/s', round(d['iters_per_s'],1))" >> gpurun_out/r03_render_bench_view.txt
