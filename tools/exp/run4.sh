cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_api.py -x -q -m gpu -k "interleaved or deterministic or cache" 2>&1 | tail -5 > gpurun_out/r03_tests3.log
for f in "" "--fused"; do for d in "" "--depth"; do python tools/render_bench.py $f $d 2>/dev/null | tail -1; done; done > gpurun_out/r03_render_bench.txt
python tools/render_bench.py --fused --depth --gaussians 3000000 --width 3840 --height 2160 --scale-lo 0.005 --scale-hi 0.05 2>/dev/null | tail -1 >> gpurun_out/r03_render_bench.txt
python tools/render_bench.py --depth --gaussians 3000000 --width 3840 --height 2160 --scale-lo 0.005 --scale-hi 0.05 2>/dev/null | tail -1 >> gpurun_out/r03_render_bench.txt
