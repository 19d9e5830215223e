cd $GRAFT_REPO_ROOT
P="--no-cpu-baseline --train-iters 0"
DENSE="--scale-lo 0.005 --scale-hi 0.05"
python bench.py $P $DENSE --gaussians 3000000 --width 3840 --height 2160 --render-depth --steps 20 --warmup 8 > gpurun_out/bench_r03_config5.json 2>/dev/null
python bench.py $P $DENSE --gaussians 3000000 --width 3840 --height 2160 --render-depth --fused-depth --steps 20 --warmup 8 > gpurun_out/bench_r03_config5_fused_depth.json 2>/dev/null
GSR_TWO_ROUND=0 python bench.py $P $DENSE --gaussians 3000000 --width 3840 --height 2160 --render-depth --fused-depth --steps 20 --warmup 8 > gpurun_out/bench_r03_config5_fused_depth_single_walk.json 2>/dev/null
