cd $GRAFT_REPO_ROOT
( time timeout 1500 python bench.py --gpus 2 --backend gloo --steps 10 --warmup 3 --train-iters 2000 > gpurun_out/r03_bench_2ranks.json 2> gpurun_out/r03_bench_2ranks.err ) 2> gpurun_out/r03_bench_2ranks.time
