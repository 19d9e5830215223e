cd $GRAFT_REPO_ROOT
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench2.json 2> gpurun_out/r03_bench2.err ) 2> gpurun_out/r03_bench2.time
