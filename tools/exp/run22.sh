cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_api.py tests/test_gpu_fullsize.py -q -m gpu -k "two_round or config5" 2>&1 | tail -3 > gpurun_out/r03_tests8.log
bash tools/exp/run20.sh
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench3.json 2> gpurun_out/r03_bench3.err
