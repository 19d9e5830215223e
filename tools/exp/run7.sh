cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_dp.py tests/test_gpu_bench.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r03_tests4.log
