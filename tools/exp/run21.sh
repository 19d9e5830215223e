cd $GRAFT_REPO_ROOT
GSR_TWO_ROUND=0 python -m pytest tests/test_gpu_fullsize.py "tests/test_gpu_train.py::test_config3_as_bench_py_times_it" -q -m gpu -x > gpurun_out/r03_tests9a.log 2>&1
HIP_LAUNCH_BLOCKING=1 AMD_LOG_LEVEL=1 python -m pytest tests/test_gpu_fullsize.py "tests/test_gpu_train.py::test_config3_as_bench_py_times_it" -q -m gpu -x > gpurun_out/r03_tests9b.log 2>&1
