cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_render.py tests/test_gpu_train.py tests/test_gpu_bench.py "tests/test_gpu_fullsize.py::test_config2_200k_sh3_1080p_forward_backward_vs_oracle" tests/test_gpu_dp.py -x -q -m gpu -s 2>&1 | tail -40 > gpurun_out/r03_tests1.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench1.json 2> gpurun_out/r03_bench1.err
