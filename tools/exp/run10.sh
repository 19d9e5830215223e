cd $GRAFT_REPO_ROOT
python tools/exp/dbg_sh0.py 2>&1 | tail -3 > gpurun_out/dbg_sh0.txt
python -m pytest tests/test_gpu_render.py tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r03_tests5.log
