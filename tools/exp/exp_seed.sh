# sweep of the config-3 scene / seed parameters (round 3): which scene makes the reference's densify
# threshold (2e-4) grow a sparse seed to ~1 M Gaussians
cd $GRAFT_REPO_ROOT
python tools/train_bench.py --gaussians 100000 --iters 50 > /dev/null 2>&1
run() { tag=$1; shift; python tools/train_bench.py --iters 7000 --densify --views 48 --sh-interval 1000 --scene objects --means-lr-schedule --phase-every 50 --log-every 500 --init sfm --scene-scale 0.003 0.008 --tex-cell 0.03 "$@" 2>gpurun_out/exp6_$tag.err | tail -1 > gpurun_out/exp6_$tag.json; }
run e4_o400 --gaussians 6000000 --init-gaussians 300000 --objects 400 0.18 0.45 --extent 4.0
run e3_o200_c45 --gaussians 4000000 --init-gaussians 300000 --objects 200 0.18 0.45 --extent 3.0 --cam-radius 4.5
run e35_o300_c5 --gaussians 5000000 --init-gaussians 300000 --objects 300 0.18 0.45 --extent 3.5 --cam-radius 5
