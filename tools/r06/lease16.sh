out=$PWD/gpurun_out/lease16; mkdir -p $out; R=$PWD
ply=/tmp/config3_trained.ply
python tools/exp/config3_rate.py 7000 $ply > $out/train_7000.json 2> $out/train.err
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_t
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -- python $R/bench.py --scene ply:$ply --width 480 --height 270 --steps 50 --warmup 10 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions --event-every 0 > $out/bench.log 2>&1
python $R/tools/summarize_prof.py /tmp/prof_t $out/kt.json | head -12
cd $R; timeout 200 python tools/exp/wave_trace.py --scene ply:$ply --width 480 --height 270 2>&1 | grep -v amdgpu | head -12
