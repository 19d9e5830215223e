out=$PWD/gpurun_out/lease17; mkdir -p $out; R=$PWD
ply=/tmp/config3_trained.ply; young=/tmp/config3_young.ply
python tools/exp/config3_rate.py 7000 $ply > $out/train_7000.json 2> $out/train.err
python tools/exp/config3_rate.py 1500 $young > $out/train_1500.json 2>> $out/train.err
run() {  # label, args...
  local label=$1; shift
  python bench.py "$@" --steps 100 --warmup 20 --train-iters 0 --no-cpu-baseline --no-pmc --no-synced-regions 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k = d['kernels']
print('$label', 'fwd', k['raster_fwd']['ms'], 'bwd', k['raster_bwd']['ms'], 'sum', round(k['raster_fwd']['ms'] + k['raster_bwd']['ms'], 4), 'wall', d['ms_per_step'])"
}
for scene in "ply:$ply" "ply:$young" room; do
  for t in '{}' '{"deep_order_grid": 0}' '{"deep_order_grid": 0, "deep_tail": 0}'; do
    GSR_TUNE="$t" run "$scene 480x270 $t" --scene $scene --gaussians 300000 --width 480 --height 270
  done
done 2>&1 | tee $out/order_small_grids.txt
