#!/bin/bash
# The first stage of the schedule (480 x 270 = 510 tiles): depth-segment knobs against config 3's rate.
out=gpurun_out/smallgrid; mkdir -p $out
for v in "GSR_NOTHING=1" "GSR_DEPTH_SEGMENTS_FWD=12" "GSR_DEPTH_SEGMENTS_FWD=16" "GSR_DEPTH_SEGMENTS_MIN=256" "GSR_DEPTH_SEGMENTS_MIN=1024" "GSR_DEPTH_SEGMENTS_FWD=4" "GSR_DEPTH_SEGMENTS=8" "GSR_NOTHING=2"; do
  echo "config3 $v: $(env $v python tools/exp/config3_rate.py 7000 2>/dev/null | tail -1)"
done | tee $out/config3_segments.txt
