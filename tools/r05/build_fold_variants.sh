#!/bin/bash
# Builds libgsraster_fold{1,2,3}.so: the library with the compositing backward's row fold on the matrix pipe (1),
# through LDS (2), through LDS for the lanes of a row as well (3) -- raster_bwd.hip: GSR_BWD_FOLD.  Loaded through
# GSR_LIBRARY by tools/r05/fold_ab.sh.  (hipcc cross-compiles without a GPU; the .so files travel with gpurun.)
set -e
cd "$(dirname "$0")/../../gaussian-splatting-toolkit_amd/csrc"
make -s >/dev/null
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -Wall -Wno-unused-result -munsafe-fp-atomics -fno-slp-vectorize"
others=$(ls *.o | grep -v '^raster_bwd')
for f in 1 2 3; do
  extra=""; [ $f = 1 ] && extra="-DGSR_BWD_MIN_WAVES=4"   # 125 VGPRs: keep the accumulators out of AGPRs (4 waves per SIMD)
  /opt/rocm/bin/hipcc $FLAGS -DGSR_BWD_FOLD=$f $extra -c raster_bwd.hip -o /tmp/raster_bwd_fold$f.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/r05/libgsraster_fold$f.so $others /tmp/raster_bwd_fold$f.o
done
ls -la ../../tools/r05/libgsraster_fold*.so
