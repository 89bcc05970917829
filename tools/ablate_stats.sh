#!/bin/bash
# profiling only: rebuild the library on the GPU box with pieces of k_stats compiled out and time each variant
cp fastplong_amd/libfastplong_amd.so /tmp/lib.keep
for abl in 0 4 6 7 3 1 2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DFPL_ABL=$abl -o fastplong_amd/libfastplong_amd.so fastplong_amd/csrc/fpl_hip.hip 2>/dev/null
  echo "ABL=$abl"; bash tools/quick_bench.sh 1000000
done
cp /tmp/lib.keep fastplong_amd/libfastplong_amd.so
