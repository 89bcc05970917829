#!/bin/bash
# profiling only: build with the FPL_DEBUG_FLAGS switches compiled in and time k_scan with pieces disabled
# (1 no histogram atomics, 2 no filter sums, 8 no match counts, 16 no Hamming block at all)
cp fastplong_amd/libfastplong_amd.so /tmp/lib.keep
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DFPL_ABLATE -o fastplong_amd/libfastplong_amd.so fastplong_amd/csrc/fpl_hip.hip 2>/dev/null
for f in 0 1 2 3 8 16 19; do echo "FLAGS=$f"; FPL_DEBUG_FLAGS=$f bash tools/quick_bench.sh 1000000; done
cp /tmp/lib.keep fastplong_amd/libfastplong_amd.so
