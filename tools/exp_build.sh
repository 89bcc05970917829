#!/bin/bash
# profiling only: run quick_bench with the library rebuilt with extra -D flags, then restore it
cp fastplong_amd/libfastplong_amd.so /tmp/lib.keep
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-atomic-optimizer-strategy=None $1 -o fastplong_amd/libfastplong_amd.so fastplong_amd/csrc/fpl_hip.hip 2>&1 | grep error
shift
bash tools/quick_bench.sh "$@"
cp /tmp/lib.keep fastplong_amd/libfastplong_amd.so
