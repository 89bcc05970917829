#!/bin/bash
# measurement aid: build kernel variants of libfastplong_amd.so side by side (gpurun_out/ab/<name>.so)
#   tools/ab_build.sh name "-DFPL_OPT_HIST=0 ..." [source-dir]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FLAGS=$2; SRC=${3:-$ROOT/fastplong_amd/csrc}
mkdir -p $ROOT/ab_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-atomic-optimizer-strategy=None $FLAGS \
    -I$ROOT/include -o $ROOT/ab_libs/$NAME.so $SRC/fpl_hip.hip 2>&1 | grep -E "error" || true
ls -la $ROOT/ab_libs/$NAME.so
