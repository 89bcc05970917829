#!/bin/bash
# profiling only: rebuild with section timers on the GPU box, print the breakdown, restore the library
cp fastplong_amd/libfastplong_amd.so /tmp/lib.keep
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DFPL_PROF -o fastplong_amd/libfastplong_amd.so fastplong_amd/csrc/fpl_hip.hip 2>&1 | grep -E "error" 
PYTHONPATH=. python tools/prof_sections.py "$@"
cp /tmp/lib.keep fastplong_amd/libfastplong_amd.so
