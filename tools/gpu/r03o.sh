mkdir -p gpurun_out/r03o
PYTHONPATH=. timeout 300 python tools/ab_bench.py --workload c5_hifi64 --reads 500000 --rounds 2 --steps 3 ab_libs/new10_abl.so@0 ab_libs/new10_abl.so@1024 ab_libs/new10_abl.so@2048 > gpurun_out/r03o/ab_c5_abl.txt 2>&1; tail -4 gpurun_out/r03o/ab_c5_abl.txt
