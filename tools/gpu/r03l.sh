mkdir -p gpurun_out/r03l
PYTHONPATH=. timeout 200 python tools/ab_bench.py ab_libs/new8.so ab_libs/new8_abl64.so > gpurun_out/r03l/ab_c3.txt 2>&1; tail -3 gpurun_out/r03l/ab_c3.txt
