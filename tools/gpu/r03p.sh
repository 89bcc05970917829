mkdir -p gpurun_out/r03p
timeout 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r03p/gpu_tests.log 2>&1; grep -E "passed|failed" gpurun_out/r03p/gpu_tests.log
PYTHONPATH=. timeout 200 python tools/ab_bench.py ab_libs/new10.so ab_libs/new11.so > gpurun_out/r03p/ab_c3.txt 2>&1; tail -3 gpurun_out/r03p/ab_c3.txt
PYTHONPATH=. timeout 200 python tools/ab_bench.py --median-len 2000 ab_libs/new10.so ab_libs/new11.so > gpurun_out/r03p/ab_c3_2k.txt 2>&1; tail -3 gpurun_out/r03p/ab_c3_2k.txt
