mkdir -p gpurun_out/r03m
timeout 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r03m/gpu_tests.log 2>&1; grep -E "passed|failed" gpurun_out/r03m/gpu_tests.log
PYTHONPATH=. timeout 200 python tools/ab_bench.py ab_libs/new8.so ab_libs/new9.so ab_libs/new9_c16.so ab_libs/new9_c8.so > gpurun_out/r03m/ab_c3.txt 2>&1; tail -5 gpurun_out/r03m/ab_c3.txt
