mkdir -p gpurun_out/r03q
timeout 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r03q/gpu_tests.log 2>&1; grep -E "passed|failed" gpurun_out/r03q/gpu_tests.log
PYTHONPATH=. timeout 200 python tools/ab_bench.py ab_libs/new11.so ab_libs/new12.so ab_libs/new12_w6.so ab_libs/new12_w5.so > gpurun_out/r03q/ab_c3.txt 2>&1; tail -5 gpurun_out/r03q/ab_c3.txt
