mkdir -p gpurun_out/r03r
timeout 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r03r/gpu_tests.log 2>&1; grep -E "passed|failed" gpurun_out/r03r/gpu_tests.log
PYTHONPATH=. timeout 200 python tools/ab_bench.py --workload c5_hifi64 --reads 500000 --rounds 2 --steps 3 ab_libs/new12.so ab_libs/new13.so > gpurun_out/r03r/ab_c5.txt 2>&1; tail -3 gpurun_out/r03r/ab_c5.txt
