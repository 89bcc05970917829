set -x
mkdir -p gpurun_out/r03b
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r03b/gpu_tests.log 2>&1; tail -5 gpurun_out/r03b/gpu_tests.log
PYTHONPATH=. timeout 600 python tools/ab_bench.py ab_libs/old.so ab_libs/new.so > gpurun_out/r03b/ab_c3.txt 2>&1; tail -3 gpurun_out/r03b/ab_c3.txt
PYTHONPATH=. timeout 600 python tools/ab_bench.py --median-len 2000 ab_libs/old.so ab_libs/new.so > gpurun_out/r03b/ab_c3_2k.txt 2>&1; tail -3 gpurun_out/r03b/ab_c3_2k.txt
PYTHONPATH=. timeout 600 python tools/ab_bench.py --workload c5_hifi64 --reads 500000 ab_libs/old.so ab_libs/new.so > gpurun_out/r03b/ab_c5.txt 2>&1; tail -3 gpurun_out/r03b/ab_c5.txt
PYTHONPATH=. timeout 600 python tools/ab_bench.py --workload c4_mixed --reads 2500000 ab_libs/old.so ab_libs/new.so > gpurun_out/r03b/ab_c4.txt 2>&1; tail -3 gpurun_out/r03b/ab_c4.txt
