mkdir -p gpurun_out/r03i
timeout 600 python -m pytest tests -x -q -m gpu > gpurun_out/r03i/gpu_tests.log 2>&1; grep -E "passed|failed" gpurun_out/r03i/gpu_tests.log
PYTHONPATH=. timeout 200 python tools/ab_bench.py --workload c5_hifi64 --reads 500000 ab_libs/new5.so ab_libs/new6.so > gpurun_out/r03i/ab_c5.txt 2>&1; tail -3 gpurun_out/r03i/ab_c5.txt
PYTHONPATH=. timeout 200 python tools/ab_bench.py ab_libs/new5.so ab_libs/new6.so ab_libs/new6_noflush.so > gpurun_out/r03i/ab_c3.txt 2>&1; tail -4 gpurun_out/r03i/ab_c3.txt
timeout 400 python bench.py --steps 10 --warmup 2 --cpu-bases 0 > gpurun_out/r03i/bench_default.json 2> gpurun_out/r03i/bench_default.err; tail -c 300 gpurun_out/r03i/bench_default.err
