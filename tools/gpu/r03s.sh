# A/B of k_trim_ends_batched's lane-per-read partial-pattern searches: 8 kb and 2 kb reads, and the new parity tests
mkdir -p gpurun_out/r03s
PYTHONPATH=. timeout 150 python tools/ab_bench.py --rounds 2 --steps 4 ab_libs/base.so ab_libs/new14.so ab_libs/new14w6.so > gpurun_out/r03s/ab_c3.txt 2>&1; tail -4 gpurun_out/r03s/ab_c3.txt
PYTHONPATH=. timeout 120 python tools/ab_bench.py --median-len 2000 --rounds 2 --steps 4 ab_libs/base.so ab_libs/new14.so ab_libs/new14w6.so > gpurun_out/r03s/ab_c3_2k.txt 2>&1; tail -4 gpurun_out/r03s/ab_c3_2k.txt
timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "partial_pattern or long_trim or adversarial or odd_command" > gpurun_out/r03s/gpu_tests.log 2>&1; grep -E "passed|failed" gpurun_out/r03s/gpu_tests.log
