# A/B of k_scan's scalar-masked ragged tile and its position-mask shortcut; the ragged-tile parity tests
mkdir -p gpurun_out/r03t
PYTHONPATH=. timeout 150 python tools/ab_bench.py --rounds 3 --steps 4 ab_libs/new14.so ab_libs/new15.so ab_libs/new15a.so ab_libs/new15b.so > gpurun_out/r03t/ab_c3.txt 2>&1; tail -5 gpurun_out/r03t/ab_c3.txt
PYTHONPATH=. timeout 120 python tools/ab_bench.py --median-len 2000 --rounds 2 --steps 4 ab_libs/new14.so ab_libs/new15.so > gpurun_out/r03t/ab_c3_2k.txt 2>&1; tail -3 gpurun_out/r03t/ab_c3_2k.txt
timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ragged or ont_like or adversarial or edge" > gpurun_out/r03t/gpu_tests.log 2>&1; grep -E "passed|failed" gpurun_out/r03t/gpu_tests.log
