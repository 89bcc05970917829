# A/B: previous-lane dword through DPP, add-with-carry in sliced_max, the deferred wait of the scalar-masked ragged tile
mkdir -p gpurun_out/r03u
PYTHONPATH=. timeout 200 python tools/ab_bench.py --rounds 3 --steps 4 ab_libs/new14.so ab_libs/new15.so ab_libs/new16.so ab_libs/new16a.so ab_libs/new16b.so ab_libs/new16c.so > gpurun_out/r03u/ab_c3.txt 2>&1; tail -7 gpurun_out/r03u/ab_c3.txt
PYTHONPATH=. timeout 120 python tools/ab_bench.py --median-len 2000 --rounds 2 --steps 4 ab_libs/new14.so ab_libs/new16.so ab_libs/new16c.so > gpurun_out/r03u/ab_c3_2k.txt 2>&1; tail -3 gpurun_out/r03u/ab_c3_2k.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r03u/gpu_tests.log 2>&1; grep -E "passed|failed" gpurun_out/r03u/gpu_tests.log
