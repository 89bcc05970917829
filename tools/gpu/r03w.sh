# A/B: k_scan touches the head of the next read in the last tile of the current one
mkdir -p gpurun_out/r03w
PYTHONPATH=. timeout 100 python tools/ab_bench.py --rounds 3 --steps 4 ab_libs/new16.so ab_libs/new17.so > gpurun_out/r03w/ab_c3.txt 2>&1; tail -3 gpurun_out/r03w/ab_c3.txt
PYTHONPATH=. timeout 100 python tools/ab_bench.py --median-len 2000 --rounds 3 --steps 4 ab_libs/new16.so ab_libs/new17.so > gpurun_out/r03w/ab_c3_2k.txt 2>&1; tail -2 gpurun_out/r03w/ab_c3_2k.txt
