# A/B: four reads per thread in the bucket kernels of the sorted statistics pass (4 M reads of 2 kb; 1 M of 8 kb)
mkdir -p gpurun_out/r03y
PYTHONPATH=. timeout 100 python tools/ab_bench.py --reads 4000000 --median-len 2000 --rounds 2 --steps 3 ab_libs/base.so ab_libs/new18.so > gpurun_out/r03y/ab_2k_4M.txt 2>&1; tail -2 gpurun_out/r03y/ab_2k_4M.txt
PYTHONPATH=. timeout 100 python tools/ab_bench.py --rounds 2 --steps 4 ab_libs/base.so ab_libs/new18.so > gpurun_out/r03y/ab_c3.txt 2>&1; tail -2 gpurun_out/r03y/ab_c3.txt
