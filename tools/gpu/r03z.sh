mkdir -p gpurun_out/r03z
PYTHONPATH=. timeout 100 python tools/ab_bench.py --reads 4000000 --median-len 2000 --rounds 2 --steps 3 ab_libs/new18.so ab_libs/new18p8.so ab_libs/new18p16.so > gpurun_out/r03z/ab_2k_4M.txt 2>&1; tail -3 gpurun_out/r03z/ab_2k_4M.txt
