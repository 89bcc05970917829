# sweep of the statistics passes' tuning variables on the final kernels (same library, same batch)
mkdir -p gpurun_out/r03v
L=fastplong_amd/libfastplong_amd.so
PYTHONPATH=. timeout 250 python tools/ab_bench.py --rounds 3 --steps 4 $L $L%FPL_STATS_EXTRA_BLOCKS=16 $L%FPL_STATS_EXTRA_BLOCKS=8 $L%FPL_STATS_MIN_BUCKET=128 $L%FPL_STATS_MIN_BUCKET=64 $L%FPL_STATS_MIN_BUCKET=512 > gpurun_out/r03v/sweep_c3.txt 2>&1; tail -7 gpurun_out/r03v/sweep_c3.txt
