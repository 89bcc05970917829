set -x
mkdir -p gpurun_out/r03e
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r03e/gpu_tests.log 2>&1; grep -E "passed|failed" gpurun_out/r03e/gpu_tests.log
PYTHONPATH=. timeout 600 python tools/ab_bench.py ab_libs/new2.so ab_libs/new3.so > gpurun_out/r03e/ab_c3.txt 2>&1; tail -3 gpurun_out/r03e/ab_c3.txt
PYTHONPATH=. timeout 600 python tools/ab_bench.py --median-len 2000 ab_libs/new2.so ab_libs/new3.so > gpurun_out/r03e/ab_c3_2k.txt 2>&1; tail -3 gpurun_out/r03e/ab_c3_2k.txt
PYTHONPATH=. timeout 600 python tools/ab_bench.py --workload c5_hifi64 --reads 500000 ab_libs/old.so ab_libs/new3.so > gpurun_out/r03e/ab_c5.txt 2>&1; tail -3 gpurun_out/r03e/ab_c5.txt
OUT=$PWD/gpurun_out/r03e; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --cpu-bases 0 --e2e-reads 0 --parity-reads 0 --steps 3 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_c3.csv; rm -rf $OUT/stats
