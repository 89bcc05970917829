set -x
mkdir -p gpurun_out/r03f
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r03f/gpu_tests.log 2>&1; grep -E "passed|failed" gpurun_out/r03f/gpu_tests.log
PYTHONPATH=. timeout 600 python tools/ab_bench.py ab_libs/new3.so ab_libs/new4.so > gpurun_out/r03f/ab_c3.txt 2>&1; tail -3 gpurun_out/r03f/ab_c3.txt
OUT=$PWD/gpurun_out/r03f; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --cpu-bases 0 --e2e-reads 0 --parity-reads 0 --steps 3 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_c3.csv; rm -rf $OUT/stats
