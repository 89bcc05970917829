mkdir -p gpurun_out/r03j
timeout 120 python tools/hang_probe.py ab_libs/new7.so > gpurun_out/r03j/probe.log 2>&1; tail -1 gpurun_out/r03j/probe.log
PYTHONPATH=. timeout 200 python tools/ab_bench.py ab_libs/new6.so ab_libs/new7.so ab_libs/new7_wps8.so ab_libs/new7_r16.so > gpurun_out/r03j/ab_c3.txt 2>&1; tail -5 gpurun_out/r03j/ab_c3.txt
OUT=$PWD/gpurun_out/r03j; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --cpu-bases 0 --e2e-reads 0 --parity-reads 0 --steps 3 --warmup 1"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_c3.csv; rm -rf $OUT/stats
