set -x
mkdir -p gpurun_out/r03a
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_bench_workload" > gpurun_out/r03a/test_bench_workloads.log 2>&1; tail -5 gpurun_out/r03a/test_bench_workloads.log
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/r03a/bench_default.json 2> gpurun_out/r03a/bench_default.err; tail -c 600 gpurun_out/r03a/bench_default.err
timeout 300 python bench.py --workload c5_hifi64 --steps 5 --warmup 1 --cpu-bases 0 --e2e-reads 0 > gpurun_out/r03a/bench_c5.json 2> gpurun_out/r03a/bench_c5.err
timeout 300 bash tools/prof_sections.sh > gpurun_out/r03a/prof_sections.txt 2>&1
timeout 300 bash tools/prof_sections.sh 1000000 2000 > gpurun_out/r03a/prof_sections_2k.txt 2>&1
df -h /dev/shm /tmp > gpurun_out/r03a/df.txt; nproc >> gpurun_out/r03a/df.txt; cat /sys/fs/cgroup/memory.max /sys/fs/cgroup/cpu.max >> gpurun_out/r03a/df.txt 2>&1; free -g >> gpurun_out/r03a/df.txt
