# the round's closing measurements on one GPU box: every -m gpu test, the smoke entry point, the profile summaries of the four
# bench workloads (profiles/collect_profile.sh), the default bench line as the driver runs it, one c5 line with its end-to-end leg,
# the 2 kb point of the read-length sweep, and seeded soaks of the final kernels (random option sets, random FASTA adapter sets)
TAG=${1:-r03_final}
SOAK=${2:-20000}
mkdir -p gpurun_out/$TAG
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/$TAG/gpu_tests.log 2>&1; grep -E "passed|failed" gpurun_out/$TAG/gpu_tests.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/$TAG/smoke.log 2>&1; tail -1 gpurun_out/$TAG/smoke.log
for wl in c3_full_pipeline c2_adapter_only c4_mixed c5_hifi64; do
  timeout 600 bash profiles/collect_profile.sh $TAG $wl > gpurun_out/$TAG/collect_$wl.log 2>&1; tail -1 gpurun_out/$TAG/collect_$wl.log
done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$TAG/bench_default.json 2> gpurun_out/$TAG/bench_default.err
timeout 400 python bench.py --workload c5_hifi64 --steps 10 --warmup 2 --cpu-bases 0 --e2e-reads 200000 --e2e-copies 0 > gpurun_out/$TAG/bench_c5_e2e.json 2> gpurun_out/$TAG/bench_c5_e2e.err
PYTHONPATH=. timeout 120 python tools/ab_bench.py --median-len 2000 --rounds 2 --steps 4 fastplong_amd/libfastplong_amd.so > gpurun_out/$TAG/len_2k.txt 2>&1; tail -1 gpurun_out/$TAG/len_2k.txt
# soaks: four chunks of random option sets (each with its own summary line), then random FASTA sets
Q=$((SOAK / 4))
for c in 0 1 2 3; do
  FPL_FUZZ_FROM=$((40000 + c * Q)) FPL_FUZZ_SEEDS=$Q timeout 150 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 8 -k random_option_sets > gpurun_out/$TAG/soak_options_$c.log 2>&1
  tail -1 gpurun_out/$TAG/soak_options_$c.log
done
FPL_FUZZ_FASTA=150 timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 8 -k random_fasta_sets > gpurun_out/$TAG/soak_fasta.log 2>&1; tail -1 gpurun_out/$TAG/soak_fasta.log
ls gpurun_out/$TAG | head -50
