# further seeded soaks of the final kernels: another seed range, every front trim in slices of its own (sorted statistics pass
# forced, FPL_STATS_MIN_BUCKET=1), longer reads (12 kb median: several scan tiles and cycle tiles per read, REDO items)
TAG=${1:-r03_soak2}
mkdir -p gpurun_out/$TAG
for c in 0 1 2 3; do
  FPL_FUZZ_FROM=$((100000 + c * 5000)) FPL_FUZZ_SEEDS=5000 timeout 120 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 8 -k random_option_sets > gpurun_out/$TAG/soak_options_b$c.log 2>&1
  tail -1 gpurun_out/$TAG/soak_options_b$c.log
done
FPL_STATS_MIN_BUCKET=1 FPL_FUZZ_FROM=200000 FPL_FUZZ_SEEDS=6000 timeout 150 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 8 -k random_option_sets > gpurun_out/$TAG/soak_minbucket1.log 2>&1; tail -1 gpurun_out/$TAG/soak_minbucket1.log
FPL_FUZZ_MEDIAN=12000 FPL_FUZZ_FROM=300000 FPL_FUZZ_SEEDS=1500 timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 8 -k random_option_sets > gpurun_out/$TAG/soak_median12k.log 2>&1; tail -1 gpurun_out/$TAG/soak_median12k.log
FPL_FUZZ_FASTA=400 timeout 150 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 8 -k random_fasta_sets > gpurun_out/$TAG/soak_fasta400.log 2>&1; tail -1 gpurun_out/$TAG/soak_fasta400.log
