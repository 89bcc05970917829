# one more seeded soak of the final kernels on one box (through gpurun): every -m gpu test, then random option sets and random FASTA
# adapter sets from seeds no earlier soak used
#   bash tools/gpu/soak_extra.sh [<first option seed> [<chunks of 5000> [<first FASTA seed> [<FASTA sets>]]]]
O=gpurun_out/r05_soak
F=${1:-900000}; N=${2:-8}; FF=${3:-3000}; FN=${4:-700}
mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; grep -E "passed|failed" $O/gpu_tests.log | tail -1
for c in $(seq 0 $((N - 1))); do
  FPL_FUZZ_FROM=$((F + c * 5000)) FPL_FUZZ_SEEDS=5000 timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 8 -k random_option_sets > $O/soak_options_$c.log 2>&1; tail -1 $O/soak_options_$c.log
done
FPL_STATS_MIN_BUCKET=1 FPL_FUZZ_FROM=$((F + 50000)) FPL_FUZZ_SEEDS=5000 timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 8 -k random_option_sets > $O/soak_minbucket1.log 2>&1; tail -1 $O/soak_minbucket1.log
FPL_FUZZ_FASTA=$FN FPL_FUZZ_FASTA_FROM=$FF timeout 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 8 -k random_fasta_sets > $O/soak_fasta.log 2>&1; tail -1 $O/soak_fasta.log
