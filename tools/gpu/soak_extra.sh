O=gpurun_out/r05_soak
mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; grep -E "passed|failed" $O/gpu_tests.log | tail -1
for c in 0 1 2 3 4 5 6 7; do
  FPL_FUZZ_FROM=$((900000 + c * 5000)) FPL_FUZZ_SEEDS=5000 timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 8 -k random_option_sets > $O/soak_options_$c.log 2>&1; tail -1 $O/soak_options_$c.log
done
FPL_STATS_MIN_BUCKET=1 FPL_FUZZ_FROM=950000 FPL_FUZZ_SEEDS=5000 timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 8 -k random_option_sets > $O/soak_minbucket1.log 2>&1; tail -1 $O/soak_minbucket1.log
FPL_FUZZ_FASTA=700 FPL_FUZZ_FASTA_FROM=3000 timeout 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 8 -k random_fasta_sets > $O/soak_fasta.log 2>&1; tail -1 $O/soak_fasta.log
