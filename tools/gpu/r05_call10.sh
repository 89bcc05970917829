O=gpurun_out/r05_c10
mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; grep -E "passed|failed" $O/gpu_tests.log | tail -2
FPL_FUZZ_FASTA=300 FPL_FUZZ_FASTA_FROM=1000 timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 8 -k random_fasta_sets > $O/soak_fasta.log 2>&1; tail -1 $O/soak_fasta.log
PYTHONPATH=. timeout 300 python tools/ab_bench.py --rounds 2 --steps 3 --workload c5_hifi64 --reads 500000 ab_libs/old.so ab_libs/new.so 2>&1 | grep -E "total|differ"
