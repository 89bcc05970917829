O=gpurun_out/r05_c9
mkdir -p $O
PYTHONPATH=. timeout 300 python tools/ab_bench.py --rounds 2 --steps 3 --workload c5_hifi64 --reads 500000 ab_libs/old.so ab_libs/new.so 2>&1 | grep -E "total|differ"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fasta or hifi or adapter_lengths or every_bench_workload or quality_byte_range_fasta" > $O/t.log 2>&1; tail -3 $O/t.log
FPL_FUZZ_FASTA=200 timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 8 -k random_fasta_sets > $O/soak_fasta.log 2>&1; tail -1 $O/soak_fasta.log
