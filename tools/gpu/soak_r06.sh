# round 6: seeded soak of the final kernels on one box (through gpurun) with the statistics pass sorted for every batch
# (FPL_STATS_MIN_BUCKET: the 6-mer table, the kept k-mer tables, the byte-by-byte rows), seeds no earlier soak used
O=gpurun_out/r06_soak
mkdir -p $O
run() { name=$1; shift; env "$@" timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 8 -k random_option_sets > $O/$name.log 2>&1; echo "$name: $(tail -1 $O/$name.log)"; }
run options_default_a FPL_FUZZ_FROM=1300000 FPL_FUZZ_SEEDS=5000
run options_default_b FPL_FUZZ_FROM=1305000 FPL_FUZZ_SEEDS=5000
run options_minbucket1_a FPL_STATS_MIN_BUCKET=1 FPL_FUZZ_FROM=1350000 FPL_FUZZ_SEEDS=5000
run options_minbucket1_b FPL_STATS_MIN_BUCKET=1 FPL_FUZZ_FROM=1355000 FPL_FUZZ_SEEDS=5000
run options_minbucket2_per64 FPL_STATS_MIN_BUCKET=2 FPL_STATS_PER=64 FPL_FUZZ_FROM=1360000 FPL_FUZZ_SEEDS=5000
run options_groups FPL_STATS_MIN_BUCKET=2 FPL_STATS_PER=64 FPL_STATS_HI_TILE=2 FPL_STATS_GROUP=3 FPL_STATS_GROUP_ROWS=70 FPL_FUZZ_FROM=1365000 FPL_FUZZ_SEEDS=5000
FPL_FUZZ_FASTA=400 FPL_FUZZ_FASTA_FROM=9000 timeout 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 8 -k random_fasta_sets > $O/fasta.log 2>&1; echo "fasta sets 9000..9399: $(tail -1 $O/fasta.log)"
