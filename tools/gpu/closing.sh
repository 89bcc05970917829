# The closing measurements of a round on one GPU box (through gpurun; every step under `timeout`):
#   bash tools/gpu/closing.sh <tag> [seeds of the option soak] [what: all | tests | profiles | soak]
# tests:    every -m gpu test, the smoke entry point
# profiles: profiles/collect_profile.sh for the four bench workloads (kernel stats, HBM / SQ / LDS counters, bench lines), the default
#           bench line as the driver runs it, one c5 line with its end-to-end leg, the read-length sweep
# soak:     seeded soaks of the final kernels -- random option sets (pair packing on: FPL_SCAN_CHUNK=4 from tests/conftest.py; one
#           chunk with every front trim in slices of its own, one with 12 kb reads), random FASTA adapter sets
# Outputs under gpurun_out/<tag>/: copy the summaries to profiles/<tag>/, then `python profiles/summarize_profile.py --merge profiles/<tag>`.
TAG=${1:-closing}
SOAK=${2:-20000}
WHAT=${3:-all}
mkdir -p gpurun_out/$TAG
O=gpurun_out/$TAG
if [ $WHAT = all ] || [ $WHAT = tests ]; then
  timeout 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; grep -E "passed|failed" $O/gpu_tests.log
  timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
fi
if [ $WHAT = all ] || [ $WHAT = profiles ]; then
  for wl in c3_full_pipeline c2_adapter_only c4_mixed c5_hifi64; do
    timeout 700 bash profiles/collect_profile.sh $TAG $wl > $O/collect_$wl.log 2>&1; tail -1 $O/collect_$wl.log
  done
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
  timeout 400 python bench.py --workload c5_hifi64 --steps 10 --warmup 2 --cpu-bases 0 --parity-reads 0 --e2e-reads 200000 --e2e-copies 0 > $O/bench_c5_e2e.json 2> $O/bench_c5_e2e.err
  timeout 600 bash tools/len_sweep.sh > $O/len_sweep.txt 2>&1; cat $O/len_sweep.txt
  timeout 100 python bench.py --reads 1000000 --median-len 2000 --steps 5 --warmup 1 --cpu-bases 0 --e2e-reads 0 --parity-reads 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1M x 2 kb:', round(d['value'],1), {k: round(v,2) for k,v in d['roofline']['kernel_ms'].items()})" | tee -a $O/len_sweep.txt
fi
if [ $WHAT = all ] || [ $WHAT = soak ]; then
  Q=$((SOAK / 4))
  for c in 0 1 2 3; do
    FPL_FUZZ_FROM=$((400000 + c * Q)) FPL_FUZZ_SEEDS=$Q timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 8 -k random_option_sets > $O/soak_options_$c.log 2>&1
    tail -1 $O/soak_options_$c.log
  done
  FPL_STATS_MIN_BUCKET=1 FPL_FUZZ_FROM=500000 FPL_FUZZ_SEEDS=$Q timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 8 -k random_option_sets > $O/soak_minbucket1.log 2>&1; tail -1 $O/soak_minbucket1.log
  FPL_FUZZ_MEDIAN=12000 FPL_FUZZ_FROM=600000 FPL_FUZZ_SEEDS=$((SOAK / 12)) timeout 250 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 8 -k random_option_sets > $O/soak_median12k.log 2>&1; tail -1 $O/soak_median12k.log
  FPL_SCAN_CHUNK=16 FPL_FUZZ_MEDIAN=2500 FPL_FUZZ_FROM=700000 FPL_FUZZ_SEEDS=$((SOAK / 8)) timeout 250 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 8 -k random_option_sets > $O/soak_pairs2k5.log 2>&1; tail -1 $O/soak_pairs2k5.log
  FPL_FUZZ_FASTA=300 timeout 250 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 8 -k random_fasta_sets > $O/soak_fasta.log 2>&1; tail -1 $O/soak_fasta.log
fi
ls $O | head -60
