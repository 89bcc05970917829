# measurement aid: the end trims of batch k + 1 beside k_scan of batch k (FPL_TRIM_AHEAD_GATE=0) with a smaller grid (FPL_TRIM_AHEAD_BLOCKS per CU)
for wl in ${WLS:-c3_full_pipeline c2_adapter_only c4_mixed c5_hifi64}; do
for cfg in "1 0" "0 2" "1 0" "0 2"; do set -- $cfg
  FPL_TRIM_AHEAD_GATE=$1 FPL_TRIM_AHEAD_BLOCKS=$2 python bench.py --workload $wl --steps 12 --warmup 3 --e2e-reads 0 --cpu-bases 0 --parity-reads 0 --full-json "" ${EXTRA:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl gate $1 blocks $2:', round(d['value'],1), round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['roofline']['kernel_ms'].items()})"
done; done
