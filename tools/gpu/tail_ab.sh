# measurement aid: the tail of a batch (k_stats_reduce_sorted, the post-only pass's reduce) left on the side stream beside the next batch's
# k_scan (default) against joined into the main stream (FPL_NO_TAIL_DEFER=1)
for wl in ${WLS:-c3_full_pipeline c2_adapter_only c4_mixed c5_hifi64}; do
for nd in 1 0 1 0; do
  FPL_NO_TAIL_DEFER=$nd python bench.py --workload $wl --steps 12 --warmup 3 --e2e-reads 0 --cpu-bases 0 --parity-reads 0 --full-json "" ${EXTRA:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', 'tail joined ' if $nd else 'tail deferred', round(d['value'],1), round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['roofline']['kernel_ms'].items()}, d['counters_check']['ok'])"
done; done
