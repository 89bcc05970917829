#!/bin/bash
# short reads (4 M x 2 kb, 1 M x 2 kb): how many blocks per CU the end trims get beside k_scan, and the old order
run() { # reads median env...
  local r=$1 m=$2; shift 2
  env "$@" python bench.py --reads $r --median-len $m --steps 8 --warmup 2 --cpu-bases 0 --e2e-reads 0 --parity-reads 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$r x $m', '$*', round(d['value'],1), d['ms_per_step'], {k: round(v,2) for k,v in d['roofline']['kernel_ms'].items()}, d['roofline'].get('kernel_ms_alone'))"
}
for cfg in "4000000 2000" "1000000 2000"; do
  set -- $cfg
  run $1 $2 FPL_TRIM_AHEAD_GATE=1
  for b in 2 3 4 5; do run $1 $2 FPL_TRIM_AHEAD_BLOCKS=$b; done
  run $1 $2 FPL_TRIM_AHEAD_GATE=1
  run $1 $2 FPL_TRIM_AHEAD_BLOCKS=2
done
