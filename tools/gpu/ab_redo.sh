#!/bin/bash
# measurement aid: k_redo beside the bucket kernels (default) against FPL_REDO_INLINE=1, through bench.py, alternating
export PYTHONPATH=.
for i in 1 2 3; do
for v in 0 1; do
FPL_REDO_INLINE=$v timeout 200 python bench.py --workload ${1:-c3_full_pipeline} --cpu-bases 0 --e2e-reads 0 --parity-reads 0 --steps 12 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inline=$v', round(d['value'],1), round(d['ms_per_step'],3), {k: round(x,2) for k,x in d['roofline']['kernel_ms'].items()}, {k: round(x,2) for k,x in d['roofline'].get('kernel_ms_in_line',{}).items()})"
done; done
