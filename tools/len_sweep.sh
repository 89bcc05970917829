#!/bin/bash
# per-read vs per-base cost of the kernels: same total bases, different read lengths
for cfg in "4000000 2000" "2000000 4000" "1000000 8000" "500000 16000" "250000 32000"; do
  set -- $cfg
  python bench.py --reads $1 --median-len $2 --steps 5 --warmup 1 --cpu-bases 0 --e2e-reads 0 --parity-reads 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['reads_per_gpu'], d['config']['bases_per_gpu'], round(d['value'],1), {k: round(v,2) for k,v in d['roofline']['kernel_ms'].items()})"
done
