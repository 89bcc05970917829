#!/bin/bash
# measurement aid: bench.py (one engine, pipelined as the driver runs it) under several settings of ONE environment variable, alternating
#   bash tools/gpu/ab_env.sh <NAME> "<v1> <v2> ..." [workload [rounds]]      ("-" = unset)
export PYTHONPATH=.
NAME=$1; VALS=$2; WL=${3:-c3_full_pipeline}; R=${4:-2}
for i in $(seq 1 $R); do
for v in $VALS; do
if [ "$v" = "-" ]; then unset $NAME; else export $NAME=$v; fi
timeout 200 python bench.py --workload $WL --cpu-bases 0 --e2e-reads 0 --parity-reads 0 --steps 12 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$NAME=$v', round(d['value'],1), round(d['ms_per_step'],3), {k: round(x,2) for k,x in d['roofline']['kernel_ms'].items()})"
done; done
