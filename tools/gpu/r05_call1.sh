# round 5, first call: the whole -m gpu suite (with the new quality-range tests), smoke, the issue-model micro-benchmark, the default bench line
O=gpurun_out/r05_c1
mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( cd tools/ubench && timeout 300 ./valu_mix2 > ../../$O/valu_mix2.txt 2>&1 ); tail -25 $O/valu_mix2.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json,sys
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('parity_sample'))
print({k:(v.get('value'), v.get('pipeline_value')) for k,v in d['e2e']['cli'].items()})
"
