O=gpurun_out/r05_c2
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batch_forms or quality_byte" > $O/t.log 2>&1; tail -2 $O/t.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_c2/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
e=d['e2e']
print(e.get('gz_inputs'))
for k,v in e['cli'].items(): print(k, v.get('rc'), round(v.get('value') or 0,2), v.get('pipeline_value') and round(v['pipeline_value'],2), v.get('ok'), [s for s in v.get('stages',[]) if s.startswith(('kernel forms','input','counter'))])
b=e.get('large_input',{})
print(b.get('error'), b.get('copies'))
for k,v in (b.get('cli') or {}).items(): print(k, v.get('rc'), round(v.get('value') or 0,2), v.get('pipeline_value') and round(v['pipeline_value'],2), v.get('process_seconds'), [s for s in v.get('stages',[]) if s.startswith(('kernel forms','counter'))], v.get('stderr_tail'))
PY
