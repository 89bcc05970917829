timeout 900 python bench.py --steps 5 --warmup 2 --cpu-bases 0 --parity-reads 0 > gpurun_out/r05_c11.json 2> gpurun_out/r05_c11.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_c11.json').read().strip().splitlines()[-1])
b=d['e2e'].get('large_input',{})
print(b.get('error'), b.get('copies'))
for k,v in (b.get('cli') or {}).items(): print(k, v.get('rc'), round(v.get('value') or 0,2), v.get('pipeline_value') and round(v['pipeline_value'],2), round(v.get('process_seconds'),2), [s for s in v['stages'] if s.startswith('host pipeline')][:1])
PY
