O=gpurun_out/r05_c6
mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; grep -E "passed|failed" $O/gpu_tests.log | tail -2
PYTHONPATH=. timeout 200 python tools/ab_bench.py --rounds 3 --steps 6 ab_libs/cur.so ab_libs/cur.so%AHEAD=1 ab_libs/cur.so%AHEAD=1,FPL_TRIM_AHEAD_GATE=0 2>&1 | grep -E "total|differ"
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-bases 0 --e2e-reads 0 --parity-reads 100000 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('parity_sample'))"
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-bases 0 --e2e-reads 300000 --e2e-copies 0 --parity-reads 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], {k:(round(v['value'],2), v['pipeline_value'] and round(v['pipeline_value'],2)) for k,v in d['e2e']['cli'].items()}, d['e2e'].get('pcie_call',{}).get('value'))"
