O=gpurun_out/r05_c4
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ahead or async or batch_forms or capacity_growth or large_batch" > $O/t.log 2>&1; tail -3 $O/t.log
PYTHONPATH=. timeout 200 python tools/ab_bench.py --rounds 3 --steps 6 ab_libs/cur.so ab_libs/cur.so%AHEAD=1 2>&1 | grep -E "total|differ"
PYTHONPATH=. timeout 200 python tools/ab_bench.py --rounds 3 --steps 4 --median-len 2000 --reads 4000000 ab_libs/cur.so ab_libs/cur.so%AHEAD=1 2>&1 | grep -E "total|differ"
PYTHONPATH=. timeout 200 python tools/ab_bench.py --rounds 2 --steps 4 --workload c5_hifi64 --reads 500000 ab_libs/cur.so ab_libs/cur.so%AHEAD=1 2>&1 | grep -E "total|differ"
PYTHONPATH=. timeout 200 python tools/ab_bench.py --rounds 2 --steps 3 --workload c4_mixed ab_libs/cur.so ab_libs/cur.so%AHEAD=1 2>&1 | grep -E "total|differ"
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-bases 0 --e2e-reads 0 --parity-reads 100000 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('parity_sample'))"
FPL_NO_TRIM_AHEAD=1 timeout 300 python bench.py --steps 20 --warmup 5 --cpu-bases 0 --e2e-reads 0 --parity-reads 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
