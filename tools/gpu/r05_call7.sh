O=gpurun_out/r05_c7
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ahead or async or batch_forms or capacity_growth or large_batch or bench_sized" > $O/t.log 2>&1; tail -2 $O/t.log
B="python bench.py --steps 20 --warmup 5 --cpu-bases 0 --e2e-reads 0 --parity-reads 100000"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"],1), round(d["ms_per_step"],3), {k: round(v,2) for k,v in d["roofline"]["kernel_ms"].items()}, d.get("parity_sample"))'
echo default; timeout 300 $B 2>/dev/null | python -c "$P"
echo no_tail_detach; FPL_NO_TAIL_DETACH=1 timeout 300 $B 2>/dev/null | python -c "$P"
echo gate0; FPL_TRIM_AHEAD_GATE=0 timeout 300 $B 2>/dev/null | python -c "$P"
echo gate0_no_detach; FPL_TRIM_AHEAD_GATE=0 FPL_NO_TAIL_DETACH=1 timeout 300 $B 2>/dev/null | python -c "$P"
echo none; FPL_NO_TRIM_AHEAD=1 timeout 300 $B 2>/dev/null | python -c "$P"
echo c4 default; timeout 300 $B --workload c4_mixed --steps 10 2>/dev/null | python -c "$P"
echo c4 gate0; FPL_TRIM_AHEAD_GATE=0 timeout 300 $B --workload c4_mixed --steps 10 2>/dev/null | python -c "$P"
echo c4 none; FPL_NO_TRIM_AHEAD=1 timeout 300 $B --workload c4_mixed --steps 10 2>/dev/null | python -c "$P"
echo c5 default; timeout 300 $B --workload c5_hifi64 --steps 10 2>/dev/null | python -c "$P"
echo c5 gate0; FPL_TRIM_AHEAD_GATE=0 timeout 300 $B --workload c5_hifi64 --steps 10 2>/dev/null | python -c "$P"
echo 2kb default; timeout 300 $B --reads 4000000 --median-len 2000 --steps 8 --parity-reads 0 2>/dev/null | python -c "$P"
echo 2kb gate0; FPL_TRIM_AHEAD_GATE=0 timeout 300 $B --reads 4000000 --median-len 2000 --steps 8 --parity-reads 0 2>/dev/null | python -c "$P"
