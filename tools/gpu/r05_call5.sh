PYTHONPATH=. timeout 200 python tools/ab_bench.py --rounds 3 --steps 6 ab_libs/cur.so ab_libs/cur.so%AHEAD=1 ab_libs/cur.so%AHEAD=1,FPL_TRIM_AHEAD_GATE=0 2>&1 | grep -E "total|differ"
PYTHONPATH=. timeout 200 python tools/ab_bench.py --rounds 3 --steps 4 --median-len 2000 --reads 4000000 ab_libs/cur.so ab_libs/cur.so%AHEAD=1 ab_libs/cur.so%AHEAD=1,FPL_TRIM_AHEAD_GATE=0 2>&1 | grep -E "total|differ"
PYTHONPATH=. timeout 200 python tools/ab_bench.py --rounds 2 --steps 3 --workload c4_mixed ab_libs/cur.so ab_libs/cur.so%AHEAD=1 2>&1 | grep -E "total|differ"
