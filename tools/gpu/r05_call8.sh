B="python bench.py --workload c5_hifi64 --steps 6 --warmup 2 --cpu-bases 0 --e2e-reads 0 --parity-reads 0"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"],1), round(d["ms_per_step"],3), {k: round(v,2) for k,v in d["roofline"]["kernel_ms"].items()})'
echo full; FPL_NO_TRIM_AHEAD=1 timeout 300 $B 2>/dev/null | python -c "$P"
echo no_filter_no_exact_2048; FPL_DEBUG_FLAGS=2048 timeout 300 $B 2>/dev/null | python -c "$P"
echo filter_no_exact_1024; FPL_DEBUG_FLAGS=1024 timeout 300 $B 2>/dev/null | python -c "$P"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/pm1 -- python $GRAFT_REPO_ROOT/bench.py --workload c5_hifi64 --steps 2 --warmup 1 --cpu-bases 0 --e2e-reads 0 --parity-reads 0 > /tmp/pm1.log 2>&1
python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pm1/**/*counter_collection.csv', recursive=True):
    disp={}
    for r in csv.DictReader(open(f)):
        k=(r['Dispatch_Id'], r['Kernel_Name'].split('(')[0].replace('void fpl::',''), r['Counter_Name'])
        disp[k]=disp.get(k,0)+float(r['Counter_Value'])
    for (d,kn,c),v in disp.items(): acc[kn][c].append(v)
for kn,d in acc.items():
    if kn.startswith('k_trim') or kn.startswith('k_scan'):
        print(kn, {c: round(sum(v)/len(v)/1e6,1) for c,v in d.items()})
PY
