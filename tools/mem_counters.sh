#!/bin/bash
# profiling only: cache / address-translation counters per kernel for the default bench batch (one --pmc pass per group)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/memc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --cpu-bases 0 --e2e-reads 0 --steps 2 --warmup 1 $*"
i=0
for grp in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_LATENCY_sum" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_EA_RD_UNCACHED_32B_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- $B > $OUT/p$i.log 2>&1 || tail -3 $OUT/p$i.log
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    disp = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fpl::", "")
        if not n.startswith("k_"):
            continue
        disp[(r["Dispatch_Id"], n, r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, n, c), v in disp.items():
        acc[n][c].append(v)
for n in sorted(acc):
    print(n, {c: "%.3g" % (sum(v) / len(v)) for c, v in sorted(acc[n].items())})
PY
