#!/bin/bash
# profiling only: VALU / SALU / LDS instruction counts per kernel for a given bench configuration
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sqv
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/p1 -- python $ROOT/bench.py --cpu-bases 0 --steps 1 --warmup 1 "$@" > $OUT/p1.log 2>&1
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p1/**/*counter_collection.csv", recursive=True):
    disp = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fpl::", "")
        if n.startswith("k_"):
            disp[(r["Dispatch_Id"], n, r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, n, c), v in disp.items():
        acc[n][c].append(v)
for n in sorted(acc):
    print(n, {c: "%.4g" % (sum(v) / len(v)) for c, v in sorted(acc[n].items())})
PY
