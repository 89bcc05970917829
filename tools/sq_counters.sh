#!/bin/bash
# profiling only: SQ instruction / activity counters per kernel for the default bench batch (two --pmc passes)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sq
rm -rf $OUT; mkdir -p $OUT   # (gpurun merges scratch output of earlier calls back: start clean)
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --cpu-bases 0 --e2e-reads 0 --steps 2 --warmup 1"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $OUT/p1 -- $B > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/p2 -- $B > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p3 -- $B > $OUT/p3.log 2>&1
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
