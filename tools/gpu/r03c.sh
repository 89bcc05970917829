set -x
OUT=$PWD/gpurun_out/r03c; mkdir -p $OUT
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --cpu-bases 0 --e2e-reads 0 --parity-reads 0 --steps 3 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_c3.csv; rm -rf $OUT/stats
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -- $B > $OUT/p1.log 2>&1
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p1/**/*counter_collection.csv", recursive=True):
    disp = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fpl::", "")
        if not n.startswith("k_"):
            continue
        disp[(r["Dispatch_Id"], n, r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, n, c), v in disp.items():
        acc[n][c].append(v)
with open("$OUT/sq_c3.txt", "w") as o:
    for n in sorted(acc):
        o.write("%s %s\n" % (n, {c: "%.4g" % (sum(v) / len(v)) for c, v in sorted(acc[n].items())}))
PY
rm -rf $OUT/p1
cd $ROOT
timeout 300 bash tools/prof_sections.sh > $OUT/prof_sections.txt 2>&1
