#!/bin/bash
# kernel timeline of a few steps of one workload: the kernel trace (start / end of every dispatch) as one csv under gpurun_out/<tag>/
# usage (through gpurun): bash tools/gpu/timeline.sh <tag> [<workload>]
set -u
TAG=${1:-timeline}; WL=${2:-c3_full_pipeline}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $ROOT/bench.py --workload $WL --cpu-bases 0 --e2e-reads 0 --parity-reads 0 --steps 6 --warmup 2 > "$OUT/bench_$WL.log" 2>&1
f=$(find /tmp/tl -name '*kernel_trace.csv' | head -1)
python - "$f" "$OUT/timeline_$WL.csv" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
with open(sys.argv[2], "w") as o:
    o.write("kernel,queue,stream,start_us,end_us\n")
    for r in rows:
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fpl::", "")[:60]
        o.write("%s,%s,%s,%.1f,%.1f\n" % (n.replace(",", ";"), r.get("Queue_Id", ""), r.get("Stream_Id", ""), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3))
P
tail -1 "$OUT/bench_$WL.log" | cut -c1-300
