#!/bin/bash
# measurement aid: the kernels of the bench's timed steps in time order (rocprofv3 kernel trace), with the idle time in front of each --
# where a step's 11.5 ms go that no kernel accounts for
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/step_gaps; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python $ROOT/bench.py --cpu-bases 0 --e2e-reads 0 --parity-reads 0 --steps 6 --warmup 3 "$@" > $OUT/log 2>&1
python - $OUT <<'PY'
import csv, glob, sys
ev = []
for f in glob.glob(sys.argv[1] + "/tr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fpl::", "")[:44], r.get("Queue_Id", "")))
ev.sort()
scans = [i for i, e in enumerate(ev) if e[2].startswith("k_scan")]
# the last three steps: from the start of the third-last k_scan to the end
i0 = scans[-8]
t0 = ev[i0][0]
busy_end = ev[i0][0]
print("   start us   dur us  idle-before us  queue  kernel   (idle-before: nothing at all was running)")
for s, e, name, q in ev[i0:scans[-4]]:
    idle = max(0, s - busy_end)
    print("%10.1f %8.1f %10.1f  %5s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, idle / 1e3, q, name))
    busy_end = max(busy_end, e)
PY
rm -rf $OUT/tr
