#!/bin/bash
# measurement aid: where the device-side time of a bin/fastplong_amd run goes -- rocprofv3 kernel + memory-copy traces of one run,
# summed per kind (upload busy, download busy, kernels busy, wall between first and last event)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-300000}; shift
OUT=$ROOT/gpurun_out; mkdir -p $OUT
FQ=/dev/shm/fpl_tl_$$.fq
cd $ROOT && python - "$FQ" "$N" <<'PY'
import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
from fastplong_amd import synth, build
seq, qual, off = synth.ont_like(int(sys.argv[2]), seed=3, median_len=8000)
off = off.astype(np.uint64)
host = C.CDLL(build.HOST_LIB)
host.fplh_write_fastq_ex.restype = C.c_int
host.fplh_write_fastq_ex.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_char_p, C.c_int, C.c_int]
assert host.fplh_write_fastq_ex(sys.argv[1].encode(), seq.ctypes.data, qual.ctypes.data, off.ctypes.data, len(off) - 1, b"r", 16, 0) == 0
PY
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/tl
FPLH_NORMAL_EXIT=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/tl -- $ROOT/bin/fastplong_amd -i $FQ -o /dev/null -j /tmp/tl.json -h /tmp/tl.html \
    --cut_front --cut_tail -x -y -V "$@" > $OUT/tl.log 2>&1
grep -E "host pipeline" $OUT/tl.log | cut -c1-220
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
ev = []
for f in glob.glob(out + "/tl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append(("kernel", r["Kernel_Name"].split("(")[0][-40:], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "")))
for f in glob.glob(out + "/tl/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append(("copy", r["Direction"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), ""))
ev.sort(key=lambda e: e[2])
t0, t1 = ev[0][2], max(e[3] for e in ev)
def busy(sel):
    iv = sorted((e[2], e[3]) for e in ev if sel(e))
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None: tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None: tot += cur_e - cur_s
    return tot, len(iv)
print("wall of the device's events: %.1f ms" % ((t1 - t0) / 1e6))
for name, sel in (("uploads (H2D)", lambda e: e[0] == "copy" and "HOST_TO_DEVICE" in e[1].upper().replace(" ", "_")),
                  ("downloads (D2H)", lambda e: e[0] == "copy" and "DEVICE_TO_HOST" in e[1].upper().replace(" ", "_")),
                  ("kernels (any)", lambda e: e[0] == "kernel"),
                  ("k_text_*", lambda e: e[0] == "kernel" and "k_text" in e[1])):
    b, n = busy(sel)
    print("%-18s busy %8.1f ms in %6d events (%.0f %% of the wall)" % (name, b / 1e6, n, 100.0 * b / (t1 - t0)))
mid = len(ev) // 2
base = ev[mid][2]
print("events around the middle of the run (us from the first of them: start, duration, what):")
for e in ev[mid:mid + 70]:
    print("  %9.1f %8.1f  %s %s" % ((e[2] - base) / 1e3, (e[3] - e[2]) / 1e3, e[0], e[1][-36:]))
dirs = collections.Counter(e[1] for e in ev if e[0] == "copy")
print("copy directions:", dict(dirs))
h2d = [e for e in ev if e[0] == "copy" and "HOST_TO_DEVICE" in e[1].upper().replace(" ", "_") and e[3] - e[2] > 100000]
if h2d:
    d = sorted(e[3] - e[2] for e in h2d)
    print("large uploads: %d, median %.0f us, p90 %.0f us" % (len(d), d[len(d) // 2] / 1e3, d[int(len(d) * 0.9)] / 1e3))
    gaps = sorted(h2d[i + 1][2] - h2d[i][3] for i in range(len(h2d) - 1))
    print("gaps between consecutive large uploads: median %.0f us, p90 %.0f us, sum %.1f ms" % (gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3, sum(g for g in gaps if g > 0) / 1e6))
    # the steady part of the run: from the 20th large upload to the 20th from the end
    if len(h2d) > 60:
        # the pipeline's uploads: everything behind the longest pause (adapter detection and the contexts come before it)
        gl = [h2d[i + 1][2] - h2d[i][3] for i in range(len(h2d) - 1)]
        st = h2d[gl.index(max(gl)) + 1:]
        span = st[-1][3] - st[0][2]
        g2 = [st[i + 1][2] - st[i][3] for i in range(len(st) - 1)]
        print("steady part: %d uploads in %.1f ms: %.0f us per upload, busy %.0f %%; gaps: <50 us %d, 50-150 %d, 150-400 %d, 400-1000 %d, >1000 %d; sum of gaps %.1f ms" % (
            len(st), span / 1e6, span / 1e3 / len(st), 100.0 * sum(e[3] - e[2] for e in st) / span,
            sum(g < 50e3 for g in g2), sum(50e3 <= g < 150e3 for g in g2), sum(150e3 <= g < 400e3 for g in g2), sum(400e3 <= g < 1e6 for g in g2), sum(g >= 1e6 for g in g2), sum(g2) / 1e6))
        kern = sorted((e[2], e[3]) for e in ev if e[0] == "kernel")
        big = sorted(range(len(g2)), key=lambda i: -g2[i])[:12]
        print("the largest gaps (us into the steady part, gap us, kernels busy inside the gap us, what ran last before it ended):")
        for i in sorted(big):
            a, b = st[i][3], st[i + 1][2]
            kb = sum(max(0, min(e, b) - max(s_, a)) for s_, e in kern if e > a and s_ < b)
            last = [e for e in ev if e[3] <= b and e[3] > a]
            print("  %9.0f %7.0f %7.0f  %s" % ((a - st[0][2]) / 1e3, (b - a) / 1e3, kb / 1e3, (last[-1][0] + " " + last[-1][1][-30:]) if last else "-"))
PY
rm -f $FQ; rm -rf $OUT/tl
