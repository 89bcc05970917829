#!/bin/bash
# measurement aid: rocprofv3 kernel stats of one bin/fastplong_amd --device_parse run (the k_text_* kernels of csrc/text_parse.h)
#   tools/text_kernels_prof.sh [reads] [chunk_mb]   -> gpurun_out/text_kernels.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-300000}; CH=${2:-32}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
FQ=/dev/shm/fpl_textprof_$$.fq
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
print("bases", int(off[-1]))
PY
BYTES=$(stat -c %s $FQ)
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/textprof
FPLH_NORMAL_EXIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/textprof -- $ROOT/bin/fastplong_amd -i $FQ -o /dev/null -j /tmp/tp.json -h /tmp/tp.html \
    --device_parse --chunk_mb $CH --cut_front --cut_tail -x -y -V > $OUT/textprof.log 2>&1
python - "$OUT" "$BYTES" "$CH" <<'PY' | tee $OUT/text_kernels.txt
import csv, glob, sys
out, nbytes, ch = sys.argv[1], int(sys.argv[2]), sys.argv[3]
rows = []
for f in glob.glob(out + "/textprof/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
print("bin/fastplong_amd --device_parse --chunk_mb %s on %.2f GB of FASTQ text; rocprofv3 --kernel-trace --stats, per launch (one launch per chunk)" % (ch, nbytes / 1e9))
tot = 0.0
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    n = r["Name"]
    if "k_text" in n:
        import re
        short = re.search(r"(k_text_\w+)", n).group(1)
        tot += float(r["TotalDurationNs"])
        print("%-18s calls %5s  avg %8.1f us  min %8.1f  max %8.1f  total %8.2f ms" % (short, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
print("all k_text_* kernels: %.2f ms for %.2f GB of text = %.0f GB/s of text through the parse (4 B of HBM traffic per byte: %.0f GB/s)" % (tot / 1e6, nbytes / 1e9, nbytes / tot, 4 * nbytes / tot))
PY
grep -E "host pipeline|device parse" $OUT/textprof.log | cut -c1-300 | tee -a $OUT/text_kernels.txt
rm -f $FQ; rm -rf $OUT/textprof
