#!/bin/bash
# measurement aid: bin/fastplong_amd -V on N synthetic reads in tmpfs, no profiler: the host pipeline's own account of a run
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-600000}; shift
FQ=/dev/shm/fpl_plain_$$.fq
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
for i in 1 2 3; do
  t0=$(date +%s.%N)
  FPLH_T0=$t0 $ROOT/bin/fastplong_amd -i $FQ -o /dev/null -j /tmp/pl.json -h /tmp/pl.html --cut_front --cut_tail -x -y -V \
      -s AAGGATTCATTCCCACGGTAACAC -e GTGTTACCGTGGGAATGAATCCTT "$@" 2>&1 | grep -E "^host pipeline|^device thread|^start-up|^since launch|^reports|page-locked alloc" | cut -c1-260
  echo "whole process $(echo "$(date +%s.%N) - $t0" | bc -l 2>/dev/null || python3 -c "import time; print(time.time() - $t0)") s"
done
rm -f $FQ
