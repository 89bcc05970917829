#!/bin/bash
# measurement aid: does the NUMA node of the CLI's threads (and with them of its page-locked arena, first touched by them) matter for the
# upload rate?  The same run pinned to the CPUs of each node in turn (taskset), and unpinned.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-600000}
FQ=/dev/shm/fpl_numa_$$.fq
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
run() { echo "== $1"; shift; "$@" $ROOT/bin/fastplong_amd -i $FQ -o /dev/null -j /tmp/n.json -h /tmp/n.html --cut_front --cut_tail -x -y -V 2>&1 | grep -E "host pipeline|device thread|text submissions" | tail -4 | cut -c1-260; }
echo "GPU numa node: $(cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' ')"
lscpu | grep -E "NUMA node"
run "unpinned" env
run "host parse" env FPLH_HOST_PARSE=1
for node in; do
  cpus=$(cat /sys/devices/system/node/node$node/cpulist)
  run "taskset node $node ($cpus)" taskset -c $cpus
done
rm -f $FQ
