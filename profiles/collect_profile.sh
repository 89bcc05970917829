#!/bin/bash
# Collect the evidence bench.py's `roofline` object cites, on the GPU box:
#   1. rocprofv3 --kernel-trace --stats of the default bench command (per-kernel average durations)
#   2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same command -> HBM bytes per launch
# Usage (from the repo root, through gpurun):  bash profiles/collect_profile.sh r01_v4 [bench args...]
# Writes gpurun_out/<tag>/{kernel_stats_1M.csv,pmc_hbm_1M.json,bench_1M.json}; copy them to profiles/<tag>/.
set -u
TAG=${1:-prof}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --cpu-bases 0 --e2e-reads 0 $*"
$BENCH --steps 10 --warmup 2 > "$OUT/bench_1M.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH --steps 5 --warmup 1 > "$OUT/stats.log" 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --output-format csv -d "$OUT/pmc_$ctr" -- $BENCH --steps 2 --warmup 1 > "$OUT/pmc_$ctr.log" 2>&1
done
python "$ROOT/profiles/summarize_profile.py" "$OUT"
