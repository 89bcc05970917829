#!/bin/bash
# Collect the evidence bench.py's `roofline` object cites, on the GPU box, for ONE workload:
#   1. rocprofv3 --kernel-trace --stats of the bench command (per-kernel average durations)
#   2. separate --pmc passes of the same command: FETCH_SIZE, WRITE_SIZE (HBM bytes per launch) and the SQ instruction
#      counters (vector / scalar / LDS wave-instructions per launch)
# Usage (from the repo root, through gpurun):  bash profiles/collect_profile.sh <tag> [<workload> [bench args...]]
# Writes gpurun_out/<tag>/{kernel_stats_<wl>.csv,pmc_<wl>.json,bench_<wl>.json}; copy them to profiles/<tag>/ and merge
# pmc_<wl>.json into profiles/kernel_counters.json with `python profiles/summarize_profile.py --merge profiles/<tag>`.
set -u
TAG=${1:-prof}; shift || true
WL=${1:-c3_full_pipeline}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
W=$OUT/work_$WL
rm -rf "$W"; mkdir -p "$W"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --workload $WL --cpu-bases 0 --e2e-reads 0 --parity-reads 0 $*"
# (the bench line that is kept carries the parity sample; the profiled repeats leave it out)
python $ROOT/bench.py --workload $WL --cpu-bases 0 --e2e-reads 0 $* --steps 10 --warmup 2 > "$OUT/bench_$WL.json" 2> "$W/bench.err"
# 24 timed launches of every kernel behind 4 warm-up ones: kernel_stats_<wl>.csv is rocprofv3's own table over all 28,
# kernel_stats_timed_<wl>.csv the same figures over the LAST 24 dispatches of every kernel (from the kernel trace) -- what
# bench.py's HIP events bracket, so the two agree without a footnote about the warm-up launch
TIMED=24
rocprofv3 --kernel-trace --stats --output-format csv -d "$W/stats" -- $BENCH --steps $TIMED --warmup 4 > "$W/stats.log" 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --output-format csv -d "$W/pmc_$ctr" -- $BENCH --steps 2 --warmup 1 > "$W/pmc_$ctr.log" 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d "$W/pmc_SQ" -- $BENCH --steps 2 --warmup 1 > "$W/pmc_SQ.log" 2>&1
# LDS: cycles the pipe is busy and cycles lost to bank conflicts (a pass of their own: the SQ block has few counter slots)
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d "$W/pmc_LDS" -- $BENCH --steps 2 --warmup 1 > "$W/pmc_LDS.log" 2>&1
python "$ROOT/profiles/summarize_profile.py" "$OUT" "$WL" $TIMED
rm -rf "$W/stats" "$W"/pmc_*/ 2>/dev/null  # (the raw traces are large; the summaries stay)
