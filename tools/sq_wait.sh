# measurement aid: what the waves of the three big kernels do with their cycles (SQ wave / wait / active counters, three --pmc passes of the c3 bench command; quad-cycles summed over the waves, millions per launch) -> gpurun_out/sqwait/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --workload c3_full_pipeline --cpu-bases 0 --e2e-reads 0 --parity-reads 0 --steps 2 --warmup 1"
mkdir -p $R/gpurun_out/sqwait
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM SQ_WAIT_INST_VMEM"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_$n -- $B > /tmp/pmc_$n.log 2>&1
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import sys, csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
try:
    for r in csv.DictReader(open(sys.argv[1])):
        k = r["Kernel_Name"].split("(")[0].replace("void fpl::", "")
        if not any(x in k for x in ("k_scan", "k_stats_sorted", "k_trim_ends_batched")): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        print(k, {c: round(v / cnt[(k, c)] / 1e6, 1) for c, v in d.items()}, "(M per launch)")
except Exception as e:
    print("no data", e)
PY
done 2>&1 | tee $R/gpurun_out/sqwait/summary.txt
