# the round's closing measurements on one GPU box: every -m gpu test, the smoke entry point, the profile summaries of the four
# bench workloads (profiles/collect_profile.sh), the default bench line as the driver runs it, one c5 line with its end-to-end leg
TAG=${1:-r03_final}
mkdir -p gpurun_out/$TAG
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/$TAG/gpu_tests.log 2>&1; grep -E "passed|failed" gpurun_out/$TAG/gpu_tests.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/$TAG/smoke.log 2>&1; tail -1 gpurun_out/$TAG/smoke.log
for wl in c3_full_pipeline c2_adapter_only c4_mixed c5_hifi64; do
  timeout 600 bash profiles/collect_profile.sh $TAG $wl > gpurun_out/$TAG/collect_$wl.log 2>&1; tail -1 gpurun_out/$TAG/collect_$wl.log
done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$TAG/bench_default.json 2> gpurun_out/$TAG/bench_default.err
timeout 400 python bench.py --workload c5_hifi64 --steps 10 --warmup 2 --cpu-bases 0 --e2e-reads 200000 --e2e-copies 0 > gpurun_out/$TAG/bench_c5_e2e.json 2> gpurun_out/$TAG/bench_c5_e2e.err
ls gpurun_out/$TAG
