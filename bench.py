#!/usr/bin/env python
"""bench.py -- Gbases/s of the per-read hot path (trim + cut + filter + stats) on MI355X.

One "step" = one pass of the whole hot path (k_trim_ends, k_cycle_stats pre, k_scan,
k_cycle_stats post) over one resident batch of synthetic ONT-like reads.  Inputs are in HBM when
the timed region starts.  One process per GPU; for N > 1 launch with torch.distributed.run: every
rank owns its own shard of reads (weak scaling, no data-path collective) and the additive counter
buffer is all-reduced over RCCL once at the end of the timed region.

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` (dominant kernel,
HIP-event timed inside the timed region) and `cpu_baseline` (the oracle on the host cores, on a
bounded sample of the same workload, outside the timed region).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
ALGO_BYTES_PER_BASE = 2.0  # one seq byte + one quality byte, each read once

WORKLOADS = {
    # BASELINE.json configs[2]: the metric's "trim+cut+filter" pipeline on the configs[1] reads
    "c3_full_pipeline": dict(opt=dict(cut_front=1, cut_tail=1, cut_front_window=5, cut_tail_window=5, polyx=1,
                                      complexity_filter=1),
                             flags="-s/-e fixed, --cut_front --cut_tail -W 5, -x, -y"),
    # BASELINE.json configs[1]: adapter trim only
    "c2_adapter_only": dict(opt=dict(), flags="-s/-e fixed"),
}


def cpu_baseline(opt, seq_t, qual_t, off_t, target_bases, max_threads=16):
    """Time the oracle (C restatement of the reference path, kind="port") on the host cores over
    the first reads of the same batch.  Outside the timed region; the oracle is only the thing
    measured here, never part of the GPU path."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from fastplong_amd import abi, synth
    from oracle import oracle

    off = off_t.cpu().numpy().astype(np.int64)
    n = int(np.searchsorted(off, target_bases))
    n = max(16, min(n, len(off) - 1))
    nb = int(off[n])
    seq = seq_t[:nb].cpu().numpy()
    qual = qual_t[:nb].cpu().numpy()
    threads = max(1, min(os.cpu_count() or 1, max_threads))
    cfg = oracle.Config(opt, synth.START_ADAPTER, synth.END_ADAPTER)
    oracle.lib()
    C = int(np.diff(off[:n + 1]).max())
    # contiguous shards, one per thread (the reference round-robins packs of 16 reads over <= 16 workers)
    cuts = [int(i * n / threads) for i in range(threads + 1)]

    def work(t):
        a, b = cuts[t], cuts[t + 1]
        if b <= a:
            return
        o = (off[a:b + 1] - off[a]).astype(np.uint64)
        oracle.process_batch(cfg, seq[off[a]:off[b]], qual[off[a]:off[b]], o, max_cycles=C)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(work, range(threads)))
    dt = time.perf_counter() - t0
    return {"value": nb / dt / 1e9, "unit": "Gbases/s", "cores": threads, "kind": "port",
            "sample": "first %d reads (%d bases) of the same batch, oracle/liboracle.so, %d threads, %.1f s" % (
                n, nb, threads, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU (1 M x ~10 kb N50)")
    ap.add_argument("--median-len", type=int, default=8000)
    ap.add_argument("--workload", default="c3_full_pipeline", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-bases", type=float, default=6e9, help="size of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--set", default="", help="ablation only: comma separated fpl_options overrides, e.g. adapter_enabled=0")
    ap.add_argument("--hbm-traffic", type=float, default=None,
                    help="HBM bytes per launch of the dominant kernel from a separate rocprofv3 --pmc run")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from fastplong_amd import abi, dist as fdist, engine, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d" % (
                args.gpus, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    wl = WORKLOADS[args.workload]
    okw = dict(wl["opt"])
    for kv in filter(None, args.set.split(",")):
        k, val = kv.split("=")
        okw[k] = float(val) if k == "ed_max" else int(val)
    opt = abi.FplOptions.default(**okw)
    # synthetic shard of this rank (weak scaling: every rank gets --reads reads of its own)
    seq_t, qual_t, off_t, max_len = synth.device_batch(args.reads, seed=1 + rank, median_len=args.median_len,
                                                       device=dev)
    n = off_t.numel() - 1
    n_bases = int(off_t[-1].item())
    # all ranks agree on the per-cycle capacity so that the counter buffers line up for the all-reduce
    C = fdist.agree_capacity(max_len, device=dev)
    eng = engine.Engine(opt, synth.START_ADAPTER, synth.END_ADAPTER, device=local_rank, max_cycles=C)
    res_t = torch.empty(n * 36, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)

    def step():
        eng.process_device(seq_t, qual_t, off_t, max_len, res_t, stream.cuda_stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    eng.reset_counters()
    eng.enable_timing(True)  # HIP events around every kernel, on the launch stream, inside the timed region
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        # the only collective of the path: sum the Stats / FilterResult counters over RCCL
        fdist.allreduce_counters(eng.counters_tensor())
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    tb = torch.tensor([n_bases], device=dev, dtype=torch.int64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
    dt = float(tt.item())
    total_bases = int(tb.item())
    ktimes, nbatches = eng.kernel_times()
    eng.enable_timing(False)

    if rank == 0:
        counters = eng.counters()
        v = abi.CountersView(counters, C, 2)
        dom = max(ktimes, key=ktimes.get)
        dom_ms = ktimes[dom] / max(1, nbatches)
        achieved = ALGO_BYTES_PER_BASE * n_bases / (dom_ms * 1e-3) / 1e9
        # HBM bytes per launch of the dominant kernel: PMC counters come from separate rocprofv3 --pmc
        # passes of this same command (profiles/hbm_traffic.json records bytes per base, corrected as
        # MI355X_MICROARCH.md prescribes); --hbm-traffic overrides, otherwise null when nothing is recorded
        traffic, traffic_src = args.hbm_traffic, "--hbm-traffic"
        if traffic is None:
            try:
                rec = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
                if rec.get("workload") == args.workload and not args.set and dom in rec["hbm_bytes_per_base"]:
                    traffic, traffic_src = rec["hbm_bytes_per_base"][dom] * n_bases, rec["source"]
            except (OSError, ValueError, KeyError):
                traffic = None
        out = {
            "metric": "Gbases/s processed (trim+cut+filter)",
            "value": total_bases * args.steps / dt / 1e9,
            "unit": "Gbases/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": "%s%s: BASELINE.json configs[%d] -- %d synthetic ONT-like reads per GPU, lognormal lengths "
                            "(median %d, sigma 0.5, N50 ~10 kb), Q~N(18,8), %s; inputs resident in HBM" % (
                                args.workload, (" [ABLATION " + args.set + "]") if args.set else "",
                                2 if args.workload.startswith("c3") else 1, n, args.median_len,
                                wl["flags"]),
                "reads_per_gpu": n, "bases_per_gpu": n_bases, "max_read_len": max_len,
                "parallelism": "shard%d (independent read shards, one RCCL all-reduce of the counters)" % world,
            },
            "roofline": {
                "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src if traffic is not None else None,
                "algorithmic_bytes_per_launch": ALGO_BYTES_PER_BASE * n_bases,
                "kernel_ms": {k: ktimes[k] / max(1, nbatches) for k in ktimes},
            },
            "counters_check": {"reads_in": int(v.pre.reads), "bases_in": int(v.pre.length_sum),
                               "fragments_out": int(v.post.reads), "bases_out": int(v.post.length_sum)},
        }
        if args.cpu_bases > 0 and world == 1:  # (rank 0 at N = 1 only: the other ranks would sit in the barrier meanwhile)
            out["cpu_baseline"] = cpu_baseline(opt, seq_t, qual_t, off_t, args.cpu_bases)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
