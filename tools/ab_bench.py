"""usage: PYTHONPATH=. python tools/ab_bench.py [--reads N] [--workload W] lib1.so lib2.so ...
Measurement aid: per-kernel times of several builds of libfastplong_amd.so (tools/ab_build.sh) on the SAME resident
batch, same box, interleaved rounds -- boxes of the pool differ by a few per cent, so kernel variants are only comparable
side by side.  With the end trims ahead (--ahead / %AHEAD=1) the POSITION on the command line matters: every context brings two side
streams, the runtime deals its hardware queues out in turn, and the second context of a process runs ~3 % slower than the first whatever
its build (base.so twice: 11.6 / 11.9 ms) -- compare builds at the same position of separate runs, or swap them and average."""
import argparse, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # (as bench.py: the library's side streams want hardware queues of their own)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fastplong_amd import abi, engine

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=1_000_000)
ap.add_argument("--workload", default="c3_full_pipeline")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--median-len", type=int, default=0)
ap.add_argument("--ahead", action="store_true", help="fpl_assume_inputs_ready: the end trims of a step beside the step before")
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
dev = torch.device("cuda:0")
wl = bench.WORKLOADS[a.workload]
if a.median_len:
    wl = dict(wl, gen=dict(wl["gen"], median_len=a.median_len))
opt = abi.FplOptions.default(**wl["opt"])
seq_t, qual_t, off_t, max_len, s_ad, e_ad, fasta = bench.make_batch(wl, a.reads, 0, dev)
n = off_t.numel() - 1
nb = int(off_t[-1].item())
res_t = torch.empty(n * 36, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
engs = []
for spec in a.libs:  # lib.so or lib.so@FLAGS (FPL_DEBUG_FLAGS for a build with -DFPL_ABLATE: wrong results, timing only)
    # ... or lib.so%NAME=VALUE[,NAME=VALUE]: tuning variables the library reads when a context is created
    spec0, _, envs = spec.partition("%")
    p, _, flags = spec0.partition("@")
    L = engine.load_library(os.path.abspath(p))
    os.environ["FPL_DEBUG_FLAGS"] = flags or "0"
    kv = [x.split("=", 1) for x in envs.split(",") if x]
    for k, v in kv:
        os.environ[k] = v
    engs.append((spec, engine.Engine(opt, s_ad, e_ad, fasta, device=0, max_cycles=max_len, lib=L)))
    if a.ahead or "AHEAD=1" in envs:
        engs[-1][1].assume_inputs_ready(True)
    for k, _v in kv:
        os.environ.pop(k, None)
os.environ.pop("FPL_DEBUG_FLAGS", None)
ref_cnt = None
tot = {p: {} for p, _ in engs}
wall = {}
for r in range(a.rounds + 1):
    for p, e in engs:
        e.reset_counters()
        e.enable_timing(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps if r else 2):
            e.process_device(seq_t, qual_t, off_t, max_len, res_t, st)
        torch.cuda.synchronize()
        if r:
            wall.setdefault(p, []).append((time.perf_counter() - t0) * 1e3 / a.steps)
        kt, nbat = e.kernel_times()
        e.enable_timing(False)
        if r == 0:  # warm-up round: also check that every build computes the same counters
            c = e.counters()
            if ref_cnt is None:
                ref_cnt = c
            elif "@" not in p.split("%")[0] and not (c == ref_cnt).all():
                print("!! %s: counters differ from %s" % (p, engs[0][0]))
            continue
        for k, v in kt.items():
            tot[p].setdefault(k, []).append(v / nbat)
print("%d reads, %.2f Gbases, %s; ms per batch (mean of %d rounds x %d steps)" % (n, nb / 1e9, a.workload, a.rounds, a.steps))
for p, _ in engs:
    m = {k: sum(v) / len(v) for k, v in tot[p].items()}
    s = sum(m.values())
    w = sum(wall[p]) / len(wall[p])
    print("%-34s total %7.3f  %s  -> %.1f Gbases/s; wall clock %.3f ms per batch -> %.1f" % (os.path.basename(p), s, "  ".join("%s %.3f" % (k, v) for k, v in m.items()), nb / s / 1e6, w, nb / w / 1e6))
