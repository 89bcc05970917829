"""usage: PYTHONPATH=. python tools/overlap_probe.py [--reads N] [--workload W] lib.so ...
Measurement aid: does the statistics pass of one batch run beside the trim + scan of the next?  Per library: K steps of ONE
context on one stream (what bench.py times), then K steps dealt in turn to TWO contexts of the same library on two streams
(every batch whole and independent -- the contexts share nothing but the device), wall clock over the device."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fastplong_amd import abi, engine

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=1_000_000)
ap.add_argument("--workload", default="c3_full_pipeline")
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--median-len", type=int, default=0)
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
res = [torch.empty(n * 36, dtype=torch.uint8, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(dev) for _ in range(2)]
for spec in a.libs:
    spec0, _, envs = spec.partition("%")
    L = engine.load_library(os.path.abspath(spec0))
    kv = [x.split("=", 1) for x in envs.split(",") if x]
    for k, v in kv:
        os.environ[k] = v
    engs = [engine.Engine(opt, s_ad, e_ad, fasta, device=0, max_cycles=max_len, lib=L) for _ in range(2)]
    for k, _v in kv:
        os.environ.pop(k, None)
    out = []
    for mode in (1, 2):
        for warm in (True, False):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(4 if warm else a.steps):
                j = i % mode
                engs[j].process_device(seq_t, qual_t, off_t, max_len, res[j], streams[j].cuda_stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        out.append(dt / a.steps * 1e3)
    print("%-40s one context %.3f ms/step (%.1f Gbases/s)   two contexts, two streams %.3f ms/step (%.1f Gbases/s)" % (
        os.path.basename(spec), out[0], nb / out[0] / 1e6, out[1], nb / out[1] / 1e6))
    # the complementary pair: trim + scan + resolve of one batch beside the statistics pass of another (FPL_DEBUG_FLAGS
    # 0x1000 / 0x2000, csrc/pipeline.h: wrong counters, timing only)
    halves = []
    for flag in ("4096", "8192"):
        os.environ["FPL_DEBUG_FLAGS"] = flag
        for k, v in kv:
            os.environ[k] = v
        halves.append(engine.Engine(opt, s_ad, e_ad, fasta, device=0, max_cycles=max_len, lib=L))
        for k, _v in kv:
            os.environ.pop(k, None)
    os.environ.pop("FPL_DEBUG_FLAGS", None)
    for j in range(2):
        halves[j].process_device(seq_t, qual_t, off_t, max_len, res[j], streams[j].cuda_stream)  # (the back-only context's first batch runs whole)
    t = []
    for who in ((0,), (1,), (0, 1)):
        for warm in (True, False):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(3 if warm else a.steps):
                for j in who:
                    halves[j].process_device(seq_t, qual_t, off_t, max_len, res[j], streams[j].cuda_stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        t.append(dt / a.steps * 1e3)
    print("%-40s front alone %.3f  back alone %.3f  (sum %.3f)   front beside back %.3f ms per pair -> %.1f Gbases/s if a pipeline did that" % (
        "", t[0], t[1], t[0] + t[1], t[2], nb / t[2] / 1e6))
    # the end trims of one batch (latency of per-lane loads) beside the scan of another (vector issue): 0x5000 = trims only, 0x9000 = scan only
    halves = []
    for flag in (str(0x5000), str(0x9000)):
        os.environ["FPL_DEBUG_FLAGS"] = flag
        for k, v in kv:
            os.environ[k] = v
        halves.append(engine.Engine(opt, s_ad, e_ad, fasta, device=0, max_cycles=max_len, lib=L))
        for k, _v in kv:
            os.environ.pop(k, None)
    os.environ.pop("FPL_DEBUG_FLAGS", None)
    for j in range(2):
        halves[j].process_device(seq_t, qual_t, off_t, max_len, res[j], streams[j].cuda_stream)
    t = []
    for who in ((0,), (1,), (0, 1), (1, 0)):
        for warm in (True, False):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(3 if warm else a.steps):
                for j in who:
                    halves[j].process_device(seq_t, qual_t, off_t, max_len, res[j], streams[j].cuda_stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        t.append(dt / a.steps * 1e3)
    print("%-40s trims alone %.3f  scan alone %.3f  (sum %.3f)   trims beside scan %.3f (trims launched first) / %.3f (scan first) ms per pair" % (
        "", t[0], t[1], t[0] + t[1], t[2], t[3]))
