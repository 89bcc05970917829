"""usage: PYTHONPATH=. python tools/block_timeline.py lib.so [--reads N]
Profiling only (library built with -DFPL_PROF_BLOCKS): start / end time and XCD of every block of k_stats_sorted for one
bench batch -> resident blocks over time, per-XCD finish times, the longest blocks."""
import argparse, ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fastplong_amd import abi, engine

ap = argparse.ArgumentParser()
ap.add_argument("lib")
ap.add_argument("--reads", type=int, default=1_000_000)
a = ap.parse_args()
dev = torch.device("cuda:0")
wl = bench.WORKLOADS["c3_full_pipeline"]
opt = abi.FplOptions.default(**wl["opt"])
seq_t, qual_t, off_t, max_len, s_ad, e_ad, fasta = bench.make_batch(wl, a.reads, 0, dev)
n = off_t.numel() - 1
res_t = torch.empty(n * 36, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
L = engine.load_library(os.path.abspath(a.lib))
e = engine.Engine(opt, s_ad, e_ad, fasta, device=0, max_cycles=max_len, lib=L)
for _ in range(3):
    e.process_device(seq_t, qual_t, off_t, max_len, res_t, st)
torch.cuda.synchronize()
buf = np.zeros(((1 << 17), 2), np.uint64)
raw = ctypes.CDLL(os.path.abspath(a.lib))
assert raw.fpl_debug_read_blockprof(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)) == 0
t0 = buf[:, 0].astype(np.int64)
t1 = (buf[:, 1] >> np.uint64(8)).astype(np.int64)
xcd = (buf[:, 1] & np.uint64(15)).astype(np.int64)
ok = t0 > 0
t0, t1, xcd = t0[ok], t1[ok], xcd[ok]
ids = np.nonzero(ok)[0]
base = t0.min()
us = lambda t: (t - base) / 100.0  # 100 MHz clock
dur = us(t1) - us(t0)
print("blocks recorded", len(t0), "span %.1f us" % us(t1).max())
heavy = dur > 50
print("blocks > 50 us:", int(heavy.sum()), "sum of their time %.1f ms, / 512 slots = %.2f ms" % (dur[heavy].sum() / 1e3, dur[heavy].sum() / 512e3))
edges = np.linspace(0, us(t1).max(), 41)
for lo, hi in zip(edges[:-1], edges[1:]):
    mid = (lo + hi) / 2
    res = ((us(t0) <= mid) & (us(t1) > mid))
    print("t=%7.0f us resident %4d  per XCD %s" % (mid, res.sum(), np.bincount(xcd[res], minlength=8).tolist()))
for x in range(8):
    m = xcd == x
    print("XCD", x, "blocks", int(m.sum()), "busy sum %.1f ms" % (dur[m].sum() / 1e3), "last end %.0f us" % us(t1[m]).max())
top = np.argsort(-dur)[:12]
gx = None
print("longest:", [(int(ids[i]), round(float(dur[i]), 1), round(float(us(t0[i])), 0)) for i in top])
