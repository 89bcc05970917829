"""usage: PYTHONPATH=. python tools/prof_sections.py [reads] [median_len]
profiling only: run one batch on a -DFPL_PROF build and print the share of wave cycles per kernel section"""
import ctypes, sys
import numpy as np
import torch
from fastplong_amd import abi, engine, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
med = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
dev = torch.device("cuda:0")
seq_t, qual_t, off_t, max_len = synth.device_batch(n, seed=1, median_len=med, device=dev)
opt = abi.FplOptions.default(cut_front=1, cut_tail=1, cut_front_window=5, cut_tail_window=5, polyx=1, complexity_filter=1)
eng = engine.Engine(opt, synth.START_ADAPTER, synth.END_ADAPTER, device=0, max_cycles=max_len + 1)
res_t = torch.empty(n * 36, dtype=torch.uint8, device=dev)
import os
lib = engine.load_library(os.environ.get("FPL_PROF_LIB"))
eng.close()
eng = engine.Engine(opt, synth.START_ADAPTER, synth.END_ADAPTER, device=0, max_cycles=max_len + 1, lib=lib)
lib.fpl_debug_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
out = (ctypes.c_ulonglong * 64)()
for it in range(2):
    eng.process_device(seq_t, qual_t, off_t, max_len, res_t, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    lib.fpl_debug_prof(out, 64)
v = np.array(out[:], dtype=np.float64)
names = {"k_scan": (0, ["dequeue/meta/prefetch", "hist_zero", "body scan", "hist_totals(body)", "ends", "hist_totals(all)",
                        "median+acc", "lev confirm (rest)", "filter+frag stats", "result/plan/fbuf", "pre-lev", "lev_pair32_run"]),
         "k_trim_ends": (16, ["meta", "trim_and_cut", "polyX", "start adapter", "end adapter", "state write"]),
         "k_trim_ends_batched": (32, ["dequeue, offsets", "trimAndCut + polyX (lane = read)", "their wave-per-read fallback", "window scan, start",
                                      "candidate confirmation, start", "partial-pattern search, start", "partial confirmation, start",
                                      "window scan, end", "candidate confirmation, end", "partial-pattern search, end",
                                      "partial confirmation, end + state"])}
for k, (b, nm) in names.items():
    tot = v[b:b + 12].sum()
    if tot == 0:
        continue
    print(k, "total wave-cycles %.3g, per read %.0f" % (tot, tot / n))
    for i, s in enumerate(nm):
        print("   %-24s %5.1f%%  %8.0f cyc/read" % (s, 100 * v[b + i] / tot, v[b + i] / n))
