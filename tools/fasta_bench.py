"""usage: PYTHONPATH=. python tools/fasta_bench.py [reads] [n_adapters]
BASELINE.json configs[4] style run (HiFi-like 20 kb reads, --adapter_fasta of N adapters) on one MI355X:
kernel times through the C-ABI, inputs resident in HBM.  A side measurement, not bench.py's metric."""
import sys
import numpy as np
import torch
from fastplong_amd import abi, engine, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
na = int(sys.argv[2]) if len(sys.argv) > 2 else 64
seq, qual, off, ads = synth.hifi_like(n, n_adapters=na)
dev = torch.device("cuda:0")
seq_t, qual_t = torch.from_numpy(seq).to(dev), torch.from_numpy(qual).to(dev)
off_t = torch.from_numpy(off.astype(np.int64)).to(dev)
max_len = int(np.diff(off.astype(np.int64)).max())
eng = engine.Engine(abi.FplOptions.default(), "", "", fasta=ads, device=0, max_cycles=max_len + 1)
res_t = torch.empty(n * 36, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
eng.process_device(seq_t, qual_t, off_t, max_len, res_t, st)
torch.cuda.synchronize()
eng.reset_counters()
eng.enable_timing(True)
for _ in range(3):
    eng.process_device(seq_t, qual_t, off_t, max_len, res_t, st)
torch.cuda.synchronize()
kt, nb = eng.kernel_times()
tot = sum(kt.values()) / nb
print("%d reads, %.2f Gbases, %d FASTA adapters: %.1f ms per batch -> %.1f Gbases/s; per kernel ms %s" % (
    n, int(off[-1]) / 1e9, na, tot, int(off[-1]) / tot / 1e6, {k: round(v / nb, 2) for k, v in kt.items()}))
