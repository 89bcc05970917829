"""usage: PYTHONPATH=. python tools/e2e_bench.py [reads]
End-to-end rates around the hot path on one MI355X (numbers for DESIGN.md section 7; never bench.py's `value`):
  1. fpl_process_batch from pageable host arrays (PCIe-inclusive C-ABI call),
  2. the CLI bin/fastplong_amd on a FASTQ file in /tmp (parse + H2D + kernels + D2H + format + write)."""
import os, subprocess, sys, time
import numpy as np
import torch
from fastplong_amd import abi, engine, synth, build

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
dev = torch.device("cuda:0")
seq_t, qual_t, off_t, max_len = synth.device_batch(n, seed=1, device=dev)
seq, qual, off = seq_t.cpu().numpy(), qual_t.cpu().numpy(), off_t.cpu().numpy().astype(np.uint64)
nb = int(off[-1])
opt = abi.FplOptions.default(cut_front=1, cut_tail=1, cut_front_window=5, cut_tail_window=5, polyx=1, complexity_filter=1)
eng = engine.Engine(opt, synth.START_ADAPTER, synth.END_ADAPTER, device=0, max_cycles=max_len + 1)
eng.process_host(seq, qual, off)  # warm-up (allocations)
t0 = time.perf_counter()
for _ in range(3):
    eng.process_host(seq, qual, off)
dt = (time.perf_counter() - t0) / 3
print("fpl_process_batch (pageable host arrays, %d reads, %.2f Gbases): %.3f s -> %.2f Gbases/s" % (n, nb / 1e9, dt, nb / dt / 1e9))
eng.close()

path = "/tmp/e2e.fq"
t0 = time.perf_counter()
with open(path, "wb") as f:
    sb, qb = seq.tobytes(), qual.tobytes()
    o = off.astype(np.int64)
    parts = []
    for i in range(n):
        parts.append(b"@r%d\n" % i)
        parts.append(sb[o[i]:o[i + 1]])
        parts.append(b"\n+\n")
        parts.append(qb[o[i]:o[i + 1]])
        parts.append(b"\n")
        if len(parts) > 50000:
            f.write(b"".join(parts)); parts = []
    f.write(b"".join(parts))
print("wrote %s (%.2f GB) in %.1f s" % (path, os.path.getsize(path) / 1e9, time.perf_counter() - t0))
cli = build.CLI
for outp in ("/dev/null", "/tmp/e2e_out.fq"):
    cmd = [cli, "-i", path, "-o", outp, "-s", synth.START_ADAPTER, "-e", synth.END_ADAPTER, "--cut_front", "--cut_tail",
           "-W", "5", "-x", "-y", "-j", "/tmp/e2e.json", "-h", "/tmp/e2e.html"] + sys.argv[2:]
    t0 = time.perf_counter()
    r = subprocess.run(cmd + ["-V"], capture_output=True, text=True, env=dict(os.environ, FPLH_TIMING="1"))
    dt = time.perf_counter() - t0
    print("CLI -> %s: rc=%d %.2f s -> %.2f Gbases/s end to end" % (outp, r.returncode, dt, nb / dt / 1e9))
    print("   " + "\n   ".join(l for l in r.stderr.splitlines() if "host pipeline" in l or "reader phases" in l or "reports:" in l or r.returncode))
