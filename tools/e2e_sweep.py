"""usage: PYTHONPATH=. python tools/e2e_sweep.py [reads]
Measurement aid (numbers for DESIGN.md; never bench.py's `value`): the CLI end to end on a FASTQ file in /dev/shm
for several reader configurations (--reader_threads, --chunk_mb, page-locked batches on / off)."""
import ctypes as C, os, subprocess, sys, time
import numpy as np
import torch
from fastplong_amd import build, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
dev = torch.device("cuda:0")
seq_t, qual_t, off_t, max_len = synth.device_batch(n, seed=1, device=dev)
seq, qual, off = seq_t.cpu().numpy(), qual_t.cpu().numpy(), off_t.cpu().numpy().astype(np.uint64)
nb = int(off[-1])
del seq_t, qual_t
torch.cuda.empty_cache()
host = C.CDLL(build.HOST_LIB)
host.fplh_write_fastq.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_char_p, C.c_int]
fq = "/dev/shm/e2e_sweep.fq"
assert host.fplh_write_fastq(fq.encode(), seq.ctypes.data, qual.ctypes.data, off.ctypes.data, n, b"r", 16) == 0
print("input %.2f GB, %.2f Gbases" % (os.path.getsize(fq) / 1e9, nb / 1e9), flush=True)
base = [build.CLI, "-i", fq, "-o", "/dev/null", "-s", synth.START_ADAPTER, "-e", synth.END_ADAPTER, "--cut_front", "--cut_tail",
        "-W", "5", "-x", "-y", "-j", "/dev/shm/e2e.json", "-h", "/dev/shm/e2e.html", "-V"]
configs = [
    ("device parse 32MB", [], {}),
    ("device parse 64MB", ["--chunk_mb", "64"], {}),
    ("device parse 128MB", ["--chunk_mb", "128"], {}),
    ("device parse 256MB", ["--chunk_mb", "256"], {}),
    ("host parse 32MB", ["--host_parse"], {}),
    ("host parse 64MB", ["--host_parse", "--chunk_mb", "64"], {}),
    ("host parse 128MB", ["--host_parse", "--chunk_mb", "128"], {}),
    ("device parse 64MB R=12", ["--chunk_mb", "64", "--reader_threads", "12"], {}),
]
for name, extra, env in configs:
    e = dict(os.environ, FPLH_TIMING="1", FPLH_T0=repr(time.time()), **env)
    t0 = time.perf_counter()
    r = subprocess.run(base + extra, capture_output=True, text=True, env=e)
    dt = time.perf_counter() - t0
    print("%-34s rc=%d process %.2f s -> %.2f Gbases/s" % (name, r.returncode, dt, nb / dt / 1e9))
    for l in r.stderr.splitlines():
        if any(k in l for k in ("host pipeline", "device thread")) or r.returncode:
            print("     " + l)
    sys.stdout.flush()
os.remove(fq)
if os.path.exists("/dev/shm/e2e_out.fq"):
    os.remove("/dev/shm/e2e_out.fq")
