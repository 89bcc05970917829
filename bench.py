#!/usr/bin/env python
"""bench.py -- Gbases/s of the per-read hot path (trim + cut + filter + stats) on MI355X.

One "step" = one pass of the whole hot path (csrc/pipeline.h: k_trim_ends, k_scan, k_resolve + k_redo, the bucket
kernels, k_stats_sorted + its reduce, the post-only pass) over one resident batch of synthetic reads.  Inputs are in
HBM when the timed region starts.  One process per GPU; for N > 1 launch with torch.distributed.run: every
rank owns its own shard of reads (weak scaling, no data-path collective) and the additive counter
buffer is all-reduced over RCCL once at the end of the timed region.

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` (dominant kernel,
HIP-event timed inside the timed region), `cpu_baseline` (the oracle on the host cores, on a
bounded sample of the same workload, outside the timed region), `parity_sample` (the records the
timed step left behind and the counters of the same reads against that oracle run) and `e2e`
(bin/fastplong_amd on the same reads as FASTQ text: /dev/null, one file, 16 --split files).
"""
import argparse
import json
import os
import sys
import time

# the library runs a batch on up to three streams side by side (the caller's, the post-only statistics pass, the next batch's end
# trims): with the runtime's default of four hardware queues per device and torch's own streams, two of them can land on one queue
# and then run in submission order.  Read when the HIP runtime starts, i.e. before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# why cpu_baseline.kind is "port" and not "reference": the reference's own per-read path does not compile in this image
REFERENCE_BLOCKED_BY = ["src/adaptertrimmer.cpp:4 #include <hwy/highway.h> (Highway absent)",
                        "src/fastqreader.h:35 #include <igzip_lib.h> (ISA-L absent)"]
LINE_LIMIT = 4096  # bytes of the stdout line: the driver keeps the last 8 KB of stdout and parses the last line

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
ALGO_BYTES_PER_BASE = 2.0  # one seq byte + one quality byte, each read once

WORKLOADS = {
    # BASELINE.json configs[2]: the metric's "trim+cut+filter" pipeline on the configs[1] reads
    "c3_full_pipeline": dict(opt=dict(cut_front=1, cut_tail=1, cut_front_window=5, cut_tail_window=5, polyx=1,
                                      complexity_filter=1),
                             flags="-s/-e fixed, --cut_front --cut_tail -W 5, -x, -y", config=2, reads=1_000_000,
                             gen=dict(kind="ont", median_len=8000, sigma_len=0.5),
                             reads_desc="lognormal lengths (median 8000, sigma 0.5, N50 ~10 kb), Q~N(18,8)"),
    # BASELINE.json configs[1]: adapter trim only
    "c2_adapter_only": dict(opt=dict(), flags="-s/-e fixed", config=1, reads=1_000_000,
                            gen=dict(kind="ont", median_len=8000, sigma_len=0.5),
                            reads_desc="lognormal lengths (median 8000, sigma 0.5, N50 ~10 kb), Q~N(18,8)"),
    # BASELINE.json configs[3]: the mixed-length shard of one GPU (20 M reads over 8 GPUs = 2.5 M per GPU), full pipeline.
    # lognormal sigma 0.9 clipped to [300, 200 000] (the generator needs 300 bases for its decorations; the
    # configuration says 200), median 6673 = 15 kb * exp(-0.81): N50 ~15 kb
    "c4_mixed": dict(opt=dict(cut_front=1, cut_tail=1, cut_front_window=5, cut_tail_window=5, polyx=1,
                              complexity_filter=1),
                     flags="-s/-e fixed, --cut_front --cut_tail -W 5, -x, -y", config=3, reads=2_500_000,
                     gen=dict(kind="ont", median_len=6673, sigma_len=0.9, min_len=200, max_len=200_000, seed0=4),
                     reads_desc="lognormal lengths (sigma 0.9, clipped to [300, 200000], N50 ~15 kb), Q~N(18,8)"),
    # BASELINE.json configs[4]: HiFi-like reads, --adapter_fasta of 64 adapters visited in header order
    # (src/adaptertrimmer.cpp:42-57), no command-line adapters
    # With -a alone the reference leaves -s / -e on "auto" (src/options.cpp:209-214); SURVEY 8(d): pass them explicitly --
    # here the first FASTA adapter and its reverse complement, as tests/test_gpu_parity.py does for the same reads.
    "c5_hifi64": dict(opt=dict(), flags="--adapter_fasta (64 random 30-45-mers), -s <first adapter> -e <its reverse complement>, "
                      "defaults otherwise", config=4, reads=500_000,
                      gen=dict(kind="hifi", mean_len=20000, sd_len=2000, n_adapters=64, seed0=5),
                      reads_desc="N(20 kb, 2 kb) lengths, Q~N(35,6), 30 % of the reads with one FASTA adapter at an end, 1 % in the middle"),
}


def make_batch(wl, n_reads, rank, dev):
    """-> (seq_t, qual_t, off_t, max_len, start_adapter, end_adapter, fasta list)"""
    from fastplong_amd import synth

    g = dict(wl["gen"])
    kind = g.pop("kind")
    seed = g.pop("seed0", 1) + rank
    if kind == "hifi":
        seq_t, qual_t, off_t, max_len, ads = synth.device_batch_hifi(n_reads, seed=seed, device=dev, **g)
        return seq_t, qual_t, off_t, max_len, ads[0], synth.revcomp(ads[0]), ads
    seq_t, qual_t, off_t, max_len = synth.device_batch(n_reads, seed=seed, device=dev, **g)
    return seq_t, qual_t, off_t, max_len, synth.START_ADAPTER, synth.END_ADAPTER, []


def oracle_prefix(opt, adapters, seq, qual, off, C, threads):
    """The oracle (C restatement of the reference path; the checker) over a CSR batch held in numpy arrays, on `threads` host
    threads: contiguous shards, one per thread (the reference round-robins packs of 16 reads over <= 16 workers).
    -> (records of every read in input order, counter buffers summed over the shards, seconds)"""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle

    n = len(off) - 1
    cfg = oracle.Config(opt, adapters[0], adapters[1], adapters[2])
    oracle.lib()
    threads = max(1, min(threads, n))
    cuts = [int(i * n / threads) for i in range(threads + 1)]

    def work(t):
        a, b = cuts[t], cuts[t + 1]
        o = (off[a:b + 1] - off[a]).astype(np.uint64)
        return oracle.process_batch(cfg, seq[int(off[a]):int(off[b])], qual[int(off[a]):int(off[b])], o, max_cycles=C)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(work, range(threads)))
    dt = time.perf_counter() - t0
    res = np.concatenate([p[0] for p in parts])
    cnt = parts[0][1].copy()
    for p in parts[1:]:
        cnt += p[1]
    return res, cnt, dt


def parity_of_timed_batch(rig, opt, adapters, seq_t, qual_t, off_t, res_t, C, local_rank, n_prefix_bases, n_prefix_reads,
                          n_strided, threads):
    """Outside the timed region, the oracle as the checker (never the thing measured on the GPU side):
      (1) the RECORDS the timed step left in res_t -- the very batch, the very kernels that were timed -- for the first
          `n` reads of the batch and for `n_strided` reads spread evenly over the rest of it, against the oracle's records;
      (2) every COUNTER (Stats pre / post, FilterResult, adapter histogram) of the first `n` reads through a fresh context
          of the HIP library, against the oracle's counter buffer for those reads.  The prefix is large enough for the library
          to take the kernels of the timed batch by itself (k_trim_ends_batched from 65 536 reads, k_stats_sorted from
          150 000); when it is not (small --reads), the library's size hooks are set so that it takes them anyway.
    -> (verdict string, description, the oracle's seconds on the prefix, reads, bases of the prefix)"""
    import numpy as np
    from fastplong_amd import abi
    from tests import parity

    n_all = off_t.numel() - 1
    off_all = off_t.cpu().numpy().astype(np.int64)
    n = int(np.searchsorted(off_all, n_prefix_bases)) if n_prefix_bases > 0 else 0
    n = min(n_all, max(n, n_prefix_reads, min(n_all, 16)))
    off = off_all[:n + 1].astype(np.uint64)
    nb = int(off[-1])
    seq, qual = seq_t[:nb].cpu().numpy(), qual_t[:nb].cpu().numpy()
    want_res, want_cnt, dt = oracle_prefix(opt, adapters, seq, qual, off, C, threads)
    got_all = np.ascontiguousarray(res_t.cpu().numpy()).view(abi.RESULT_DTYPE)[:n_all]
    what = []
    try:
        # (1a) records of the timed batch, prefix
        parity.assert_results_equal(got_all[:n], want_res, seq, off)
        what.append("records of the TIMED batch (res_t of the last timed step): reads 0..%d" % (n - 1))
        # (1b) ... and a strided sample of the rest
        if n_strided > 0 and n_all > n:
            idx = np.unique(np.linspace(n, n_all - 1, min(n_strided, n_all - n)).astype(np.int64))
            lens = off_all[idx + 1] - off_all[idx]
            soff = np.zeros(len(idx) + 1, np.uint64)
            soff[1:] = np.cumsum(lens)
            it = torch_gather_reads(seq_t, qual_t, off_all, idx, soff)
            sres, _, _ = oracle_prefix(opt, adapters, it[0], it[1], soff, C, threads)
            parity.assert_results_equal(got_all[idx], sres, it[0], soff)
            what.append("+ %d reads spread evenly over reads %d..%d" % (len(idx), n, n_all - 1))
        # (2) counters of the prefix through a fresh context
        hooks = {}
        if n_all >= 65536 > n:
            hooks["FPL_TRIM_BATCH_MIN"] = "1"
        if n_all >= 150000 > n:
            hooks["FPL_STATS_SORT_MIN"] = "1"
        saved = {k: os.environ.get(k) for k in hooks}
        os.environ.update(hooks)  # (read once, in fpl_create)
        try:
            eng = rig.engine(opt, adapters[0], adapters[1], adapters[2], local_rank, C)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        try:
            rt = eng.process_device(seq_t[:nb], qual_t[:nb], off_t[:n + 1], C)
            rig.synchronize(seq_t.device)
            got_res = eng.results_to_numpy(rt, n)
            got_cnt = eng.counters()
            nad = eng.n_adapters
        finally:
            eng.close()
        parity.assert_results_equal(got_res, want_res, seq, off)
        parity.assert_counters_equal(got_cnt, want_cnt, C, nad)
        what.append("every counter (and the records again) of reads 0..%d (%d bases) through a fresh context%s" % (
            n - 1, nb, (" with " + " ".join("%s=%s" % kv for kv in sorted(hooks.items())) + " so that it takes the timed batch's kernels")
            if hooks else " (same kernels as the timed batch by the library's own size rules: no hook set)"))
        verdict = "ok"
    except AssertionError as e:
        verdict = "MISMATCH: " + str(e)[:300]
    return verdict, "; ".join(what) + "; bit for bit against oracle/liboracle.so, outside the timed region", dt, n, nb


def torch_gather_reads(seq_t, qual_t, off_all, idx, soff):
    """the reads `idx` of the resident batch as one small CSR batch in host memory"""
    import numpy as np

    total = int(soff[-1])
    seq = np.empty(total, np.uint8)
    qual = np.empty(total, np.uint8)
    for k, i in enumerate(idx):
        a, b = int(off_all[i]), int(off_all[i + 1])
        seq[int(soff[k]):int(soff[k + 1])] = seq_t[a:b].cpu().numpy()
        qual[int(soff[k]):int(soff[k + 1])] = qual_t[a:b].cpu().numpy()
    return seq, qual


def profile_record(workload):
    """What the separate rocprofv3 --pmc passes of this workload recorded (profiles/kernel_counters.json, written by
    profiles/summarize_profile.py): HBM bytes and vector wave-instructions per base and kernel.  {} when there is none."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "kernel_counters.json")))
        return rec.get("workloads", {}).get(workload, {})
    except (OSError, ValueError):
        return {}


def mem_headroom():
    """bytes this process may still put into the page cache / tmpfs: the cgroup's limit (v2 or v1) minus what is in use,
    and MemAvailable -- whichever is smaller"""
    lim = None
    for mx, cur in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                    ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            m = open(mx).read().strip()
            if m != "max" and int(m) < (1 << 60):
                lim = int(m) - int(open(cur).read().strip())
            break
        except (OSError, ValueError):
            continue
    avail = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
    except OSError:
        pass
    vals = [v for v in (lim, avail) if v is not None]
    return min(vals) if vals else None


CLI_FLAGS = {
    "c3_full_pipeline": ["--cut_front", "--cut_tail", "-W", "5", "-x", "-y"],
    "c4_mixed": ["--cut_front", "--cut_tail", "-W", "5", "-x", "-y"],
    "c2_adapter_only": [],
    "c5_hifi64": [],
}


E2E_TIMEOUT_S = 180  # one CLI run of the end-to-end leg (seconds)
E2E_SPLIT = 16  # files (and per-worker writer threads) of the to_split_files run: --split 16 -w 16
NULLDEV_DEVICES = 8  # the host_ceiling runs: bin/fastplong_amd --gpus 8 against tools/nulldev (a device that takes no time)
# One CLI run of the end-to-end leg: name, where the trimmed FASTQ goes (None: a file in the scratch directory), extra flags,
# "null": run against the null device library with that many devices (the host side alone), "input": which file it reads
# ("fq" the plain text; "gz_multi" / "gz_single": a gzip copy of the first GZ_READS reads, made of many members / of one)
E2E_RUNS = (
    dict(name="to_dev_null_first_pass", target="/dev/null"),
    dict(name="to_dev_null", target="/dev/null"),
    # the same with the HOST's parsers (AVX2 line scan + copies into page-locked CSR arrays); the default lets the device parse:
    # the chunk parsers only load the file's bytes (fpl_process_text_async)
    dict(name="host_parse", target="/dev/null", flags=["--host_parse"]),
    dict(name="to_file", target=None),
    dict(name="to_split_files", target=None, flags=["--split", str(E2E_SPLIT), "-w", str(E2E_SPLIT)]),
    # batches large enough for the kernel forms the headline times (csrc/pipeline.h: k_trim_ends_batched from 65 536 reads,
    # k_stats_sorted from 150 000): the CLI's batches are its parsers' chunks
    dict(name="chunk_512mb", target="/dev/null", flags=["--chunk_mb", "512", "--reader_threads", "8"]),
    dict(name="chunk_1536mb", target="/dev/null", flags=["--chunk_mb", "1536", "--reader_threads", "4"]),
    dict(name="gz_in_multi", target="/dev/null", input="gz_multi"),
    dict(name="gz_in_single", target="/dev/null", input="gz_single"),
    dict(name="gz_in_single_stream", target="/dev/null", input="gz_single", flags=["--gz_stream"]),
    dict(name="gz_out", target="GZ", input="fq_sub"),
    dict(name="null8_to_dev_null", target="/dev/null", null=NULLDEV_DEVICES),
    dict(name="null8_host_parse", target="/dev/null", null=NULLDEV_DEVICES, flags=["--host_parse"]),
    dict(name="null8_to_dev_null_rt16", target="/dev/null", null=NULLDEV_DEVICES, flags=["--reader_threads", "16"]),
    dict(name="null8_to_split_files", target=None, null=NULLDEV_DEVICES, flags=["--split", str(E2E_SPLIT), "-w", str(E2E_SPLIT)]),
    dict(name="null8_to_file", target=None, null=NULLDEV_DEVICES),
)
GZ_READS = 50_000  # reads of the gzip legs (the first reads of the batch)
E2E_DEFAULT_RUNS = ("to_dev_null_first_pass", "to_dev_null", "host_parse", "to_file", "to_split_files")
E2E_FULL_RUNS = E2E_DEFAULT_RUNS + ("gz_in_multi", "gz_in_single", "gz_in_single_stream", "gz_out")
E2E_LARGE_RUNS = ("to_dev_null", "host_parse", "to_file", "to_split_files", "chunk_512mb", "chunk_1536mb", "null8_to_dev_null",
                  "null8_host_parse", "null8_to_dev_null_rt16", "null8_to_split_files", "null8_to_file")


def gzip_single_member(src, dst, level=1):
    """src -> dst as ONE gzip member: libdeflate through ctypes when the image has it (one call on the whole text), else zlib"""
    import ctypes as C

    data = open(src, "rb").read()
    try:
        L = C.CDLL("libdeflate.so.0")
        L.libdeflate_alloc_compressor.restype = C.c_void_p
        L.libdeflate_gzip_compress_bound.restype = C.c_size_t
        L.libdeflate_gzip_compress_bound.argtypes = [C.c_void_p, C.c_size_t]
        L.libdeflate_gzip_compress.restype = C.c_size_t
        L.libdeflate_gzip_compress.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.libdeflate_free_compressor.argtypes = [C.c_void_p]
        comp = L.libdeflate_alloc_compressor(level)
        cap = L.libdeflate_gzip_compress_bound(comp, len(data))
        buf = C.create_string_buffer(cap)
        n = L.libdeflate_gzip_compress(comp, data, len(data), buf, cap)
        L.libdeflate_free_compressor(comp)
        assert n > 0
        with open(dst, "wb") as f:
            f.write(buf.raw[:n])
        return "libdeflate level %d" % level
    except (OSError, AssertionError, AttributeError):
        import gzip

        with gzip.open(dst, "wb", compresslevel=level) as f:
            f.write(data)
        return "zlib level %d" % level


def end_to_end(workload, opt, adapters, seq_t, qual_t, off_t, n_reads, n_gpus=1, copies=None, with_pcie=True, run_names=None):
    """End to end around the hot path, on the first n_reads reads of the resident batch -- reported NEXT to `value`,
    never as `value`:
      (1) the command line bin/fastplong_amd --gpus n_gpus: FASTQ text in the page cache (tmpfs; `copies` copies of the
          reads under names of their own, n_gpus by default) -> chunk-parallel parse into page-locked CSR batches ->
          H2D / kernels / D2H two batches deep per device -> trimmed FASTQ + fastplong.json / .html; wall time of the
          whole process (HIP start-up, reports and exit included) and of its host pipeline alone;
      (2) n_gpus == 1: the PCIe-inclusive C-ABI call, fpl_process_batch_async / fpl_wait from page-locked arrays, two deep."""
    import glob
    import shutil
    import signal
    import subprocess

    import ctypes as C
    import numpy as np
    import torch

    from fastplong_amd import abi, build, engine

    ad_start, ad_end, ad_fasta = adapters
    off = off_t[:n_reads + 1].cpu().numpy().astype(np.uint64)
    nb = int(off[-1])
    seq = seq_t[:nb].cpu().numpy()
    qual = qual_t[:nb].cpu().numpy()
    res = {"reads": n_reads, "bases": nb, "unit": "Gbases/s", "n_gpus": n_gpus}

    # (2) first: it needs the arrays page-locked, the file writer below does not care
    if with_pcie and n_gpus == 1:
        eng = engine.Engine(opt, ad_start, ad_end, ad_fasta, device=torch.cuda.current_device(),
                            max_cycles=int(np.diff(off.astype(np.int64)).max()))
        n_parts = 16
        n_pcie = min(n_reads, 400_000)  # (page-locking tens of gigabytes takes longer than the measurement)
        nb_pcie = int(off[n_pcie])
        cuts = [int(i * n_pcie / n_parts) for i in range(n_parts + 1)]
        parts = []
        for i in range(n_parts):
            a, b = cuts[i], cuts[i + 1]
            lo, hi = int(off[a]), int(off[b])
            ps, pq, po = eng.pinned_array(hi - lo), eng.pinned_array(hi - lo), eng.pinned_array(b - a + 1, np.uint64)
            ps[:], pq[:], po[:] = seq[lo:hi], qual[lo:hi], off[a:b + 1] - off[a]
            parts.append((ps, pq, po, np.zeros(b - a, dtype=abi.RESULT_DTYPE)))
        for rep in range(2):  # the first round allocates the staging slots
            t0 = time.perf_counter()
            for ps, pq, po, rr in parts:
                if eng.in_flight() == abi.FPL_MAX_IN_FLIGHT:
                    eng.wait()
                eng.submit_host(ps, pq, po, rr)
            while eng.in_flight():
                eng.wait()
            dt = time.perf_counter() - t0
        res["pcie_call"] = {"value": nb_pcie / dt / 1e9, "seconds": dt, "reads": n_pcie, "bases": nb_pcie,
                            "what": "fpl_process_batch_async/fpl_wait, %d batches from page-locked arrays, two in flight "
                                    "(H2D 2 B/base + kernels + D2H of the records)" % n_parts}
        eng.close()
        del parts

    # (1)
    build.build_host()
    host = C.CDLL(build.HOST_LIB)
    host.fplh_write_fastq_ex.restype = C.c_int
    host.fplh_write_fastq_ex.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_char_p, C.c_int, C.c_int]
    copies = copies or n_gpus
    per_copy = 2.0 * nb + 16.0 * n_reads
    # room per copy of the input in the scratch directory: the text itself, and as much again (+ slack) when a run writes its output there
    run_names = tuple(run_names) if run_names else E2E_DEFAULT_RUNS
    specs = [r for r in E2E_RUNS if r["name"] in run_names]
    writes = any(r.get("target") is None for r in specs)
    room = 2.3 if writes else 1.2
    tmp = None
    head = mem_headroom()
    for d in ("/dev/shm", "/tmp"):
        try:
            free = shutil.disk_usage(d).free
        except OSError:
            continue
        if d == "/dev/shm" and head is not None:
            free = min(free, head // 2)  # (tmpfs pages count against the container's memory; the run itself needs room too)
        fit = copies  # (per directory: what /dev/shm could not hold must not shrink the run that /tmp can)
        while fit > 1 and free < room * per_copy * fit:
            fit -= 1
        if free > room * per_copy * fit:
            tmp, copies = d, fit
            break
    if tmp is None:
        res["cli"] = None
        res["note"] = "no scratch directory with %.0f GB free" % (room * per_copy / 1e9)
        return res
    res["copies"] = copies
    res["bases"] = nb * copies
    res["reads"] = n_reads * copies
    fq = os.path.join(tmp, "fpl_e2e_%d.fq" % os.getpid())
    outp = os.path.join(tmp, "fpl_e2e_%d.out.fq" % os.getpid())
    fa = os.path.join(tmp, "fpl_e2e_%d.fa" % os.getpid())
    js, html = fq + ".json", fq + ".html"
    try:
        t0 = time.perf_counter()
        for c in range(copies):
            rc = host.fplh_write_fastq_ex(fq.encode(), seq.ctypes.data, qual.ctypes.data, off.ctypes.data, n_reads,
                                          b"r" if copies == 1 else b"c%d_" % c, 16, 1 if c else 0)
            assert rc == 0, "writing %s failed" % fq
        res["input"] = "%s, %.2f GB of FASTQ text (%d x %d reads, written in %.1f s, in the page cache)" % (
            fq, os.path.getsize(fq) / 1e9, copies, n_reads, time.perf_counter() - t0)
        cmd = [build.CLI, "-i", fq, "-s", ad_start, "-e", ad_end, "-j", js, "-h", html, "-V"] + CLI_FLAGS[workload]
        if n_gpus > 1:
            cmd += ["--gpus", str(n_gpus)]
        if ad_fasta:
            with open(fa, "w") as f:
                for i, a in enumerate(ad_fasta):
                    f.write(">ad%03d\n%s\n" % (i, a))
            cmd += ["-a", fa]
        runs = {}
        json_check = None
        # gzip copies of the first GZ_READS reads for the gz legs: many members (what bgzip, fastp / fastplong and this host write;
        # made by this host: -o x.fq.gz with every stage off) and ONE member (what `gzip file.fq` writes; libdeflate through ctypes)
        sub = os.path.join(tmp, "fpl_e2e_%d.sub.fq" % os.getpid())
        gz_multi, gz_single, gz_outp = sub + ".multi.gz", sub + ".single.gz", sub + ".out.fq.gz"
        n_sub = min(n_reads, GZ_READS)
        nb_sub = int(off[n_sub])
        if any(r.get("input", "fq") != "fq" for r in specs):
            rc = host.fplh_write_fastq_ex(sub.encode(), seq.ctypes.data, qual.ctypes.data, off.ctypes.data, n_sub, b"r", 16, 0)
            assert rc == 0
            t0 = time.perf_counter()
            subprocess.run([build.CLI, "-i", sub, "-o", gz_multi, "-A", "-Q", "-L", "-j", js, "-h", html], stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, timeout=E2E_TIMEOUT_S, check=True)
            t1 = time.perf_counter()
            single_by = gzip_single_member(sub, gz_single)
            res["gz_inputs"] = "%d reads, %.2f GB of text -> %.2f GB in many members (this host, %.1f s), %.2f GB in one member (%s, %.1f s)" % (
                n_sub, os.path.getsize(sub) / 1e9, os.path.getsize(gz_multi) / 1e9, t1 - t0, os.path.getsize(gz_single) / 1e9,
                single_by, time.perf_counter() - t1)
        null_dir = None
        if any(r.get("null") for r in specs):
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
            from nulldev import build as null_build

            null_dir = os.path.dirname(null_build.build())
        # the first pass over a file that has only just been written pays the kernel's first-touch bookkeeping of its
        # page-cache pages (every read marks them accessed / moves them between LRU lists, under contention from 16
        # parser threads): it is reported, the steady state is the second pass
        for spec in specs:
            name = spec["name"]
            target = spec.get("target")
            target = gz_outp if target == "GZ" else (target or outp)
            which = spec.get("input", "fq")
            infile = {"fq": fq, "fq_sub": sub, "gz_multi": gz_multi, "gz_single": gz_single}[which]
            run_bases = nb * copies if which == "fq" else nb_sub
            # one file on tmpfs takes 6-9 GB/s whoever writes it (page-cache insertion of ONE inode serialises in the kernel);
            # --split N gives every worker's writer thread a file of its own
            flags = list(spec.get("flags", []))
            env = dict(os.environ, FPLH_T0=repr(time.time()), FPLH_TIMING="1")
            run_cmd = [build.CLI, "-i", infile] + cmd[3:]
            if spec.get("null"):  # the host side alone: N devices that take no time (tools/nulldev/fpl_null.cpp)
                env["LD_LIBRARY_PATH"] = null_dir + os.pathsep + env.get("LD_LIBRARY_PATH", "")
                env["FPL_NULL_DEVICES"] = str(spec["null"])
                run_cmd = [a for a in run_cmd]
                if "--gpus" in run_cmd:
                    i = run_cmd.index("--gpus")
                    del run_cmd[i:i + 2]
                run_cmd += ["--gpus", str(spec["null"])]
            import resource

            ru0 = resource.getrusage(resource.RUSAGE_CHILDREN)
            t0 = time.perf_counter()
            # (a run that does not come back must not take the bench line with it: its own session, so that the whole process
            # group can be killed, and a bounded wait for its pipes afterwards -- a process stuck in the driver may never close them)
            proc = subprocess.Popen(run_cmd + flags + ["-o", target], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                    start_new_session=True, env=env)
            try:
                so, se = proc.communicate(timeout=E2E_TIMEOUT_S)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
                try:
                    so, se = proc.communicate(timeout=10)
                except subprocess.TimeoutExpired:
                    so, se = "", "(the process did not go away within 10 s of SIGKILL)"
                runs[name] = {"rc": -1, "process_seconds": time.perf_counter() - t0, "value": 0.0, "pipeline_seconds": None,
                              "pipeline_value": None, "stages": [], "stderr_tail": "timed out after %d s: %s" % (E2E_TIMEOUT_S, (se or "")[-300:])}
                break
            import types
            p = types.SimpleNamespace(returncode=proc.returncode, stderr=se or "")
            dt = time.perf_counter() - t0
            pipe = None
            keep = []
            for line in p.stderr.splitlines():
                if line.startswith("host pipeline:"):
                    pipe = float(line.split("wall ")[1].split(" s")[0])
                if line.startswith(("host pipeline:", "start-up:", "reports:", "since launch:", "chunk parsers", "counter merge:", "kernel forms:", "input:", "device parse:")):
                    keep.append(line)
            ru1 = resource.getrusage(resource.RUSAGE_CHILDREN)
            cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
            runs[name] = {"rc": p.returncode, "process_seconds": dt, "value": run_bases / dt / 1e9, "bases": run_bases,
                          "pipeline_seconds": pipe, "pipeline_value": (run_bases / pipe / 1e9) if pipe else None, "stages": keep,
                          # host CPU the whole process took (user + system, all threads): what bounds an N-device run on a host
                          "cpu_seconds": cpu_s, "cpu_seconds_per_gbase": cpu_s / (run_bases / 1e9)}
            if flags:
                runs[name]["flags"] = " ".join(flags)
            if name == "to_dev_null" and p.returncode == 0:  # (the report of THIS run: later runs write the same file names)
                jr = json.load(open(js))
                json_check = {"reads_in": jr["summary"]["before_filtering"]["total_reads"],
                              "reads_out": jr["summary"]["after_filtering"]["total_reads"],
                              "ok": jr["summary"]["before_filtering"]["total_reads"] == n_reads * copies}
            if which != "fq" and p.returncode == 0:  # the gz legs must see every read of their input too
                jr = json.load(open(js))
                runs[name]["reads_in"] = jr["summary"]["before_filtering"]["total_reads"]
                runs[name]["ok"] = runs[name]["reads_in"] == n_sub
            if spec.get("null"):
                runs[name]["what"] = ("the HOST side alone: bin/fastplong_amd --gpus %d against tools/nulldev (devices that take no time, "
                                      "no page-locking, no PCIe traffic) on the %d CPUs of this box" % (spec["null"], len(os.sched_getaffinity(0))))
            if p.returncode != 0:
                runs[name]["stderr_tail"] = p.stderr[-500:]
            if "--split" in flags:
                parts = sorted(glob.glob(os.path.join(tmp, "*." + os.path.basename(outp))))
                runs[name]["files"] = len(parts)
                runs[name]["bytes_written"] = sum(os.path.getsize(f) for f in parts)
                runs[name]["what"] = runs[name].get("what", "") + " --split %d -w %d: %d files, one writer thread per worker (writev gather lists)" % (
                    E2E_SPLIT, E2E_SPLIT, len(parts))
                for f in parts:
                    try:
                        os.remove(f)
                    except OSError:
                        pass
            elif target in (outp, gz_outp) and os.path.exists(target):
                runs[name]["bytes_written"] = os.path.getsize(target)
            for f in (outp, gz_outp):
                try:
                    os.remove(f)
                except OSError:
                    pass
        res["cli"] = runs
        if runs.get("to_dev_null", {}).get("rc") == 0:
            res["value"] = runs["to_dev_null"]["value"]
            # what `value` wrote where -- a reader of the N-GPU line must not take it for a file sink
            res["sink"] = "/dev/null (the trimmed FASTQ goes to /dev/null as gather lists over the batches' arrays; JSON + HTML reports are written)"
            res["command"] = "bin/fastplong_amd --gpus %d -i <%d x %d reads in tmpfs> -o /dev/null" % (n_gpus, copies, n_reads)
            res["what"] = ("bin/fastplong_amd%s -i <FASTQ in tmpfs> -o /dev/null + JSON + HTML: input bases / wall time of the "
                           "whole process; cli.to_file = the same with the trimmed FASTQ written to tmpfs; "
                           "pipeline_value = without process start-up (HIP context) and the report writers" % (
                               " --gpus %d" % n_gpus if n_gpus > 1 else ""))
            if json_check:
                res["json_check"] = json_check
    finally:
        for f in [fq, outp, js, html, fa, fq.replace('.fq', '.sub.fq')] + glob.glob(os.path.join(tmp, "fpl_e2e_%d.sub.fq.*" % os.getpid())) + glob.glob(os.path.join(tmp, "*." + os.path.basename(outp))):
            try:
                os.remove(f)
            except OSError:
                pass
    return res


def _r(x, nd=6):
    """floats to `nd` significant digits (the line is for reading; the full object keeps every digit)"""
    return float("%.*g" % (nd, x)) if isinstance(x, float) else x


def compact_line(out):
    """The stdout line: the contract's fields, `roofline`, `cpu_baseline`, the two self-checks and an `e2e` of scalars --
    numbers and short strings only.  Everything else (per-stage prose of the CLI runs, descriptions) stays in the full
    object (--full-json and stderr)."""
    c = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                             "vs_baseline", "dtype", "data") if k in out}
    cfg = out.get("config", {})
    c["config"] = {"workload": cfg.get("workload", "")[:160], "reads_per_gpu": cfg.get("reads_per_gpu"),
                   "bases_per_gpu": cfg.get("bases_per_gpu"), "max_read_len": cfg.get("max_read_len"),
                   "parallelism": cfg.get("parallelism", "").split(" ")[0], "trims_ahead": not cfg.get("pipelining", "none").startswith("none")}
    rf = out.get("roofline")
    if rf:
        c["roofline"] = {k: _r(rf.get(k)) for k in ("bound", "limited_by", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                                    "traffic_source", "algorithmic_bytes_per_launch")}
        c["roofline"]["kernel_ms"] = {k: _r(v) for k, v in (rf.get("kernel_ms") or {}).items()}
        if rf.get("kernel_ms_in_line"):
            c["roofline"]["k_trim_ends_ms_in_line"] = _r(rf["kernel_ms_in_line"].get("k_trim_ends"))
        c["roofline"]["kernel_ms_alone"] = _r(rf.get("kernel_ms_alone"))
        c["roofline"]["frac_alone"] = _r(rf.get("frac_alone"))
        if rf.get("issue"):
            c["roofline"]["issue_frac_of_kernel_time"] = _r(rf["issue"]["frac_of_kernel_time"])
        pa = rf.get("path") or {}
        c["roofline"]["path"] = {k: _r(pa.get(k)) for k in ("achieved", "frac", "measured_bytes_per_base")}
    cb = out.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = dict(cb, value=_r(cb["value"]), sample=cb["sample"][:120])
    if "counters_check" in out:
        c["counters_check"] = out["counters_check"]
    if "parity_sample" in out:
        c["parity_sample"] = out["parity_sample"][:200]
    e = out.get("e2e")
    if e:
        ce = {"unit": "Gbases/s", "n_gpus": e.get("n_gpus"), "reads": e.get("reads")}
        if "error" in e:
            ce["error"] = e["error"][:200]

        def leg(runs, name, key="value"):
            r = (runs or {}).get(name)
            return _r(r.get(key), 4) if r and r.get("rc") == 0 and r.get(key) is not None else None

        runs = e.get("cli") or {}
        ce["value"] = _r(e.get("value"), 4)
        ce["pipeline_value"] = leg(runs, "to_dev_null", "pipeline_value")
        ce["first_pass"] = leg(runs, "to_dev_null_first_pass")
        ce["cpu_s_per_gbase"] = leg(runs, "to_dev_null", "cpu_seconds_per_gbase")
        ce["parse"] = "device"  # (the default: fpl_process_text_async; host_parse*: the same run with --host_parse)
        ce["host_parse"] = leg(runs, "host_parse")
        ce["host_parse_pipeline"] = leg(runs, "host_parse", "pipeline_value")
        ce["host_parse_cpu_s_per_gbase"] = leg(runs, "host_parse", "cpu_seconds_per_gbase")
        ce["to_file"] = leg(runs, "to_file")
        ce["to_split"] = leg(runs, "to_split_files")
        if e.get("pcie_call"):
            ce["pcie_call"] = _r(e["pcie_call"]["value"], 4)
        if e.get("json_check"):
            ce["json_ok"] = e["json_check"].get("ok")
        for name, short in (("gz_in_multi", "gz_in"), ("gz_in_single", "gz_in_single"), ("gz_out", "gz_out")):
            if name in runs:
                ce[short] = leg(runs, name)
        big = e.get("large_input") or {}
        if big:
            br = big.get("cli") or {}
            ce["large"] = {"reads": big.get("reads"), "value": leg(br, "to_dev_null"), "pipeline_value": leg(br, "to_dev_null", "pipeline_value"),
                           "to_file": leg(br, "to_file"), "to_split": leg(br, "to_split_files"),
                           "chunk_512mb": leg(br, "chunk_512mb"), "chunk_1536mb": leg(br, "chunk_1536mb"),
                           "host_parse": leg(br, "host_parse"), "null8_host_parse": leg(br, "null8_host_parse", "pipeline_value"),
                           "null8": leg(br, "null8_to_dev_null", "pipeline_value"), "null8_rt16": leg(br, "null8_to_dev_null_rt16", "pipeline_value"),
                           "null8_to_file": leg(br, "null8_to_file")}
            if "error" in big:
                ce["large"] = {"error": big["error"][:200]}
        failed = [n for n, r in list(runs.items()) + list((big.get("cli") or {}).items()) if r.get("rc") != 0]
        if failed:
            ce["failed_runs"] = failed
        c["e2e"] = ce
    return c


def emit(out, full_path):
    """the full object -> `full_path` and stderr; its compact form -> ONE stdout line of at most LINE_LIMIT bytes"""
    full = json.dumps(out)
    if full_path:
        try:
            os.makedirs(os.path.dirname(full_path) or ".", exist_ok=True)
            with open(full_path, "w") as f:
                f.write(full + "\n")
        except OSError as e:
            print("bench.py: could not write %s: %s" % (full_path, e), file=sys.stderr)
    print(full, file=sys.stderr, flush=True)
    c = compact_line(out)
    c["full"] = full_path
    line = json.dumps(c)
    if len(line) > LINE_LIMIT:  # never again a line the driver cannot read: shed the optional objects, longest first
        for k in ("e2e", "counters_check", "parity_sample"):
            if len(line) <= LINE_LIMIT:
                break
            c[k] = "see " + str(full_path)
            line = json.dumps(c)
    print(line, flush=True)


class Rig:
    """What the rank code of main() stands on: the device, the collective backend, where the batch and the engine come
    from.  bench.py always runs the default -- a GPU, RCCL, the HIP library; tests/test_distributed_gloo.py swaps in CPU
    tensors, gloo and a stand-in engine to run THIS rank code (sharding by seed, capacity agreement, timed region,
    counter all-reduce, MAX / SUM reductions of the line) with world_size 2 on a box without GPUs."""
    backend = "nccl"

    def device(self, local_rank):
        import torch

        assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU path)"
        torch.cuda.set_device(local_rank)
        return torch.device("cuda", local_rank)

    def synchronize(self, dev):
        import torch

        torch.cuda.synchronize(dev)

    def stream(self, dev):
        import torch

        return torch.cuda.current_stream(dev).cuda_stream

    def make_batch(self, wl, n_reads, rank, dev):
        return make_batch(wl, n_reads, rank, dev)

    def engine(self, opt, ad_start, ad_end, ad_fasta, local_rank, C):
        from fastplong_amd import engine

        return engine.Engine(opt, ad_start, ad_end, ad_fasta, device=local_rank, max_cycles=C)


def main(argv=None, rig=None):
    rig = rig or Rig()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (0 = the workload's own: 1 M for c2/c3, 2.5 M for c4, 0.5 M for c5)")
    ap.add_argument("--median-len", type=int, default=0, help="ablation only: median read length of the ONT-like workloads")
    ap.add_argument("--e2e-reads", type=int, default=1_000_000,
                    help="reads of rank 0's batch written as FASTQ (once per GPU) and run through ONE bin/fastplong_amd --gpus N (0 = skip)")
    ap.add_argument("--e2e-copies", type=int, default=4,
                    help="N = 1: a second end-to-end run over this many copies of the reads, so that the process's fixed cost "
                         "(HIP context, reports, exit) weighs less (0 or 1 = skip)")
    ap.add_argument("--parity-reads", type=int, default=200_000,
                    help="outside the timed region: the first reads of the timed batch (at N = 1 at least the CPU baseline's sample) whose "
                         "records and counters are checked against the oracle (0 = skip)")
    ap.add_argument("--parity-strided", type=int, default=2000,
                    help="... and this many reads spread evenly over the rest of the timed batch (records only)")
    ap.add_argument("--workload", default="c3_full_pipeline", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-bases", type=float, default=6e9, help="size of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--set", default="", help="ablation only: comma separated fpl_options overrides, e.g. adapter_enabled=0")
    ap.add_argument("--e2e-full", action="store_true",
                    help="also run the long end-to-end legs: gzip in / out, the large input (--e2e-copies) with --chunk_mb and the "
                         "null-device host-ceiling runs (minutes; the default run keeps to /dev/null, one file, --split on --e2e-reads)")
    ap.add_argument("--full-json", default=os.path.join("gpurun_out", "bench_full.json"),
                    help="where the FULL result object goes (the stdout line is its compact form, a few KB)")
    ap.add_argument("--hbm-traffic", type=float, default=None,
                    help="HBM bytes per launch of the dominant kernel from a separate rocprofv3 --pmc run")
    args = ap.parse_args(argv)

    import numpy as np
    import torch
    import torch.distributed as dist

    from fastplong_amd import abi, dist as fdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d" % (
                args.gpus, args.gpus))
    dev = rig.device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dev.type == "cuda":
            dist.init_process_group(rig.backend, rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(rig.backend, rank=rank, world_size=world)

    wl = WORKLOADS[args.workload]
    okw = dict(wl["opt"])
    for kv in filter(None, args.set.split(",")):
        k, val = kv.split("=")
        okw[k] = float(val) if k == "ed_max" else int(val)
    opt = abi.FplOptions.default(**okw)
    # synthetic shard of this rank (weak scaling: every rank gets its own reads)
    if args.median_len:
        wl = dict(wl, gen=dict(wl["gen"], median_len=args.median_len))
    n_want = args.reads if args.reads > 0 else wl["reads"]
    seq_t, qual_t, off_t, max_len, ad_start, ad_end, ad_fasta = rig.make_batch(wl, n_want, rank, dev)
    n = off_t.numel() - 1
    n_bases = int(off_t[-1].item())
    # all ranks agree on the per-cycle capacity so that the counter buffers line up for the all-reduce
    C = fdist.agree_capacity(max_len, device=dev)
    eng = rig.engine(opt, ad_start, ad_end, ad_fasta, local_rank, C)
    # the batch is resident in HBM before the timed region starts: say so, and the library starts the end trims of step k + 1
    # beside the kernels of step k (fpl_assume_inputs_ready; the asynchronous host path does the same from its own copy events)
    trim_ahead = False
    if hasattr(eng, "assume_inputs_ready") and not os.environ.get("FPL_NO_TRIM_AHEAD"):
        eng.assume_inputs_ready(True)
        trim_ahead = True
    res_t = torch.empty(n * 36, dtype=torch.uint8, device=dev)
    stream = rig.stream(dev)

    def step():
        eng.process_device(seq_t, qual_t, off_t, max_len, res_t, stream)

    for _ in range(args.warmup):
        step()
    rig.synchronize(dev)
    eng.reset_counters()
    eng.enable_timing(True)  # HIP events around every kernel, on the launch stream, inside the timed region
    if world > 1:
        dist.barrier()
    rig.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        # the only collective of the path: sum the Stats / FilterResult counters over RCCL
        fdist.allreduce_counters(eng.counters_tensor())
    rig.synchronize(dev)
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
    counters_timed = eng.counters() if rank == 0 else None  # (the counters of the timed steps: the pass below adds to them)
    # With the end trims ahead of the main stream, the k_trim_ends stage of the timed region is what is LEFT of them in line, not a
    # kernel duration.  Three more steps with every kernel in line (promise withdrawn), outside the timed region, give the stage
    # times the roofline object needs when k_trim_ends is the dominant stage (c5); the other stages' in-region times stand.
    inline_ms = None
    if trim_ahead:
        eng.assume_inputs_ready(False)
        eng.enable_timing(True)
        for _ in range(3):
            step()
        rig.synchronize(dev)
        kin, nin = eng.kernel_times()
        inline_ms = {k: kin[k] / max(1, nin) for k in kin}
    eng.enable_timing(False)

    cpu_group = None
    if world > 1:  # a CPU-side barrier for the end-to-end leg: the other ranks must not spin on their GPUs meanwhile
        try:
            cpu_group = dist.new_group(backend="gloo")
        except Exception:
            cpu_group = None
    n_cu = 256
    try:
        n_cu = int(torch.cuda.get_device_properties(dev).multi_processor_count) if dev.type == "cuda" else 256
    except Exception:
        pass
    adapters = (ad_start, ad_end, ad_fasta)
    out = None
    if rank == 0:
        counters = counters_timed
        v = abi.CountersView(counters, C, eng.n_adapters)
        ms_per_step = dt / args.steps * 1e3
        # HBM bytes and vector wave-instructions per launch: PMC counters come from separate rocprofv3 --pmc passes of
        # this same command (profiles/kernel_counters.json records them per base, HBM bytes corrected as
        # MI355X_MICROARCH.md prescribes); --hbm-traffic overrides the dominant kernel's, null when nothing is recorded
        prof = {} if (args.set or args.median_len) else profile_record(args.workload)
        hbm_pb, valu_pb = prof.get("hbm_bytes_per_base", {}), prof.get("valu_insts_per_base", {})
        # (HIP-event stages, csrc/pipeline.h: k_trim_ends, k_scan and k_stats are single launches -- their event times are kernel
        # durations, and the dominant kernel is the longest of them; k_resolve, k_stats_prep, k_stats_reduce, k_stats_extra
        # group short kernels)
        per_step = {k: ktimes[k] / max(1, nbatches) for k in ktimes}
        if inline_ms and "k_trim_ends" in inline_ms:
            per_step = dict(per_step, k_trim_ends=inline_ms["k_trim_ends"])  # (its duration, not what the overlap leaves in line)
        single = {k: v for k, v in per_step.items() if k in ("k_trim_ends", "k_scan", "k_stats")} or per_step
        dom = max(single, key=single.get)
        dom_ms = per_step[dom]
        achieved = ALGO_BYTES_PER_BASE * n_bases / (dom_ms * 1e-3) / 1e9
        # (the counters' names: the trim stage is k_trim_ends_batched, the wave-per-read k_trim_ends, or -- with a FASTA list -- both;
        # the one that issues more is the one the stage's time belongs to)
        trim_names = [k for k in ("k_trim_ends_batched", "k_trim_ends") if k in hbm_pb]
        dom_kernel = {"k_stats": "k_stats_sorted" if "k_stats_sorted" in hbm_pb else "k_stats",
                      "k_trim_ends": max(trim_names, key=lambda k: valu_pb.get(k, 0.0)) if trim_names else "k_trim_ends"}.get(dom, dom)
        traffic, traffic_src = args.hbm_traffic, "--hbm-traffic"
        if traffic is None and dom_kernel in hbm_pb:
            traffic, traffic_src = hbm_pb[dom_kernel] * n_bases, prof.get("source")
        path_bytes = sum(hbm_pb.values()) * n_bases if hbm_pb else None
        path_achieved = ALGO_BYTES_PER_BASE * n_bases / (ms_per_step * 1e-3) / 1e9
        # vector issue: one wave-instruction per clock and CU for mixed code (measured, tools/ubench)
        clock_ghz = 2.4
        issue = None
        if dom_kernel in valu_pb:
            v_insts = valu_pb[dom_kernel] * n_bases
            issue_ms = v_insts / (n_cu * clock_ghz * 1e9) * 1e3
            issue = {"kernel": dom_kernel, "valu_wave_insts_per_launch": v_insts, "cus": n_cu, "clock_ghz": clock_ghz,
                     "issue_ms": issue_ms, "kernel_ms": dom_ms, "frac_of_kernel_time": issue_ms / dom_ms,
                     "model": "1 vector wave-instruction per clock and CU (mixed code, measured: profiles/r02_ubench)",
                     "source": prof.get("source")}
        limited_by = "hbm"
        if issue and issue["frac_of_kernel_time"] > achieved / HBM_PEAK_GBS:
            limited_by = "valu_issue"
        reads_ok = int(v.pre.reads) == args.steps * n * (world if world > 1 else 1) and \
            int(v.pre.length_sum) == args.steps * total_bases
        out = {
            "metric": "Gbases/s processed (trim+cut+filter)",
            "value": total_bases * args.steps / dt / 1e9,
            "unit": "Gbases/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": "%s%s: BASELINE.json configs[%d] -- %d synthetic reads per GPU, %s, %s; inputs resident in HBM" % (
                    args.workload, (" [ABLATION " + " ".join(filter(None, [args.set, args.median_len and "median-len=%d" % args.median_len])) + "]")
                    if (args.set or args.median_len) else "", wl["config"], n, wl["reads_desc"], wl["flags"]),
                "reads_per_gpu": n, "bases_per_gpu": n_bases, "max_read_len": max_len,
                "parallelism": "shard%d (independent read shards, one RCCL all-reduce of the counters)" % world,
                "pipelining": ("the steps are enqueued back to back on one stream; the end trims of step k + 1 run on a stream of their own beside "
                               "k_scan of step k, two of their blocks per CU (fpl_assume_inputs_ready: the batch is resident), so "
                               "kernel_ms.k_trim_ends is what is LEFT of them in line and kernel_ms.k_scan is k_scan WITH the trims beside it -- "
                               "roofline.kernel_ms_alone / frac_alone: the same kernel with nothing beside it") if trim_ahead else "none: every kernel of a step in line",
            },
            "roofline": {
                # priced against the HBM roofline as the contract prescribes (algorithmic bytes / kernel time / 8 TB/s);
                # `limited_by` says what the counters say the kernel actually waits for
                "bound": "hbm", "limited_by": limited_by, "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src if traffic is not None else None,
                "algorithmic_bytes_per_launch": ALGO_BYTES_PER_BASE * n_bases,
                "kernel_ms": {k: ktimes[k] / max(1, nbatches) for k in ktimes},
                "kernel_ms_in_line": inline_ms,  # three steps behind the timed region with the trims NOT ahead: k_trim_ends' own duration
                # the dominant kernel ALONE: the timed region runs it beside the next batch's end trims (two of their blocks per CU, each in the
                # place of one of its five), which lengthens it; three in-line steps behind the region time it with nothing beside it
                "kernel_ms_alone": (inline_ms or {}).get(dom), "frac_alone": (ALGO_BYTES_PER_BASE * n_bases / (inline_ms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS)
                if inline_ms and inline_ms.get(dom) else None,
                "duration_source": ("HIP events of the timed region" if not (inline_ms and dom == "k_trim_ends") else
                                    "HIP events of three in-line steps behind the timed region (in the region the stage overlaps the step before); "
                                    "with a FASTA list the stage is two launches, k_trim_ends_batched<8 words, chain> + k_trim_ends<2>"),
                "issue": issue,
                "path": {"what": "the whole launch sequence of a step against the same roofline: 2 B/base x bases / ms_per_step",
                         "achieved": path_achieved, "frac": path_achieved / HBM_PEAK_GBS,
                         "measured_bytes_per_base": (path_bytes / n_bases) if path_bytes else None,
                         "ratio_to_algorithmic": (path_bytes / (ALGO_BYTES_PER_BASE * n_bases)) if path_bytes else None},
            },
            "counters_check": {"reads_in": int(v.pre.reads), "bases_in": int(v.pre.length_sum),
                               "fragments_out": int(v.post.reads), "bases_out": int(v.post.length_sum),
                               "expected_reads_in": args.steps * n * world if world > 1 else args.steps * n,
                               "expected_bases_in": args.steps * total_bases, "ok": bool(reads_ok)},
        }
        if args.parity_reads > 0:
            # ONE run of the oracle serves both: the checker of the timed batch's records / counters and -- at N = 1 -- the CPU
            # baseline (its seconds over the same reads; rank 0 at N = 1 only: the other ranks would sit in the barrier meanwhile)
            threads = max(1, min(os.cpu_count() or 1, 16))
            with_cpu = args.cpu_bases > 0 and world == 1
            verdict, pwhat, odt, pn, pb = parity_of_timed_batch(rig, opt, adapters, seq_t, qual_t, off_t, res_t, C, local_rank,
                                                                args.cpu_bases if with_cpu else 0, min(n, args.parity_reads),
                                                                args.parity_strided, threads)
            out["parity_sample"] = verdict
            out["parity_sample_what"] = pwhat
            if with_cpu:
                out["cpu_baseline"] = {"value": pb / odt / 1e9, "unit": "Gbases/s", "cores": min(threads, pn), "kind": "port",
                                       "reference_buildable": False, "blocked_by": REFERENCE_BLOCKED_BY,
                                       "sample": "first %d reads (%d bases) of the same batch, oracle/liboracle.so, %d threads, %.1f s "
                                                 "(the run whose records and counters parity_sample compares)" % (pn, pb, min(threads, pn), odt)}
        elif args.cpu_bases > 0 and world == 1:
            import numpy as _np
            offc = off_t.cpu().numpy().astype(_np.int64)
            pn = max(16, min(int(_np.searchsorted(offc, args.cpu_bases)), n))
            pb = int(offc[pn])
            threads = max(1, min(os.cpu_count() or 1, 16))
            _, _, odt = oracle_prefix(opt, adapters, seq_t[:pb].cpu().numpy(), qual_t[:pb].cpu().numpy(), offc[:pn + 1].astype(_np.uint64),
                                      C, threads)
            out["cpu_baseline"] = {"value": pb / odt / 1e9, "unit": "Gbases/s", "cores": min(threads, pn), "kind": "port",
                                   "reference_buildable": False, "blocked_by": REFERENCE_BLOCKED_BY,
                                   "sample": "first %d reads (%d bases) of the same batch, oracle/liboracle.so, %d threads, %.1f s" % (
                                       pn, pb, min(threads, pn), odt)}
    eng.close()  # (idempotent)
    run_e2e = args.e2e_reads > 0 and not args.set and dev.type == "cuda"
    if run_e2e and world > 1:
        # the end-to-end number the >= 30 Gbases/s target is about: ONE bin/fastplong_amd --gpus N over an N-times larger
        # FASTQ in tmpfs.  The ranks give their device memory back first and wait on the CPU.
        keep = None
        if rank == 0:
            ne = min(n, args.e2e_reads)
            keep = (seq_t[:int(off_t[ne].item())].cpu(), qual_t[:int(off_t[ne].item())].cpu(), off_t[:ne + 1].cpu(), ne)
        del seq_t, qual_t, off_t, res_t
        torch.cuda.empty_cache()
        if cpu_group is not None:
            dist.barrier(group=cpu_group)
        if rank == 0:
            try:
                out["e2e"] = end_to_end(args.workload, opt, adapters, keep[0], keep[1], keep[2], keep[3], n_gpus=world,
                                        with_pcie=False, run_names=("to_dev_null_first_pass", "to_dev_null"))
            except Exception as e:  # the line must still be printed
                out["e2e"] = {"error": repr(e)[:300]}
        if cpu_group is not None:
            dist.barrier(group=cpu_group)
    elif run_e2e and rank == 0:
        ne = min(n, args.e2e_reads)
        out["e2e"] = end_to_end(args.workload, opt, adapters, seq_t, qual_t, off_t, ne,
                                run_names=E2E_FULL_RUNS if args.e2e_full else E2E_DEFAULT_RUNS)
        if args.e2e_full and args.e2e_copies > 1:
            try:
                big = end_to_end(args.workload, opt, adapters, seq_t, qual_t, off_t, ne, copies=args.e2e_copies,
                                 with_pcie=False, run_names=E2E_LARGE_RUNS)
                out["e2e"]["large_input"] = big
            except Exception as e:
                out["e2e"]["large_input"] = {"error": repr(e)[:300]}
    if rank == 0:
        emit(out, args.full_json)
    if world > 1:
        if cpu_group is not None:
            dist.barrier(group=cpu_group)
        else:
            dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
