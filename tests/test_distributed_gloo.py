"""The N > 1 path on the CPU: world_size-2 `gloo` processes, each owning a shard of the reads;
capacity agreement (MAX) + counter all-reduce (SUM) must reproduce the single-process counters.
The per-rank counter buffers are produced by the oracle here (data for the collective under test);
on GPUs they come from fpl_counters_device_ptr() and the backend is RCCL."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fastplong_amd import abi, dist as fdist, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    from oracle import oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = oracle.Config(abi.FplOptions.default(cut_front=1, cut_tail=1, polyx=1, complexity_filter=1),
                        synth.START_ADAPTER, synth.END_ADAPTER)
    seq, qual, off = synth.ont_like(40, seed=77, median_len=700, p_middle=0.2)
    a, b = fdist.shard_range(len(off) - 1, rank, world)
    lo, hi = int(off[a]), int(off[b])
    o = (off[a:b + 1] - off[a]).astype(np.uint64)
    c_local = int(np.diff(o.astype(np.int64)).max())  # ranks start with different capacities
    _, cnt = oracle.process_batch(cfg, seq[lo:hi], qual[lo:hi], o, max_cycles=c_local)
    merged, c = fdist.merge_host_counters(cnt, c_local, cfg.n_adapters)
    np.save(os.path.join(outdir, "merged_%d.npy" % rank), merged)
    with open(os.path.join(outdir, "c_%d.txt" % rank), "w") as f:
        f.write(str(c))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_counters_allreduce_gloo(orc, tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    cfg = orc.Config(abi.FplOptions.default(cut_front=1, cut_tail=1, polyx=1, complexity_filter=1),
                     synth.START_ADAPTER, synth.END_ADAPTER)
    seq, qual, off = synth.ont_like(40, seed=77, median_len=700, p_middle=0.2)
    c = int(np.diff(off.astype(np.int64)).max())
    _, want = orc.process_batch(cfg, seq, qual, off, max_cycles=c)
    for r in range(world):
        assert int(open(tmp_path / ("c_%d.txt" % r)).read()) == c
        got = np.load(tmp_path / ("merged_%d.npy" % r))
        assert np.array_equal(got, want)


def test_shard_range_partitions():
    for n in (0, 1, 7, 64, 1000003):
        for w in (1, 2, 3, 8):
            r = [fdist.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1
