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


class _StandInEngine:
    """An engine for a box without GPUs: the oracle behind the calls bench.py makes (test infrastructure -- the product
    engine has no CPU path).  The counter buffer is a CPU tensor, so the all-reduce under test really moves it."""

    def __init__(self, opt, ad_start, ad_end, ad_fasta, C):
        from oracle import oracle

        self.orc = oracle
        self.cfg = oracle.Config(opt, ad_start, ad_end, ad_fasta)
        self.C = C
        self.n_adapters = self.cfg.n_adapters
        self.cnt = torch.zeros(abi.counters_len(C, self.n_adapters), dtype=torch.int64)
        self.calls = 0

    def process_device(self, seq_t, qual_t, off_t, max_len, res_t=None, stream=None):
        r, c = self.orc.process_batch(self.cfg, seq_t.numpy(), qual_t.numpy(), off_t.numpy().astype(np.uint64), max_cycles=self.C)
        self.cnt += torch.from_numpy(c)
        self.calls += 1
        out = torch.from_numpy(np.ascontiguousarray(r).view(np.uint8).copy())
        if res_t is not None:  # (the caller's record buffer, as the C-ABI fills it)
            res_t[:out.numel()].copy_(out)
            return res_t
        return out

    @staticmethod
    def results_to_numpy(results_t, n):
        return results_t.numpy().view(abi.RESULT_DTYPE)[:n]

    def reset_counters(self):
        self.cnt.zero_()

    def enable_timing(self, on):
        self.calls = 0

    def kernel_times(self):
        return {"k_trim_ends": 1.0 * self.calls, "k_scan": 2.0 * self.calls, "k_resolve": 0.1 * self.calls, "k_stats_prep": 0.1 * self.calls, "k_stats": 1.5 * self.calls}, self.calls

    def counters_tensor(self):
        return self.cnt

    def counters(self):
        return self.cnt.numpy().copy()

    def close(self):
        pass


def _stand_in_rig():
    sys.path.insert(0, ROOT)
    import bench

    class Rig(bench.Rig):
        backend = "gloo"

        def device(self, local_rank):
            return torch.device("cpu")

        def synchronize(self, dev):
            pass

        def stream(self, dev):
            return 0

        def make_batch(self, wl, n_reads, rank_, dev):
            # every rank its own reads (seeded by the rank, as bench.make_batch does), of different maximum length
            seq, qual, off = synth.ont_like(n_reads, seed=1 + rank_, median_len=500 + 300 * rank_, p_middle=0.1)
            return (torch.from_numpy(seq), torch.from_numpy(qual), torch.from_numpy(off.astype(np.int64)),
                    int(np.diff(off.astype(np.int64)).max()), synth.START_ADAPTER, synth.END_ADAPTER, [])

        def engine(self, opt, ad_start, ad_end, ad_fasta, local_rank, C):
            return _StandInEngine(opt, ad_start, ad_end, ad_fasta, C)

    return bench, Rig()


def _bench_rank(rank, world, port, outdir):
    bench, rig = _stand_in_rig()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import contextlib
    import io

    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--reads", "25", "--cpu-bases", "0",
                    "--e2e-reads", "0", "--full-json", os.path.join(outdir, "full_%d.json" % rank)], rig=rig)
    with open(os.path.join(outdir, "out_%d.txt" % rank), "w") as f:
        f.write(buf.getvalue())


def test_bench_rank_code_path_gloo(orc, tmp_path):
    """bench.py's own N > 1 code -- per-rank shard, agree_capacity, the timed region with its barriers, the counter
    all-reduce, the MAX / SUM reductions behind the JSON line -- with world_size 2 over gloo on CPU tensors"""
    import json

    world, steps, reads = 2, 3, 25
    mp.spawn(_bench_rank, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert open(tmp_path / "out_1.txt").read().strip() == ""  # only rank 0 prints
    line = json.loads(open(tmp_path / "out_0.txt").read().strip())
    assert line["n_gpus"] == 2 and line["steps"] == steps and line["scaling"] == "weak" and line["unit"] == "Gbases/s"
    # what the line must add up to: both ranks' reads, `steps` times, through one all-reduce
    cfg_opt = abi.FplOptions.default(cut_front=1, cut_tail=1, cut_front_window=5, cut_tail_window=5, polyx=1,
                                     complexity_filter=1)
    tot_reads = tot_bases = tot_out = 0
    C = 0
    shards = []
    for r in range(world):
        seq, qual, off = synth.ont_like(reads, seed=1 + r, median_len=500 + 300 * r, p_middle=0.1)
        shards.append((seq, qual, off))
        C = max(C, int(np.diff(off.astype(np.int64)).max()))
    for seq, qual, off in shards:
        _, cnt = orc.process_batch(orc.Config(cfg_opt, synth.START_ADAPTER, synth.END_ADAPTER), seq, qual, off, max_cycles=C)
        v = abi.CountersView(cnt, C, 2)
        tot_reads += int(v.pre.reads)
        tot_bases += int(v.pre.length_sum)
        tot_out += int(v.post.reads)
    cc = line["counters_check"]
    assert cc["reads_in"] == steps * tot_reads == steps * world * reads
    assert cc["bases_in"] == steps * tot_bases and cc["fragments_out"] == steps * tot_out
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 * 1e9 - tot_bases) < 1e-6 * tot_bases  # value = all ranks' bases / time
    assert line["roofline"]["kernel"] == "k_scan" and line["config"]["reads_per_gpu"] == reads
    # the checks bench.py makes on its own line: the counters add up to steps x reads, and the first reads of rank 0's batch
    # went through a fresh engine and were compared with the oracle record by record
    assert cc["ok"] is True and cc["expected_reads_in"] == steps * world * reads
    assert line["parity_sample"] == "ok", line["parity_sample"]
    assert line["roofline"]["path"]["frac"] > 0 and line["roofline"]["bound"] == "hbm"


def test_bench_line_is_one_short_json_line(orc, tmp_path, monkeypatch, capsys):
    """The driver parses the LAST stdout line and keeps 8 KB of stdout: bench.py's line must stay a few KB whatever the
    end-to-end leg collected (round 5's 21.6 KB line was lost).  bench.main at N = 1 with the stand-in engine, then the
    compact form of an object carrying a round-5-sized `e2e`."""
    import json

    bench, rig = _stand_in_rig()
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    full = tmp_path / "full.json"
    bench.main(["--steps", "2", "--warmup", "1", "--reads", "30", "--cpu-bases", "1", "--e2e-reads", "0", "--full-json", str(full)], rig=rig)
    cap = capsys.readouterr()
    lines = [l for l in cap.out.splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < bench.LINE_LIMIT
    line = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "cpu_baseline",
              "counters_check", "parity_sample"):
        assert k in line, k
    assert line["roofline"]["frac"] > 0 and line["roofline"]["bound"] == "hbm" and "kernel_ms" in line["roofline"]
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["reference_buildable"] is False and len(cb["blocked_by"]) == 2 and cb["value"] > 0
    whole = json.loads(open(full).read())
    assert whole["value"] == line["value"] and line["full"] == str(full)
    # a round-5-sized end-to-end object must not lengthen the line beyond the limit
    stage = ["host pipeline: 1622 batches, wall 8.35295 s; busy: parse 0.98072 s " + "x" * 400] * 6
    run = lambda v: {"rc": 0, "process_seconds": 1.0, "value": v, "bases": 10, "pipeline_seconds": 0.5, "pipeline_value": 2 * v, "stages": stage,
                     "what": "y" * 300}
    names = [r["name"] for r in bench.E2E_RUNS]
    whole["e2e"] = {"reads": 1000000, "bases": 9 * 10**9, "n_gpus": 1, "value": 15.4, "pcie_call": {"value": 27.0, "what": "z" * 200},
                    "cli": {n: run(10.0 + i) for i, n in enumerate(names)}, "json_check": {"ok": True},
                    "large_input": {"reads": 3000000, "cli": {n: run(20.0 + i) for i, n in enumerate(names)}}}
    assert len(json.dumps(whole)) > 20000
    bench.emit(whole, str(tmp_path / "full2.json"))
    out2 = [l for l in capsys.readouterr().out.splitlines() if l.strip()]
    assert len(out2) == 1 and len(out2[0]) < bench.LINE_LIMIT
    l2 = json.loads(out2[0])
    assert l2["e2e"]["value"] == 15.4 and l2["e2e"]["to_file"] == 10.0 + names.index("to_file") and l2["e2e"]["large"]["null8"] is not None
    assert l2["roofline"] == line["roofline"] and l2["cpu_baseline"] == line["cpu_baseline"]
