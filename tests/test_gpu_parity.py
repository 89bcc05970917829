"""Parity tests proper: the HIP path through the C-ABI (libfastplong_amd.so) against the oracle
on the same seeded inputs, bit for bit (integer / byte / index work => exact equality), plus
size-independent properties at larger sizes.  Need a real MI355X: run with -m gpu."""
import os

import numpy as np
import pytest

from fastplong_amd import abi, synth
from tests import parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine_mod():
    import torch

    assert torch.cuda.is_available(), "the -m gpu tests need a GPU"
    from fastplong_amd import engine

    engine.load_library()
    return engine


CASES = {
    "defaults_adapters": dict(opt=dict(), start=synth.START_ADAPTER, end=synth.END_ADAPTER),
    "full_pipeline": dict(opt=dict(cut_front=1, cut_tail=1, cut_front_window=5, cut_tail_window=5, polyx=1,
                                   complexity_filter=1), start=synth.START_ADAPTER, end=synth.END_ADAPTER),
    "no_adapter_trimming": dict(opt=dict(adapter_enabled=0), start="", end=""),
    "quality_filter_only": dict(opt=dict(adapter_enabled=0, length_filter=0), start="", end=""),
    "nasty_options": dict(opt=dict(trim_front=3, trim_tail=2, cut_front=1, cut_tail=1, cut_front_window=7,
                                   cut_front_quality=15, cut_tail_window=3, cut_tail_quality=25, polyx=1,
                                   polyx_min_len=8, complexity_filter=1, complexity_percent=40,
                                   qualified_qual=33 + 20, unqualified_percent_limit=30, n_base_percent_limit=5,
                                   avg_qual_req=12, required_length=30, max_length=350, ed_max=0.3,
                                   trimming_extension=5), start=synth.START_ADAPTER, end=synth.END_ADAPTER),
    "auto_literal": dict(opt=dict(), start="auto", end="auto"),  # undetected "auto" is used literally
    "long_window": dict(opt=dict(cut_front=1, cut_tail=1, cut_front_window=50, cut_tail_window=1000),
                        start=synth.START_ADAPTER, end=synth.END_ADAPTER),
}


def _run_both(orc, engine_mod, cfgd, seq, qual, off, fasta=(), via="host"):
    import torch

    cfg = orc.Config(abi.FplOptions.default(**cfgd["opt"]), cfgd["start"], cfgd["end"], fasta)
    n = len(off) - 1
    C = max(1, int(np.diff(off.astype(np.int64)).max()) if n else 1)
    want_res, want_cnt = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
    eng = engine_mod.Engine(cfg.opt, cfgd["start"], cfgd["end"], fasta, device=0, max_cycles=C)
    if via == "host":
        got_res = eng.process_host(seq, qual, off)
    else:
        st = torch.from_numpy(seq.copy()).cuda()
        qt = torch.from_numpy(qual.copy()).cuda()
        ot = torch.from_numpy(off.astype(np.int64)).cuda()
        rt = eng.process_device(st, qt, ot, C)
        torch.cuda.synchronize()
        got_res = eng.results_to_numpy(rt, n)
    got_cnt = eng.counters()
    assert eng.max_cycles == C
    eng.close()
    parity.assert_results_equal(got_res, want_res, seq, off)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)
    return want_res, want_cnt


@pytest.mark.parametrize("name", sorted(CASES))
def test_adversarial_reads_bit_exact(orc, engine_mod, name):
    seq, qual, off = synth.adversarial(3000, seed=100 + len(name))
    _run_both(orc, engine_mod, CASES[name], seq, qual, off)


@pytest.mark.parametrize("la,lb", [(64, 65), (65, 64), (100, 150), (250, 33), (8, 12), (15, 16), (4, 32), (16, 0), (0, 31), (24, 7), (7, 6), (5, 4),
                                   (40, 50), (33, 64), (64, 20), (45, 45), (32, 16), (17, 31), (29, 32)])
def test_odd_command_line_adapter_lengths_bit_exact(orc, engine_mod, la, lb):
    """-s / -e adapters beyond 64 bases leave the LDS tables of k_trim_ends (one 64-column Peq word per byte value) and
    take the global-memory path; beyond 200 bases they are longer than the end window; below 16 bases (partial
    pattern shorter than 16 columns) or beyond 32 the kernels' SHORT instantiations do not apply"""
    rng = np.random.default_rng(la * 1000 + lb)
    start = "".join("ACGT"[i] for i in rng.integers(0, 4, la))
    end = "".join("ACGT"[i] for i in rng.integers(0, 4, lb))
    seq, qual, off = synth.adversarial(1500, seed=la + lb, start_adapter=start, end_adapter=end)
    _run_both(orc, engine_mod, dict(opt=dict(cut_front=1, polyx=1), start=start, end=end), seq, qual, off)


@pytest.mark.parametrize("name", ["defaults_adapters", "full_pipeline"])
def test_ont_like_reads_bit_exact(orc, engine_mod, name):
    # lengths up to tens of kb: several cycle tiles, several scan tiles per read
    seq, qual, off = synth.ont_like(400, seed=5, median_len=6000, p_middle=0.05)
    res, _ = _run_both(orc, engine_mod, CASES[name], seq, qual, off, via="device")
    assert (res["n_frag"] == 2).any()


@pytest.mark.parametrize("name", sorted(CASES))
def test_wave_per_read_trim_kernel_bit_exact(orc, engine_mod, monkeypatch, name):
    """batches of fewer than 65 536 reads take k_trim_ends<1> (a wave per read) instead of k_trim_ends_batched; the test suite
    forces the batched kernel everywhere else (tests/conftest.py)"""
    monkeypatch.delenv("FPL_TRIM_BATCH_MIN", raising=False)
    a = synth.adversarial(1500, seed=300 + len(name))
    b = synth.ont_like(300, seed=7, median_len=3000, p_middle=0.05)
    reads = []
    for (s_, q_, o_) in (a, b):
        reads += [(s_[int(o_[i]):int(o_[i + 1])], q_[int(o_[i]):int(o_[i + 1])]) for i in range(len(o_) - 1)]
    seq, qual, off = synth.pack(reads)
    _run_both(orc, engine_mod, CASES[name], seq, qual, off)


@pytest.mark.parametrize("min_bucket,per", [(1, 0), (8, 128), (0, 0)])
def test_sorted_statistics_pass_bit_exact(orc, engine_mod, monkeypatch, min_bucket, per):
    """k_stats_sorted is what batches of >= 150 000 reads take (test_large_batch_properties runs one); here it is forced on
    a batch the oracle can check in full -- every front trim in slices of its own, several slices per bucket with the rare
    front trims handed to the EXTRA pass, and (0, 0) the built-in thresholds with the size limit lifted"""
    if min_bucket:
        monkeypatch.setenv("FPL_STATS_MIN_BUCKET", str(min_bucket))
    else:
        monkeypatch.setenv("FPL_STATS_SORT_MIN", "1")
    if per:
        monkeypatch.setenv("FPL_STATS_PER", str(per))
    a = synth.ont_like(1500, seed=41, median_len=2500, p_middle=0.05)
    b = synth.adversarial(1500, seed=42)
    reads = []
    for (s_, q_, o_) in (a, b):
        reads += [(s_[int(o_[i]):int(o_[i + 1])], q_[int(o_[i]):int(o_[i + 1])]) for i in range(len(o_) - 1)]
    seq, qual, off = synth.pack(reads)
    _run_both(orc, engine_mod, CASES["full_pipeline"], seq, qual, off, via="device")


@pytest.mark.parametrize("hi_tile,group,rows,per", [(1, 3, 0, 64), (1, 16, 70, 64), (2, 2, 0, 128), (3, 5, 64, 64), (4, 16, 0, 0)])
def test_sorted_statistics_pass_slice_groups_bit_exact(orc, engine_mod, monkeypatch, hi_tile, group, rows, per):
    """k_stats_sorted: from cycle tile hi_tile - 1 on an item is a group of consecutive slices of one front trim -- one table set,
    one slab, for as long as the rows counted so far fit (FPL_STATS_GROUP_ROWS: a small limit hands groups over in pieces)"""
    monkeypatch.setenv("FPL_STATS_MIN_BUCKET", "2")
    if per:
        monkeypatch.setenv("FPL_STATS_PER", str(per))
    monkeypatch.setenv("FPL_STATS_HI_TILE", str(hi_tile))
    monkeypatch.setenv("FPL_STATS_GROUP", str(group))
    if rows:
        monkeypatch.setenv("FPL_STATS_GROUP_ROWS", str(rows))
    a = synth.ont_like(2500, seed=43, median_len=1500, sigma_len=0.8, min_len=300, max_len=20000, p_middle=0.05)
    b = synth.adversarial(1000, seed=44)
    reads = []
    for (s_, q_, o_) in (a, b):
        reads += [(s_[int(o_[i]):int(o_[i + 1])], q_[int(o_[i]):int(o_[i + 1])]) for i in range(len(o_) - 1)]
    seq, qual, off = synth.pack(reads)
    _run_both(orc, engine_mod, CASES["full_pipeline"], seq, qual, off, via="device")


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("case", ["defaults_adapters", "full_pipeline"])
def test_sorted_statistics_pass_row_setup_bit_exact(orc, engine_mod, monkeypatch, seed, case):
    """k_stats_sorted's row set-up (tests/test_kernels_emu.py::_reads_for_row_setup): odd bytes and U's around the lane and tile
    boundaries of the 5-mer stream, reads ending right at them, clean reads"""
    from tests.test_kernels_emu import _reads_for_row_setup
    monkeypatch.setenv("FPL_STATS_MIN_BUCKET", "1")
    monkeypatch.setenv("FPL_STATS_PER", "128")
    seq, qual, off = _reads_for_row_setup(seed, n=3000)
    _run_both(orc, engine_mod, CASES[case], seq, qual, off, via="device")


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("case", ["defaults_adapters", "full_pipeline"])
def test_sorted_statistics_pass_kmer6_bit_exact(orc, engine_mod, monkeypatch, seed, case):
    """k_stats_sorted's 6-mer table (tests/test_kernels_emu.py::_reads_for_kmer6): N's at every offset of a lane's eight bytes,
    bytes of the base classes 0 and 2 in every kind of row -- the shipped instructions (ds_add_u32 on the 6-mer table, the global
    atomics of the byte-by-byte rows)"""
    from tests.test_kernels_emu import _reads_for_kmer6
    monkeypatch.setenv("FPL_STATS_MIN_BUCKET", "1")
    if seed == 2:
        monkeypatch.setenv("FPL_STATS_PER", "128")
    seq, qual, off = _reads_for_kmer6(seed, n=4000)
    _run_both(orc, engine_mod, CASES[case], seq, qual, off, via="device")


def test_multi_adapter_fasta_bit_exact(orc, engine_mod):
    fasta = ["ACGTTGCAATGCCGTA", "TTGACCAGTAGGCATCAGGATCCA", "GATTACA",
             "CCCCGGGGAAAATTTTCCCCGGGGAAAATTTTCCCCGGGGAAAATTTTCCCCGGGGAAAATTTTCCCCGGGG",
             "".join("ACGT"[(i * 7 + i // 3) % 4] for i in range(200))]
    seq, qual, off = synth.adversarial(2000, seed=33, fasta=fasta)
    _run_both(orc, engine_mod, CASES["defaults_adapters"], seq, qual, off, fasta=fasta)


def test_hifi_like_64_adapters_bit_exact(orc, engine_mod):
    seq, qual, off, ads = synth.hifi_like(60, seed=5, mean_len=9000, sd_len=1500, n_adapters=64)
    cfgd = dict(opt=dict(), start=ads[0], end=synth.revcomp(ads[0]))
    _run_both(orc, engine_mod, cfgd, seq, qual, off, fasta=sorted(ads))


@pytest.mark.parametrize("lens,ed_max,n_ad,seed", [((30, 45), 0.25, 64, 1), ((23, 24, 31, 32, 33, 64), 0.25, 70, 2), ((23, 40, 64), 0.4, 9, 3),
                                                   ((32, 64), 1.0, 5, 4), ((22, 23, 30), 0.25, 9, 5)])
def test_fasta_filter_packed_scores_bit_exact(orc, engine_mod, lens, ed_max, n_ad, seed):
    """k_trim_ends<2>: adapter sets whose every adapter has 23 bases or more take the packed-score form of the lane-per-adapter
    filter (fasta_may_trim32p; tests/test_kernels_emu.py holds it against the plain form verdict by verdict).  ed_max 1.0: the
    thresholds leave the packed fields' range (plain form); the set with a 22-mer: plain form as a whole; 70 adapters: two groups"""
    rng = np.random.default_rng(1000 + seed)
    rnd = lambda n: "".join("ACGT"[i] for i in rng.integers(0, 4, int(n)))  # noqa: E731
    fasta = [rnd(rng.integers(lens[0], lens[-1] + 1) if len(lens) == 2 else rng.choice(lens)) for _ in range(n_ad)]
    start, end = fasta[0], synth.revcomp(fasta[0])
    seq, qual, off = synth.adversarial(300, seed=40 + seed, start_adapter=start, end_adapter=end, fasta=fasta)
    reads = [(seq[int(off[i]):int(off[i + 1])], qual[int(off[i]):int(off[i + 1])]) for i in range(len(off) - 1)]
    for k in range(240):  # an adapter (whole, or its partial pattern and a few bases more) with up to three errors at an end of a random read
        ad = fasta[int(rng.integers(0, len(fasta)))]
        body = rnd(rng.integers(40, 2000))
        piece = list(ad if k % 3 else (ad[-(16 + k % 9):] if k % 2 else ad[:16 + k % 9]))
        for _ in range(k % 4):
            piece[int(rng.integers(0, len(piece)))] = "ACGT"[int(rng.integers(0, 4))]
        piece = "".join(piece)
        r = np.frombuffer(((piece + body) if k % 2 else (body + piece)).encode(), np.uint8)
        reads.append((r, np.full(len(r), 70, np.uint8)))
    seq, qual, off = synth.pack(reads)
    res, cnt = _run_both(orc, engine_mod, dict(opt=dict(ed_max=ed_max, trimming_extension=5), start=start, end=end), seq, qual, off, fasta=fasta)
    assert (res["r1_start"] > 0).any()


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("FPL_FUZZ_FASTA_FROM", "0")),
                                            int(os.environ.get("FPL_FUZZ_FASTA_FROM", "0")) + int(os.environ.get("FPL_FUZZ_FASTA", "8")))))
def test_random_fasta_sets_of_16_to_64_mers_bit_exact(orc, engine_mod, seed):
    """adapter sets made of 16..64-base ACGT adapters only (k_trim_ends<2>: the lane-per-adapter filter in front of the exact
    trims, fasta_may_trim) -- 1, 3, 64, 65 and 130 FASTA adapters (one, two and three groups of 64 lanes), mutated /
    truncated copies near both ends, every ed_max; FPL_FUZZ_FASTA=<n> widens it for a soak"""
    rng = np.random.default_rng(52000 + seed)
    rnd = lambda n: "".join("ACGT"[i] for i in rng.integers(0, 4, int(n)))  # noqa: E731
    n_fa = int(rng.choice([1, 3, 64, 65, 130]))
    fasta = [rnd(rng.choice([16, 17, 24, 31, 32, 33, 40, 48, 63, 64])) for _ in range(n_fa)]
    if rng.random() < 0.5:  # near-duplicates: several adapters of a group can trim the same end, the first in list order wins
        fasta += [f[:int(rng.integers(16, len(f) + 1))] for f in fasta[:4]]
    start, end = rnd(rng.choice([16, 24, 32, 33, 64])), rnd(rng.choice([16, 27, 32, 45, 64]))
    opt = dict(ed_max=float(rng.choice([0.0, 0.1, 0.25, 0.4])), trimming_extension=int(rng.choice([0, 10, 30])),
               cut_front=int(rng.integers(2)), polyx=int(rng.integers(2)))
    a = synth.adversarial(400, seed=seed, start_adapter=start, end_adapter=end, fasta=fasta)
    b = synth.hifi_like(12, seed=seed, mean_len=3000, sd_len=800, n_adapters=4)[:3]
    reads = []
    for (s_, q_, o_) in (a, b):
        reads += [(s_[int(o_[i]):int(o_[i + 1])], q_[int(o_[i]):int(o_[i + 1])]) for i in range(len(o_) - 1)]
    seq, qual, off = synth.pack(reads)
    res, cnt = _run_both(orc, engine_mod, dict(opt=opt, start=start, end=end), seq, qual, off, fasta=fasta)
    assert (res["r1_start"] > 0).any()


def _lowq_ont_like(n, seed, median_len=3000):
    """ONT-like reads whose qualities dip far below the --break / --mask thresholds in stretches"""
    rng = np.random.default_rng(seed)
    seq, qual, off = synth.ont_like(n, seed=seed, median_len=median_len, p_middle=0.2, p_polya=0.1)
    qual = qual.copy()
    for i in range(n):
        a, b = int(off[i]), int(off[i + 1])
        pos = a + int(rng.integers(0, max(1, (b - a) // 2)))
        while pos < b:
            run = int(rng.integers(10, 400))
            if rng.random() < 0.4:
                qual[pos:min(b, pos + run)] = np.clip(np.round(rng.normal(6, 3, min(b, pos + run) - pos)), 2, 40) + 33
            pos += run + int(rng.integers(50, 2000))
    return seq, qual, off


@pytest.mark.parametrize("be,me,bw,mw", [(1, 0, 100, 50), (0, 1, 100, 50), (1, 1, 100, 50), (1, 1, 5, 7), (1, 1, 1000, 300)])
def test_break_and_mask_bit_exact(orc, engine_mod, be, me, bw, mw):
    """--break / --mask (src/seprocessor.cpp:234-262): fragment records, N regions, FilterResult and both Stats"""
    opt = abi.FplOptions.default(cut_front=1, cut_tail=1, polyx=1, complexity_filter=1, break_enabled=be, break_window=bw,
                                 break_quality=12, mask_enabled=me, mask_window=mw, mask_quality=13,
                                 n_base_percent_limit=95, unqualified_percent_limit=90, complexity_percent=5)
    cfg = orc.Config(opt, synth.START_ADAPTER, synth.END_ADAPTER)
    seq, qual, off = _lowq_ont_like(300, seed=41 + bw)
    C = int(np.diff(off.astype(np.int64)).max())
    want_res, want_cnt, want_f, want_r = orc.process_batch_ex(cfg, seq, qual, off, max_cycles=C)
    eng = engine_mod.Engine(cfg.opt, synth.START_ADAPTER, synth.END_ADAPTER, device=0, max_cycles=C)
    got_res = eng.process_host(seq, qual, off)
    got_f, got_r = eng.fragments()
    got_cnt = eng.counters()
    # a second, smaller batch through the same context: the lists describe the LAST batch only
    eng.process_host(seq[:int(off[50])], qual[:int(off[50])], off[:51])
    f2, r2 = eng.fragments()
    eng.close()
    parity.assert_fragments_equal(got_f, got_r, want_f, want_r)
    parity.assert_results_equal(got_res, want_res, seq, off)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)
    keep = want_f["read"] < 50
    parity.assert_fragments_equal(f2, r2, want_f[keep], want_r)
    if bw == 100:  # (the other window sizes are there for the search arithmetic, not for a particular mix)
        assert (want_f["break_no"] > 0).any() == bool(be) and (want_f["region_count"] > 0).any() == bool(me)
        assert (want_f["code"] == abi.FPL_PASS_FILTER).any()


def random_case(seed):
    """(option dict, start adapter, end adapter, seq, qual, off) of one seeded corner of the option space"""
    rng = np.random.default_rng(1000 + seed)
    pick = lambda *v: v[int(rng.integers(len(v)))]  # noqa: E731
    okw = dict(
        trim_front=pick(0, 0, 1, 7, 40), trim_tail=pick(0, 0, 2, 9, 33), cut_front=pick(0, 1), cut_tail=pick(0, 1),
        cut_front_window=pick(1, 4, 5, 30, 300), cut_front_quality=pick(5, 15, 20, 30), cut_tail_window=pick(1, 4, 7, 64, 500),
        cut_tail_quality=pick(5, 15, 20, 30), polyx=pick(0, 1), polyx_min_len=pick(3, 8, 10, 25),
        adapter_enabled=pick(0, 1, 1, 1), ed_max=pick(0.0, 0.1, 0.25, 0.4), trimming_extension=pick(0, 5, 10, 30),
        qual_filter=pick(0, 1, 1), qualified_qual=33 + pick(5, 15, 20, 30), unqualified_percent_limit=pick(0, 20, 40, 90),
        n_base_limit=pick(0, 3, 1000000), n_base_percent_limit=pick(0, 5, 10, 95), avg_qual_req=pick(0, 0, 10, 20),
        length_filter=pick(0, 1, 1), required_length=pick(1, 20, 100, 400), max_length=pick(0, 0, 300, 2000),
        complexity_filter=pick(0, 1), complexity_percent=pick(0, 5, 30, 60, 100),
        break_enabled=pick(0, 0, 1), break_window=pick(1, 5, 40, 100, 700), break_quality=pick(5, 10, 15, 30),
        mask_enabled=pick(0, 0, 1), mask_window=pick(1, 5, 15, 50, 400), mask_quality=pick(5, 10, 15, 30))
    start = pick(synth.START_ADAPTER, synth.START_ADAPTER, "", "ACGTNACGTAGGCATCGATCGGCTA")
    end = pick(synth.END_ADAPTER, synth.END_ADAPTER, "", synth.revcomp(synth.START_ADAPTER)[:18])
    a = synth.adversarial(120, seed=seed, start_adapter=start or synth.START_ADAPTER, end_adapter=end or synth.END_ADAPTER)
    b = _lowq_ont_like(60, seed=seed, median_len=int(os.environ.get("FPL_FUZZ_MEDIAN", "1500")))  # (soaks: longer reads)
    reads = []
    for (s_, q_, o_) in (a, b):
        reads += [(s_[int(o_[i]):int(o_[i + 1])], q_[int(o_[i]):int(o_[i + 1])]) for i in range(len(o_) - 1)]
    seq, qual, off = synth.pack(reads)
    return okw, start, end, seq, qual, off


# 13107: a --mask'ed read whose unmasked piece starts three bases in front of a cycle-tile boundary (found by a 20 000-seed soak)
@pytest.mark.parametrize("seed", list(range(int(os.environ.get("FPL_FUZZ_FROM", "0")),
                                            int(os.environ.get("FPL_FUZZ_FROM", "0")) + int(os.environ.get("FPL_FUZZ_SEEDS", "32")))) + [13107])
def test_random_option_sets_bit_exact(orc, engine_mod, seed):
    """seeded random corners of the option space (including --break / --mask) on adversarial + ONT-like reads"""
    okw, start, end, seq, qual, off = random_case(seed)
    cfg = orc.Config(abi.FplOptions.default(**okw), start, end)
    C = max(1, int(np.diff(off.astype(np.int64)).max()))
    eng = engine_mod.Engine(cfg.opt, start, end, device=0, max_cycles=C)
    got_res = eng.process_host(seq, qual, off)
    got_cnt = eng.counters()
    if okw["break_enabled"] or okw["mask_enabled"]:
        want_res, want_cnt, want_f, want_r = orc.process_batch_ex(cfg, seq, qual, off, max_cycles=C)
        got_f, got_r = eng.fragments()
        parity.assert_fragments_equal(got_f, got_r, want_f, want_r)
    else:
        want_res, want_cnt = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
    eng.close()
    parity.assert_results_equal(got_res, want_res, seq, off)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("FPL_FUZZ_AHEAD_FROM", "500")),
                                            int(os.environ.get("FPL_FUZZ_AHEAD_FROM", "500")) + int(os.environ.get("FPL_FUZZ_AHEAD_SEEDS", "12")))))
def test_random_option_sets_back_to_back_with_trims_ahead(orc, engine_mod, monkeypatch, seed):
    """the same seeded corners of the option space, three resident batches (case, a second case's reads, the case again) enqueued
    back to back under fpl_assume_inputs_ready: the end trims of batch k + 1 run beside the kernels of batch k (the lane-per-read
    trims and the sorted statistics pass, by the size hooks).  Every record of every batch and the accumulated counters against the
    oracle.  (--break / --mask batches never run ahead: they pass through here as in-line batches.)"""
    import torch

    for k, v in (("FPL_TRIM_BATCH_MIN", "1"), ("FPL_STATS_SORT_MIN", "1")):
        if k not in os.environ:
            monkeypatch.setenv(k, v)
    okw, start, end, seq, qual, off = random_case(seed)
    _, _, _, seq2, qual2, off2 = random_case(seed + 7919)
    batches = [(seq, qual, off), (seq2, qual2, off2), (seq, qual, off)]
    cfg = orc.Config(abi.FplOptions.default(**okw), start, end)
    C = max(1, max(int(np.diff(o.astype(np.int64)).max()) for _, _, o in batches))
    eng = engine_mod.Engine(cfg.opt, start, end, device=0, max_cycles=C)
    eng.assume_inputs_ready(True)
    dev = [(torch.from_numpy(s_).cuda(), torch.from_numpy(q_).cuda(), torch.from_numpy(o_.astype(np.int64)).cuda(), len(o_) - 1)
           for s_, q_, o_ in batches]
    torch.cuda.synchronize()
    res = [eng.process_device(st, qt, ot, C) for st, qt, ot, _ in dev]
    torch.cuda.synchronize()
    got = [eng.results_to_numpy(r, b[3]) for r, b in zip(res, dev)]
    got_cnt = eng.counters()
    deferred = okw["break_enabled"] or okw["mask_enabled"]
    assert eng.batch_forms()["trims_ahead"] == (0 if deferred else 2)
    eng.close()
    want_cnt = np.zeros(abi.counters_len(C, cfg.n_adapters), np.int64)
    for (s_, q_, o_), g in zip(batches, got):
        if deferred:
            want_res, cnt1 = orc.process_batch_ex(cfg, s_, q_, o_, max_cycles=C)[:2]
            want_cnt += cnt1
        else:
            want_res, _ = orc.process_batch(cfg, s_, q_, o_, max_cycles=C, counters=want_cnt)
        parity.assert_results_equal(g, want_res, s_, o_)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)


def test_very_long_reads_bit_exact(orc, engine_mod):
    """BASELINE configs[3] goes up to 200 kb per read, ultra-long ONT reads beyond a megabase: thousands of cycle
    tiles, long histories in every kernel"""
    rng = np.random.default_rng(77)
    reads = []
    for L in (1200000, 262144, 200000, 131073, 65536, 65535, 9000, 50):  # (ultra-long ONT reads exceed a megabase)
        s, q, o = synth.ont_like(1, seed=int(L) % 1000, median_len=L, sigma_len=0.0, min_len=L, max_len=L, p_middle=1.0)
        reads.append((s[:int(o[1])], q[:int(o[1])]))
    for _ in range(40):
        s, q, o = synth.ont_like(1, seed=int(rng.integers(1 << 30)), median_len=5000)
        reads.append((s[:int(o[1])], q[:int(o[1])]))
    seq, qual, off = synth.pack(reads)
    _run_both(orc, engine_mod, CASES["full_pipeline"], seq, qual, off, via="device")


def test_create_rejects_bad_break_mask_window(engine_mod):
    for kw in (dict(break_enabled=1, break_window=0), dict(mask_enabled=1, mask_window=-3)):
        with pytest.raises(engine_mod.FplError):
            engine_mod.Engine(abi.FplOptions.default(**kw), synth.START_ADAPTER, synth.END_ADAPTER, device=0, max_cycles=64)


def test_edge_batches(orc, engine_mod):
    cfgd = CASES["full_pipeline"]
    # empty batch
    eng = engine_mod.Engine(abi.FplOptions.default(**cfgd["opt"]), cfgd["start"], cfgd["end"], device=0, max_cycles=4)
    r = eng.process_host(np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert len(r) == 0 and not eng.counters().any()
    eng.close()
    # empty reads, 1-base reads, a read of only N, only polyA, one long read among short ones
    reads = [b"", b"A", b"N" * 50, b"A" * 300, b"ACGT" * 5000, b"", b"acgtn" * 20, b"G" * 16, b"T" * 15]
    rng = np.random.default_rng(1)
    seq = np.frombuffer(b"".join(reads), np.uint8).copy()
    qual = (33 + rng.integers(2, 45, len(seq))).astype(np.uint8)
    off = np.zeros(len(reads) + 1, np.uint64)
    off[1:] = np.cumsum([len(r) for r in reads])
    for name in ("defaults_adapters", "full_pipeline", "nasty_options"):
        _run_both(orc, engine_mod, CASES[name], seq, qual, off)


def test_capacity_growth_and_accumulation(orc, engine_mod):
    """counters accumulate across batches; the per-cycle capacity grows on demand"""
    cfgd = CASES["full_pipeline"]
    cfg = orc.Config(abi.FplOptions.default(**cfgd["opt"]), cfgd["start"], cfgd["end"])
    b1 = synth.adversarial(500, seed=1)
    b2 = synth.ont_like(60, seed=2, median_len=3000)
    c2 = int(np.diff(b2[2].astype(np.int64)).max())
    eng = engine_mod.Engine(cfg.opt, cfgd["start"], cfgd["end"], device=0, max_cycles=16)
    eng.process_host(*b1)
    eng.process_host(*b2)
    got = eng.counters()
    C = eng.max_cycles
    assert C >= c2
    eng.close()
    want = np.zeros(abi.counters_len(C, 2), np.int64)
    orc.process_batch(cfg, *b1, max_cycles=C, counters=want)
    orc.process_batch(cfg, *b2, max_cycles=C, counters=want)
    parity.assert_counters_equal(got, want, C, 2)


def test_large_batch_properties(engine_mod, monkeypatch):
    """BASELINE-sized reads without the oracle: size-independent properties.
    (1) conservation: every read is dropped or yields fragments inside r1 inside the read;
    (2) idempotence of the statistics: post-filter Stats of a run == pre-filter Stats of a run
        over exactly the passing fragments with trimming/filters off;
    (3) partition invariance: counters of one batch == sum of counters of its two halves.
    The whole batch goes through the statistics pass over sorted reads (k_stats_sorted, with the built-in bucket threshold;
    batches of >= 150 000 reads take it by themselves), the halves and the fragment batch through the unsorted pass: (2)
    and (3) also hold the two implementations against each other."""
    import torch

    monkeypatch.setenv("FPL_STATS_SORT_MIN", "1")

    opt = abi.FplOptions.default(cut_front=1, cut_tail=1, cut_front_window=5, cut_tail_window=5, polyx=1,
                                 complexity_filter=1)
    seq_t, qual_t, off_t, max_len = synth.device_batch(20000, seed=9, median_len=8000)
    n = off_t.numel() - 1
    eng = engine_mod.Engine(opt, synth.START_ADAPTER, synth.END_ADAPTER, device=0, max_cycles=max_len)
    rt = eng.process_device(seq_t, qual_t, off_t, max_len)
    torch.cuda.synchronize()
    res = eng.results_to_numpy(rt, n)
    cnt = eng.counters()
    C = eng.max_cycles
    eng.close()
    monkeypatch.delenv("FPL_STATS_SORT_MIN")  # (the library reads it when a context is created)
    off = off_t.cpu().numpy().astype(np.int64)
    lens = np.diff(off)
    v = abi.CountersView(cnt, C, 2)
    assert int(v.pre.reads) == n and int(v.pre.length_sum) == int(lens.sum())
    assert int(v.pre.cyc[:, 0, :].sum()) == int(lens.sum())
    assert int(v.pre.base_qual_hist.sum()) == int(lens.sum())
    # (1)
    alive = res["dropped"] == 0
    assert (res["r1_start"][alive] + res["r1_len"][alive] <= lens[alive]).all()
    for i in range(2):
        m = res["n_frag"] > i
        assert (res["frag_start"][m, i] >= res["r1_start"][m]).all()
        assert (res["frag_start"][m, i] + res["frag_len"][m, i] <= res["r1_start"][m] + res["r1_len"][m]).all()
    assert int(v.filter.sum()) == int(res["n_frag"].sum())
    passing = [(res["n_frag"] > i) & (res["code"][:, i] == 0) for i in range(2)]
    assert int(v.post.reads) == int(passing[0].sum() + passing[1].sum())
    assert int(v.post.length_sum) == int(res["frag_len"][:, 0][passing[0]].sum() + res["frag_len"][:, 1][passing[1]].sum())
    # (2) build the batch of passing fragments on the device and run it with everything off
    starts = np.concatenate([off[:-1][passing[i]] + res["frag_start"][:, i][passing[i]] for i in range(2)])
    flens = np.concatenate([res["frag_len"][:, i][passing[i]] for i in range(2)]).astype(np.int64)
    foff = np.zeros(len(flens) + 1, np.int64)
    foff[1:] = np.cumsum(flens)
    idx = torch.from_numpy(np.repeat(starts - foff[:-1], flens)).cuda() + torch.arange(int(foff[-1]), device="cuda")
    fseq, fqual = seq_t[idx], qual_t[idx]
    plain = abi.FplOptions.default(adapter_enabled=0, qual_filter=0, length_filter=0)
    eng2 = engine_mod.Engine(plain, "", "", device=0, max_cycles=C)
    eng2.process_device(fseq, fqual, torch.from_numpy(foff).cuda(), int(flens.max()))
    cnt2 = eng2.counters()
    eng2.close()
    v2 = abi.CountersView(cnt2, C, 2)
    for f in ("cyc", "base_qual_hist", "median_hist", "median_bases", "kmer", "reads", "length_sum"):
        assert np.array_equal(np.asarray(getattr(v.post, f)), np.asarray(getattr(v2.pre, f))), f
    # (3)
    h = n // 2
    eng3 = engine_mod.Engine(opt, synth.START_ADAPTER, synth.END_ADAPTER, device=0, max_cycles=C)
    cut = int(off[h])
    eng3.process_device(seq_t[:cut], qual_t[:cut], off_t[:h + 1].contiguous(), max_len)
    eng3.process_device(seq_t[cut:].contiguous(), qual_t[cut:].contiguous(), (off_t[h:] - cut).contiguous(), max_len)
    cnt3 = eng3.counters()
    eng3.close()
    assert np.array_equal(cnt, cnt3)


def test_counters_tensor_all_reduce_over_rccl(orc, engine_mod):
    """bench.py --gpus N and multi-GPU hosts sum the counter buffers with torch.distributed (backend nccl = RCCL) on a
    zero-copy view of the device buffer: exercise exactly that call on a one-rank group"""
    import socket

    import torch
    import torch.distributed as dist

    from fastplong_amd import dist as fdist
    cfg = orc.Config(abi.FplOptions.default(), synth.START_ADAPTER, synth.END_ADAPTER)
    seq, qual, off = synth.adversarial(300, seed=3)
    C = int(np.diff(off.astype(np.int64)).max())
    _, want_cnt = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
    eng = engine_mod.Engine(cfg.opt, synth.START_ADAPTER, synth.END_ADAPTER, device=0, max_cycles=C)
    eng.process_host(seq, qual, off)
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        assert fdist.agree_capacity(C, device=torch.device("cuda", 0)) == C
        t = eng.counters_tensor()
        assert t.is_cuda and t.dtype == torch.int64 and t.numel() == len(want_cnt)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)  # one rank: the sum is the buffer itself
        torch.cuda.synchronize()
        assert np.array_equal(t.cpu().numpy(), want_cnt)
        t += 1  # a view, not a copy: the context sees the change
        torch.cuda.synchronize()
        assert np.array_equal(eng.counters(), want_cnt + 1)
    finally:
        dist.destroy_process_group()
        eng.close()


def test_async_pipeline_two_deep_bit_exact(orc, engine_mod):
    """fpl_process_batch_async / fpl_wait: batches of different sizes (the staging slots grow at different moments)
    from page-locked arrays, FPL_MAX_IN_FLIGHT deep; every batch's records and the accumulated counters equal the
    oracle's, and the misuse cases report FPL_ERR_STATE instead of queueing silently"""
    cfgd = CASES["full_pipeline"]
    cfg = orc.Config(abi.FplOptions.default(**cfgd["opt"]), cfgd["start"], cfgd["end"])
    sizes = [40, 300, 7, 120, 500, 64]
    batches = [synth.ont_like(n, seed=100 + i, median_len=1500, p_middle=0.05) if i % 2 == 0 else
               synth.adversarial(n, seed=200 + i) for i, n in enumerate(sizes)]
    C = max(int(np.diff(o.astype(np.int64)).max()) for _, _, o in batches)
    want, want_cnt = [], None
    for s, q, o in batches:
        r, c = orc.process_batch(cfg, s, q, o, max_cycles=C)
        want.append(r)
        want_cnt = c if want_cnt is None else want_cnt + c
    eng = engine_mod.Engine(cfg.opt, cfgd["start"], cfgd["end"], device=0, max_cycles=C)
    with pytest.raises(engine_mod.FplError, match="invalid state"):
        eng.wait()  # nothing in flight
    pinned = []
    for s, q, o in batches:  # page-locked copies, alive until the end of the test
        ps, pq, po = eng.pinned_array(len(s)), eng.pinned_array(len(q)), eng.pinned_array(len(o), np.uint64)
        ps[:], pq[:], po[:] = s, q, o
        pinned.append((ps, pq, po, np.zeros(max(len(o) - 1, 1), dtype=abi.RESULT_DTYPE)))
    done = 0
    for i, (ps, pq, po, res) in enumerate(pinned):
        if eng.in_flight() == abi.FPL_MAX_IN_FLIGHT:
            with pytest.raises(engine_mod.FplError, match="invalid state"):
                eng.submit_host(ps, pq, po, res)  # a third batch needs a wait first
            eng.wait()
            done += 1
        eng.submit_host(ps, pq, po, res)
    with pytest.raises(engine_mod.FplError, match="invalid state"):
        eng.process_host(*batches[0])  # the synchronous call does not jump the queue
    while eng.in_flight():
        eng.wait()
        done += 1
    assert done == len(batches)
    for (s, q, o), w, (_, _, _, res) in zip(batches, want, pinned):
        parity.assert_results_equal(res[:len(o) - 1], w, s, o)
    parity.assert_counters_equal(eng.counters(), want_cnt, C, 2)
    # the in-process merge entry point with one context: agrees on C (a no-op here) and leaves the totals alone
    import ctypes as Ct
    arr = (Ct.c_void_p * 1)(eng.h)
    assert eng.L.fpl_allreduce_counters(arr, 1) == 0
    parity.assert_counters_equal(eng.counters(), want_cnt, C, 2)
    assert eng.L.fpl_allreduce_counters(arr, 0) == abi.FPL_ERR_ARG
    eng.close()


def test_allreduce_counters_runs_rccl_on_one_gpu(orc, engine_mod, monkeypatch):
    """fpl_allreduce_counters with FPL_RCCL_FORCE=1: the merge of ONE context goes through the whole RCCL path of the C-ABI --
    finding and dlopen-ing librccl, ncclCommInitAll over the context's device, the grouped in-place int64 ncclAllReduce on the
    context's stream, ncclCommDestroy -- which N > 1 contexts take in bin/fastplong_amd --gpus N.  One rank: the buffer must
    come out as it went in, and the capacity agreement still happens."""
    import ctypes as Ct

    cfg = orc.Config(abi.FplOptions.default(cut_front=1, polyx=1), synth.START_ADAPTER, synth.END_ADAPTER)
    seq, qual, off = synth.adversarial(800, seed=77)
    C = int(np.diff(off.astype(np.int64)).max())
    _, want_cnt = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
    eng = engine_mod.Engine(cfg.opt, synth.START_ADAPTER, synth.END_ADAPTER, device=0, max_cycles=C)
    eng.process_host(seq, qual, off)
    arr = (Ct.c_void_p * 1)(eng.h)
    monkeypatch.setenv("FPL_RCCL_FORCE", "1")
    for _ in range(2):  # (a second merge builds a second communicator)
        rc = eng.L.fpl_allreduce_counters(arr, 1)
        assert rc == 0, (eng.L.fpl_strerror(rc), eng.L.fpl_last_error(eng.h))
        assert (eng.L.fpl_last_error(eng.h) or b"") == b""
        parity.assert_counters_equal(eng.counters(), want_cnt, C, 2)
    used = eng.L.fpl_rccl_library().decode()
    assert "librccl" in used, used
    # communicators made ahead of the merge (what bin/fastplong_amd does on a thread of its own while its batches run) are the
    # ones the merge then uses; made twice for the same devices they are made once; handed back with (NULL, 0)
    assert eng.L.fpl_comm_init(arr, 1) == 0 and eng.L.fpl_comm_init(arr, 1) == 0
    rc = eng.L.fpl_allreduce_counters(arr, 1)
    assert rc == 0, (eng.L.fpl_strerror(rc), eng.L.fpl_last_error(eng.h))
    parity.assert_counters_equal(eng.counters(), want_cnt, C, 2)
    assert eng.L.fpl_comm_init(None, 0) == 0
    assert eng.L.fpl_comm_init(arr, 0) == abi.FPL_ERR_ARG
    # the process has ONE librccl mapped: the loader took the copy torch brought along rather than a second one
    mapped = {line.split()[-1] for line in open("/proc/self/maps") if "librccl" in line}
    assert len(mapped) == 1, mapped
    # the batch path still works on the context afterwards (streams / device state untouched)
    eng.process_host(seq, qual, off)
    parity.assert_counters_equal(eng.counters(), 2 * want_cnt, C, 2)
    eng.close()


def test_offsets_beyond_4_gib_bit_exact(orc, engine_mod):
    """A batch whose byte offsets pass 2^32 (the bench batch is 9 GB per array): the records of the LAST reads --
    the ones whose addresses need more than 32 bits in every kernel -- equal the oracle's on the same reads taken
    alone (results are per read), the counters conserve reads and bases, and they do not depend on how the batch is
    cut (partition invariance at full size)."""
    import torch

    n_reads, median = 560_000, 8000
    free, _ = torch.cuda.mem_get_info(0)
    if free < 40 * 2 ** 30:
        pytest.skip("needs ~25 GiB of free HBM")
    dev = torch.device("cuda", 0)
    seq_t, qual_t, off_t, max_len = synth.device_batch(n_reads, seed=77, median_len=median, device=dev)
    n_bytes = int(off_t[-1].item())
    assert n_bytes > 2 ** 32 + 2 ** 28, n_bytes
    cfgd = CASES["full_pipeline"]
    opt = abi.FplOptions.default(**cfgd["opt"])
    C = max_len
    eng = engine_mod.Engine(opt, cfgd["start"], cfgd["end"], device=0, max_cycles=C)
    rt = eng.process_device(seq_t, qual_t, off_t, max_len)
    torch.cuda.synchronize()
    res = eng.results_to_numpy(rt, n_reads)
    cnt = eng.counters()
    eng.close()
    off = off_t.cpu().numpy().astype(np.int64)
    # (1) the tail of the batch, beyond 4 GiB, against the oracle
    tail = 1500
    first = n_reads - tail
    assert off[first] > 2 ** 32
    a = int(off[first])
    s_tail = seq_t[a:].cpu().numpy()
    q_tail = qual_t[a:].cpu().numpy()
    o_tail = (off[first:] - a).astype(np.uint64)
    cfg = orc.Config(opt, cfgd["start"], cfgd["end"])
    want_res, _ = orc.process_batch(cfg, s_tail, q_tail, o_tail, max_cycles=C)
    parity.assert_results_equal(res[first:], want_res, s_tail, o_tail)
    # ... and a window that straddles the 4 GiB line
    k = int(np.searchsorted(off, 2 ** 32)) - 200
    a, b = int(off[k]), int(off[k + 400])
    want_mid, _ = orc.process_batch(cfg, seq_t[a:b].cpu().numpy(), qual_t[a:b].cpu().numpy(),
                                    (off[k:k + 401] - a).astype(np.uint64), max_cycles=C)
    parity.assert_results_equal(res[k:k + 400], want_mid)
    # (2) conservation
    v = abi.CountersView(cnt, C, 2)
    assert int(v.pre.reads) == n_reads and int(v.pre.length_sum) == n_bytes
    assert int(np.asarray(v.pre.base_qual_hist).sum()) == n_bytes
    assert int(np.asarray(v.pre.cyc)[:, 0, :].sum()) == n_bytes  # kind 0 (base contents) over all cycles and classes
    passing = 0
    for f in range(2):
        m = (res["n_frag"] > f) & (res["code"][:, f] == abi.FPL_PASS_FILTER) & (res["dropped"] == 0)
        passing += int(res["frag_len"][m, f].astype(np.int64).sum())
    assert int(v.post.length_sum) == passing == int(np.asarray(v.post.cyc)[:, 0, :].sum())
    # (3) partition invariance at this size: two halves, the second one entirely beyond 2^31 bytes
    h = n_reads // 2
    cut = int(off[h])
    eng2 = engine_mod.Engine(opt, cfgd["start"], cfgd["end"], device=0, max_cycles=C)
    r1 = eng2.process_device(seq_t[:cut], qual_t[:cut], off_t[:h + 1].contiguous(), max_len)
    r2 = eng2.process_device(seq_t[cut:], qual_t[cut:], (off_t[h:] - cut).contiguous(), max_len)
    torch.cuda.synchronize()
    cnt2 = eng2.counters()
    res2 = np.concatenate([eng2.results_to_numpy(r1, h), eng2.results_to_numpy(r2, n_reads - h)])
    eng2.close()
    assert np.array_equal(cnt, cnt2)
    parity.assert_results_equal(res2, res)


@pytest.mark.parametrize("workload", ["c2_adapter_only", "c3_full_pipeline", "c4_mixed", "c5_hifi64"])
@pytest.mark.parametrize("forced", [False, True])
def test_every_bench_workload_bit_exact(orc, engine_mod, monkeypatch, workload, forced):
    """What bench.py runs, as bench.py builds it (its generators, its option sets, its adapters -- c5 with -s / -e set to the
    first FASTA adapter and its reverse complement), at a size the oracle checks in full: bench.parity_of_timed_batch is the very
    function behind the `parity_sample` field of the bench line (records of the step's own buffer + counters of a fresh context).  forced: with the kernels that only batches of bench size
    take (k_trim_ends_batched, k_stats_sorted) forced on the small batch."""
    import torch

    import bench

    if forced:
        monkeypatch.setenv("FPL_TRIM_BATCH_MIN", "1")
        monkeypatch.setenv("FPL_STATS_SORT_MIN", "1")
    else:
        monkeypatch.delenv("FPL_TRIM_BATCH_MIN", raising=False)
        monkeypatch.delenv("FPL_STATS_SORT_MIN", raising=False)
    wl = dict(bench.WORKLOADS[workload])
    g = dict(wl["gen"])
    if g["kind"] == "hifi":
        g.update(mean_len=6000, sd_len=1500)
    elif "max_len" in g:
        g.update(max_len=40000)
    else:
        g.update(median_len=3000)
    wl["gen"] = g
    rig = bench.Rig()
    dev = torch.device("cuda", 0)
    seq_t, qual_t, off_t, max_len, ad_start, ad_end, ad_fasta = rig.make_batch(wl, 700, 0, dev)
    if workload == "c5_hifi64":
        assert ad_start and ad_end == synth.revcomp(ad_start) and len(ad_fasta) == 64
    opt = abi.FplOptions.default(**wl["opt"])
    verdict, what, n = _bench_parity(bench, rig, opt, (ad_start, ad_end, ad_fasta), seq_t, qual_t, off_t, max_len, 500, 200)
    assert verdict == "ok" and n == 500, verdict
    assert "200 reads spread evenly over reads 500..699" in what


def _bench_parity(bench, rig, opt, adapters, seq_t, qual_t, off_t, max_len, n_prefix, n_strided):
    """one step of the batch into a record buffer, as bench.py's timed loop does, then bench.py's own check of that buffer"""
    import torch

    n = off_t.numel() - 1
    eng = rig.engine(opt, adapters[0], adapters[1], adapters[2], 0, max_len)
    res_t = torch.empty(n * 36, dtype=torch.uint8, device=seq_t.device)
    eng.process_device(seq_t, qual_t, off_t, max_len, res_t, rig.stream(seq_t.device))
    torch.cuda.synchronize()
    eng.close()
    verdict, what, _, pn, _ = bench.parity_of_timed_batch(rig, opt, adapters, seq_t, qual_t, off_t, res_t, max_len, 0, 0, n_prefix,
                                                          n_strided, 16)
    return verdict, what, pn


@pytest.mark.parametrize("workload,n_reads,median", [("c3_full_pipeline", 160_000, 3000), ("c4_mixed", 155_000, 2500)])
def test_bench_sized_batch_no_hooks_bit_exact(orc, engine_mod, monkeypatch, workload, n_reads, median):
    """A batch large enough for the library to pick, BY ITS OWN SIZE RULES, the kernels bench.py times (k_trim_ends_batched from
    65 536 reads, k_stats_sorted + bucket kernels from 150 000): no FPL_* hook in the environment.  Every record and every
    counter of the whole batch against the oracle (16 host threads), through bench.py's own checker."""
    import torch

    import bench

    for k in ("FPL_TRIM_BATCH_MIN", "FPL_STATS_SORT_MIN", "FPL_STATS_MIN_BUCKET", "FPL_STATS_PER"):
        monkeypatch.delenv(k, raising=False)
    wl = dict(bench.WORKLOADS[workload])
    g = dict(wl["gen"], median_len=median)
    if "max_len" in g:
        g.update(max_len=60000)
    wl["gen"] = g
    rig = bench.Rig()
    seq_t, qual_t, off_t, max_len, ad_start, ad_end, ad_fasta = rig.make_batch(wl, n_reads, 0, torch.device("cuda", 0))
    n = off_t.numel() - 1
    assert n >= 150_000
    opt = abi.FplOptions.default(**wl["opt"])
    verdict, what, pn = _bench_parity(bench, rig, opt, (ad_start, ad_end, ad_fasta), seq_t, qual_t, off_t, max_len, n, 0)
    assert verdict == "ok" and pn == n, verdict
    assert "no hook set" in what


WIDE_QUAL_CASES = {
    # thresholds inside the reference's own option ranges (src/options.cpp:133-181: -q / -e 0..93, cut qualities 1..30) ...
    "full_pipeline": CASES["full_pipeline"]["opt"],
    "q60_e60_cut30": dict(cut_front=1, cut_tail=1, cut_front_window=5, cut_tail_window=4, cut_front_quality=30, cut_tail_quality=30,
                          polyx=1, complexity_filter=1, qualified_qual=33 + 60, unqualified_percent_limit=30, avg_qual_req=60),
    "q93_e93": dict(cut_front=1, cut_tail=1, cut_front_window=1, cut_tail_window=1000, cut_front_quality=30, cut_tail_quality=1,
                    qualified_qual=33 + 93, unqualified_percent_limit=0, avg_qual_req=93, break_enabled=0),
    # ... and beyond them (the C-ABI takes any integer; the arithmetic is the same)
    "cut60_q80": dict(cut_front=1, cut_tail=1, cut_front_window=7, cut_tail_window=3, cut_front_quality=60, cut_tail_quality=85,
                      polyx=1, qualified_qual=33 + 80, unqualified_percent_limit=50, avg_qual_req=75, complexity_filter=1),
    "break_mask_hi": dict(cut_front=1, break_enabled=1, break_window=20, break_quality=70, mask_enabled=1, mask_window=9, mask_quality=88,
                          n_base_percent_limit=95, unqualified_percent_limit=90),
}


def _wide_quality_batch(seed, n_ont=900, n_adv=1500, median_len=2500):
    a = synth.ont_like(n_ont, seed=seed, median_len=median_len, p_middle=0.05)
    b = synth.adversarial(n_adv, seed=seed + 1)
    reads = []
    for (s_, q_, o_) in (a, b):
        reads += [(s_[int(o_[i]):int(o_[i + 1])], q_[int(o_[i]):int(o_[i + 1])]) for i in range(len(o_) - 1)]
    seq, qual, off = synth.pack(reads)
    return seq, synth.wide_qualities(qual, off, seed + 2), off


@pytest.mark.parametrize("sorted_stats", [True, False])
@pytest.mark.parametrize("name", sorted(WIDE_QUAL_CASES))
def test_full_quality_byte_range_bit_exact(orc, engine_mod, monkeypatch, name, sorted_stats):
    """Quality bytes over the whole FASTQ range '!'..'~' (every other generator stays within Q2..Q50): reads of all '~' (HiFi),
    all '!', uniform 33..126, the two ends of the range in alternating runs -- through the FORCED k_trim_ends_batched
    (tests/conftest.py) and the forced k_stats_sorted (packed 22-bit raw-quality sums, the Q20 / Q30 tests as bit 7 of q + 75 /
    q + 65, the v_dot4 packs) or the unsorted k_stats, with thresholds up to the top of the reference's option ranges and beyond"""
    if sorted_stats:
        monkeypatch.setenv("FPL_STATS_SORT_MIN", "1")
    else:
        monkeypatch.delenv("FPL_STATS_SORT_MIN", raising=False)
    seq, qual, off = _wide_quality_batch(900 + len(name))
    assert qual.min() == 33 and qual.max() == 126
    okw = WIDE_QUAL_CASES[name]
    if okw.get("break_enabled") or okw.get("mask_enabled"):
        cfg = orc.Config(abi.FplOptions.default(**okw), synth.START_ADAPTER, synth.END_ADAPTER)
        C = int(np.diff(off.astype(np.int64)).max())
        want_res, want_cnt, want_f, want_r = orc.process_batch_ex(cfg, seq, qual, off, max_cycles=C)
        eng = engine_mod.Engine(cfg.opt, synth.START_ADAPTER, synth.END_ADAPTER, device=0, max_cycles=C)
        got_res = eng.process_host(seq, qual, off)
        got_f, got_r = eng.fragments()
        got_cnt = eng.counters()
        eng.close()
        parity.assert_fragments_equal(got_f, got_r, want_f, want_r)
        parity.assert_results_equal(got_res, want_res, seq, off)
        parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)
        return
    _, cnt = _run_both(orc, engine_mod, dict(opt=okw, start=synth.START_ADAPTER, end=synth.END_ADAPTER), seq, qual, off, via="device")
    v = abi.CountersView(cnt, int(np.diff(off.astype(np.int64)).max()), 2)
    assert v.pre.base_qual_hist[126] > 0 and v.pre.base_qual_hist[33] > 0


def test_full_quality_byte_range_fasta_chain_bit_exact(orc, engine_mod):
    """the same qualities through k_trim_ends<2> (64-adapter FASTA chain; its trimAndCut / polyX are the wave-per-read forms)"""
    seq, qual, off, ads = synth.hifi_like(60, seed=9, mean_len=5000, sd_len=1500, n_adapters=64)
    qual = synth.wide_qualities(qual, off, 17, share=0.9)
    cfgd = dict(opt=dict(cut_front=1, cut_tail=1, cut_front_quality=30, cut_tail_quality=30, qualified_qual=33 + 60, avg_qual_req=50),
                start=ads[0], end=synth.revcomp(ads[0]))
    _run_both(orc, engine_mod, cfgd, seq, qual, off, fasta=sorted(ads))


def test_full_quality_byte_range_bench_sized_no_hooks_bit_exact(orc, engine_mod, monkeypatch):
    """160 000 reads with qualities over 33..126 and NO hook in the environment: the kernels the library picks by its own size rules
    (k_trim_ends_batched, k_stats_sorted with slices of 16 320 reads -- 16 320 x 126 is what the 22-bit fields have to hold),
    every record and every counter against the oracle on 16 host threads"""
    import torch

    import bench

    for k in ("FPL_TRIM_BATCH_MIN", "FPL_STATS_SORT_MIN", "FPL_STATS_MIN_BUCKET", "FPL_STATS_PER"):
        monkeypatch.delenv(k, raising=False)
    wl = dict(bench.WORKLOADS["c3_full_pipeline"])
    wl["gen"] = dict(wl["gen"], median_len=1500)
    rig = bench.Rig()
    dev = torch.device("cuda", 0)
    seq_t, qual_t, off_t, max_len, ad_start, ad_end, ad_fasta = rig.make_batch(wl, 160_000, 0, dev)
    n = off_t.numel() - 1
    # per-read modes as in synth.wide_qualities, built on the device: 0 keep, 1 all '~', 2 all '!', 3 uniform 33..126, 4 84..126
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    mode = torch.randint(0, 5, (n,), generator=g, device=dev)
    # the first 20 000 reads all '~' and nothing else: one slice of k_stats_sorted whose cells hold nothing but the largest byte
    mode[:20_000] = 1
    per_base = torch.repeat_interleave(mode, off_t[1:] - off_t[:-1])
    uni = torch.randint(33, 127, (qual_t.numel(),), generator=g, device=dev, dtype=torch.int64).to(torch.uint8)
    qual_t = torch.where(per_base == 1, torch.full_like(qual_t, 126), qual_t)
    qual_t = torch.where(per_base == 2, torch.full_like(qual_t, 33), qual_t)
    qual_t = torch.where(per_base == 3, uni, qual_t)
    qual_t = torch.where(per_base == 4, torch.clamp(uni, min=84), qual_t).contiguous()
    del per_base, uni
    opt = abi.FplOptions.default(**dict(wl["opt"], qualified_qual=33 + 40, avg_qual_req=30))
    verdict, what, pn = _bench_parity(bench, rig, opt, (ad_start, ad_end, ad_fasta), seq_t, qual_t, off_t, max_len, n, 0)
    assert verdict == "ok" and pn == n, verdict
    assert "no hook set" in what


@pytest.mark.parametrize("sorted_stats,sizes,gate", [(False, [900, 1500, 700, 2500, 1100], "0"), (True, [900, 1500, 700, 2500, 1100], "0"),
                                                     (True, [900, 1500, 700, 2500, 1100], "1"), (True, [70000, 90000, 66000, 120000], "0")])
def test_end_trims_ahead_of_the_previous_batch_bit_exact(orc, engine_mod, monkeypatch, sorted_stats, sizes, gate):
    """fpl_assume_inputs_ready (ABI v6): with resident inputs the end trims of batch k + 1 run on a stream of their own beside the
    kernels of batch k -- beside its k_scan (FPL_TRIM_AHEAD_GATE=0, the default) or behind its statistics pass (=1) -- with two
    ReadState[] / work-counter sets, batches alternating.  Different batches back to back, sizes up and down (the workspace grows
    in between), against the same batches through a context without the promise -- every record, every counter -- and against the
    oracle for the last one; fpl_get_batch_forms says that the trims really ran ahead.  The last case has batches the library
    sends ahead by its own size rule, large enough for the two kernels to share the chip for a while."""
    import torch

    if sorted_stats:
        monkeypatch.setenv("FPL_STATS_SORT_MIN", "1")
    if max(sizes) < 65536:
        monkeypatch.setenv("FPL_TRIM_BATCH_MIN", "1")  # (the library only sends the trims of batches of >= 65 536 reads ahead)
    monkeypatch.setenv("FPL_TRIM_AHEAD_GATE", gate)
    opt = abi.FplOptions.default(cut_front=1, cut_tail=1, cut_front_window=5, cut_tail_window=5, polyx=1, complexity_filter=1)
    batches = [synth.ont_like(n, seed=60 + i, median_len=700, p_middle=0.05) for i, n in enumerate(sizes)]
    C = max(int(np.diff(o.astype(np.int64)).max()) for _, _, o in batches)
    out = {}
    for promise in (False, True):
        eng = engine_mod.Engine(opt, synth.START_ADAPTER, synth.END_ADAPTER, device=0, max_cycles=C)
        if promise:
            eng.assume_inputs_ready(True)
        dev_batches = [(torch.from_numpy(s_).cuda(), torch.from_numpy(q_).cuda(), torch.from_numpy(o_.astype(np.int64)).cuda(), len(o_) - 1)
                       for s_, q_, o_ in batches]
        torch.cuda.synchronize()  # (the promise: complete on the device when the calls are made)
        res = [eng.process_device(st, qt, ot, C) for st, qt, ot, _ in dev_batches]  # all five enqueued before anything is waited for
        torch.cuda.synchronize()
        out[promise] = ([eng.results_to_numpy(r, b[3]) for r, b in zip(res, dev_batches)], eng.counters())
        assert eng.batch_forms()["trims_ahead"] == (len(batches) - 1 if promise else 0)
        eng.close()
    for a, b, (s_, _, o_) in zip(out[False][0], out[True][0], batches):
        parity.assert_results_equal(b, a, s_, o_)
    assert np.array_equal(out[False][1], out[True][1])
    cfg = orc.Config(opt, synth.START_ADAPTER, synth.END_ADAPTER)
    want_res, _ = orc.process_batch(cfg, *batches[-1], max_cycles=C)
    parity.assert_results_equal(out[True][0][-1], want_res, batches[-1][0], batches[-1][2])


def test_batch_forms_say_which_kernels_a_batch_took(engine_mod, monkeypatch):
    """fpl_get_batch_forms (ABI v6): the library picks k_trim_ends_batched / k_stats_sorted by the batch's size; a host that
    flushes small batches runs the other forms, and this call is how it finds out"""
    import torch

    for k in ("FPL_TRIM_BATCH_MIN", "FPL_STATS_SORT_MIN", "FPL_STATS_MIN_BUCKET", "FPL_STATS_PER"):
        monkeypatch.delenv(k, raising=False)
    opt = abi.FplOptions.default(cut_front=1, cut_tail=1, polyx=1)
    seq, qual, off = synth.ont_like(2000, seed=3, median_len=600)
    eng = engine_mod.Engine(opt, synth.START_ADAPTER, synth.END_ADAPTER, device=0, max_cycles=int(np.diff(off.astype(np.int64)).max()))
    eng.process_host(seq, qual, off)
    eng.process_host(seq[:int(off[500])], qual[:int(off[500])], off[:501])
    assert eng.batch_forms() == dict(batches=2, reads=2500, trim_batched=0, stats_sorted=0, largest=2000, trims_ahead=0)
    eng.reset_counters()
    assert eng.batch_forms()["batches"] == 0
    # 160 000 short reads: both thresholds crossed
    n = 160_000
    lens = torch.full((n,), 400, dtype=torch.int64)
    off_t = torch.zeros(n + 1, dtype=torch.int64)
    off_t[1:] = torch.cumsum(lens, 0)
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    st = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")[torch.randint(0, 4, (n * 400,), generator=g, device="cuda")]
    qt = torch.randint(40, 70, (n * 400,), generator=g, device="cuda", dtype=torch.int64).to(torch.uint8)
    eng.process_device(st, qt, off_t.cuda(), 400)
    torch.cuda.synchronize()
    assert eng.batch_forms() == dict(batches=1, reads=n, trim_batched=1, stats_sorted=1, largest=n, trims_ahead=0)
    eng.close()
    # the test hooks move the thresholds, and the report follows what really ran
    monkeypatch.setenv("FPL_TRIM_BATCH_MIN", "1")
    monkeypatch.setenv("FPL_STATS_SORT_MIN", "1")
    eng = engine_mod.Engine(opt, synth.START_ADAPTER, synth.END_ADAPTER, device=0, max_cycles=int(np.diff(off.astype(np.int64)).max()))
    eng.process_host(seq, qual, off)
    assert eng.batch_forms() == dict(batches=1, reads=2000, trim_batched=1, stats_sorted=1, largest=2000, trims_ahead=0)
    eng.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_long_trim_scans_fall_back_bit_exact(orc, engine_mod, seed):
    """k_trim_ends_batched: lanes whose trimAndCut / polyX scans outlast the iteration cap are redone by the wave-per-read forms"""
    from tests.test_kernels_emu import _reads_with_long_scans

    seq, qual, off = _reads_with_long_scans(seed, n=600)
    cfgd = dict(opt=dict(cut_front=1, cut_tail=1, cut_front_window=5, cut_tail_window=4, polyx=1, n_base_percent_limit=60),
                start=synth.START_ADAPTER, end=synth.END_ADAPTER)
    res, _ = _run_both(orc, engine_mod, cfgd, seq, qual, off)
    assert (res["dropped"] != 0).any()


@pytest.mark.parametrize("seed,kind", [(1, "default"), (2, "repeat"), (3, "mixed"), (4, "repeat"), (5, "default")])
def test_partial_pattern_searches_lane_per_read_bit_exact(orc, engine_mod, seed, kind):
    """k_trim_ends_batched: the partial-pattern searches walk each lane's candidate columns; reads with more candidates than the
    cap (adapters with repeats, patterns with insertions) are redone by the wave-per-read search"""
    from tests.test_kernels_emu import REPEAT_END, REPEAT_START, _reads_with_partial_adapters

    start = synth.START_ADAPTER if kind == "default" else REPEAT_START
    end = REPEAT_END if kind == "repeat" else synth.END_ADAPTER
    seq, qual, off = _reads_with_partial_adapters(seed, start, end, n=1400)
    res, _ = _run_both(orc, engine_mod, dict(opt=dict(), start=start, end=end), seq, qual, off)
    assert (res["r1_start"] > 0).sum() > 100


@pytest.mark.parametrize("opts", [dict(), dict(complexity_filter=1, n_base_percent_limit=60), dict(adapter_enabled=0, complexity_filter=1)])
def test_scan_ragged_last_tiles_bit_exact(orc, engine_mod, opts):
    """k_scan: reads that end at every offset of a 32-byte chunk / a 1984-byte tile, in N, in one base, in lower-case letters"""
    from tests.test_kernels_emu import _reads_ending_anywhere

    on = opts.get("adapter_enabled", 1)
    for seed in (5, 6):
        seq, qual, off = _reads_ending_anywhere(seed)
        _run_both(orc, engine_mod, dict(opt=opts, start=synth.START_ADAPTER if on else "", end=synth.END_ADAPTER if on else ""), seq, qual, off)


@pytest.mark.parametrize("opts,chunk", [(dict(), "4"), (dict(cut_front=1, cut_tail=1, complexity_filter=1, n_base_percent_limit=60), "3"),
                                        (dict(cut_front=1, cut_tail=1, polyx=1), "64"), (dict(), "2")])
def test_scan_pair_packing_bit_exact(orc, engine_mod, monkeypatch, opts, chunk):
    """k_scan: the head of a chunk's next read in the free lanes of a read's last tile -- every split of the 62 lanes, next reads
    on both sides of the length a head needs, N / lower case / middle adapters in and across the head, dropped next reads, reads
    whose trimmed end is too long to host (tests/test_kernels_emu.py::_reads_for_pair_packing)"""
    from tests.test_kernels_emu import _reads_for_pair_packing

    monkeypatch.setenv("FPL_SCAN_CHUNK", chunk)
    for seed in (7, 8):
        seq, qual, off = _reads_for_pair_packing(seed)
        res, _ = _run_both(orc, engine_mod, dict(opt=opts, start=synth.START_ADAPTER, end=synth.END_ADAPTER), seq, qual, off, via="device")
        assert (res["n_frag"] == 2).any()


def test_plain_scan_without_chunks_bit_exact(orc, engine_mod, monkeypatch):
    """the built-in chunk rule (no FPL_SCAN_CHUNK): a batch this small is dealt one read per dequeue, no read has a next one"""
    from tests.test_kernels_emu import _reads_for_pair_packing

    monkeypatch.delenv("FPL_SCAN_CHUNK", raising=False)
    seq, qual, off = _reads_for_pair_packing(9)
    _run_both(orc, engine_mod, CASES["full_pipeline"], seq, qual, off, via="device")


def test_long_reads_split_by_middle_adapters_bit_exact(orc, engine_mod):
    """k_resolve -> k_redo: split reads beyond 16 kb (front of the REDO list) and below (its far end), many per block"""
    seq, qual, off = synth.ont_like(500, seed=8, median_len=17000, sigma_len=0.3, p_middle=0.6)
    res, _ = _run_both(orc, engine_mod, CASES["full_pipeline"], seq, qual, off, via="device")
    split = res["n_frag"] == 2
    assert (split & (res["r1_len"] > 16384)).sum() > 50 and (split & (res["r1_len"] <= 16384)).sum() > 50
