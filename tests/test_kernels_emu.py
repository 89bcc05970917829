"""The product kernels (fastplong_amd/csrc/kernels.h), compiled for the host on the test-only
lock-step emulator, against the oracle on seeded batches.  This is host-logic coverage for the
`-m "not gpu"` run; the real parity tests (-m gpu) run the same comparisons on an MI355X."""
import os

import numpy as np
import pytest

from fastplong_amd import abi, synth
from tests import parity
from tests.emu import emu


def _lev_cases(rng, n):
    for _ in range(n):
        alen = int(rng.choice([1, 2, 6, 15, 16, 17, 24, 33, 63, 64, 65, 100, 128, 129, 200, 255]))
        ad = "".join("ACGTN"[i] for i in rng.integers(0, 5, alen))
        yield ad


def test_bitparallel_levenshtein_matches_oracle(orc):
    rng = np.random.default_rng(3)
    L = emu.lib()
    for ad in _lev_cases(rng, 120):
        alen = len(ad)
        adb = ad.encode()
        for _ in range(6):
            # (shift, m) slices as the kernels use them: whole adapter, suffix, prefix
            m = int(rng.integers(1, alen + 1))
            shift = int(rng.choice([0, alen - m]))
            pat = ad[shift:shift + m]
            n = int(rng.choice([m, m, max(0, m - 3), m + 5]))
            if rng.random() < 0.6:
                text = list(pat[:n].ljust(n, "A"))
                for i in np.nonzero(rng.random(n) < 0.2)[0]:
                    text[i] = "ACGT"[rng.integers(4)]
                text = "".join(text)
            else:
                text = "".join("ACGT"[i] for i in rng.integers(0, 4, n))
            want = orc.edit_distance(pat, text)
            got = L.emu_lev_bp64(adb, alen, shift, m, text.encode(), n)
            assert got == want, (ad, shift, m, text)
            # wave-cooperative thresholded form: exact when <= thr, otherwise anything > thr
            for thr in (want, max(0, want - 1), want + 3, 0):
                gw = L.emu_lev_wave(adb, alen, shift, m, text.encode(), n, thr)
                assert (gw == want) if want <= thr else (gw > thr), (ad, shift, m, text, thr, gw, want)
        plen = min(16, alen)
        for _ in range(4):
            text = "".join("ACGTN"[i] for i in rng.integers(0, 5, plen))
            assert L.emu_lev_bp32_start(adb, alen, text.encode(), plen) == orc.edit_distance(ad[alen - plen:], text)
            assert L.emu_lev_bp32_end(adb, alen, text.encode(), plen) == orc.edit_distance(ad[:plen], text)


CASES = {
    "defaults_adapters": dict(opt=dict(), start=synth.START_ADAPTER, end=synth.END_ADAPTER),
    "full_pipeline": dict(opt=dict(cut_front=1, cut_tail=1, cut_front_window=5, cut_tail_window=5, polyx=1,
                                   complexity_filter=1), start=synth.START_ADAPTER, end=synth.END_ADAPTER),
    "no_adapter_trimming": dict(opt=dict(adapter_enabled=0), start="", end=""),
    "nasty_options": dict(opt=dict(trim_front=3, trim_tail=2, cut_front=1, cut_tail=1, cut_front_window=7,
                                   cut_front_quality=15, cut_tail_window=3, cut_tail_quality=25, polyx=1,
                                   polyx_min_len=8, complexity_filter=1, complexity_percent=40,
                                   qualified_qual=33 + 20, unqualified_percent_limit=30, n_base_percent_limit=5,
                                   avg_qual_req=12, required_length=30, max_length=350, ed_max=0.3,
                                   trimming_extension=5), start=synth.START_ADAPTER, end=synth.END_ADAPTER),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_emulated_kernels_match_oracle_adversarial(orc, name):
    case = CASES[name]
    cfg = orc.Config(abi.FplOptions.default(**case["opt"]), case["start"], case["end"])
    seq, qual, off = synth.adversarial(110, seed=sum(map(ord, name)) % 1000, start_adapter=synth.START_ADAPTER,
                                       end_adapter=synth.END_ADAPTER)
    C = int(np.diff(off.astype(np.int64)).max()) + 3
    want_res, want_cnt = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
    got_res, got_cnt = emu.process_batch(cfg, seq, qual, off, C)
    parity.assert_results_equal(got_res, want_res, seq, off)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)


def test_emulated_kernels_match_oracle_ont_like(orc):
    cfg = orc.Config(abi.FplOptions.default(cut_front=1, cut_tail=1, cut_front_window=5, cut_tail_window=5, polyx=1,
                                            complexity_filter=1), synth.START_ADAPTER, synth.END_ADAPTER)
    seq, qual, off = synth.ont_like(24, seed=2, median_len=1500, p_middle=0.3, p_polya=0.3)
    C = int(np.diff(off.astype(np.int64)).max()) + 1
    want_res, want_cnt = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
    got_res, got_cnt = emu.process_batch(cfg, seq, qual, off, C)
    parity.assert_results_equal(got_res, want_res, seq, off)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)
    assert (want_res["n_frag"] == 2).any()  # the middle-adapter split path ran


@pytest.mark.parametrize("sorted_stats,which", [(True, 0), (False, 1), (True, 2)])
@pytest.mark.parametrize("okw_all", [[
    dict(cut_front=1, cut_tail=1, cut_front_window=5, cut_tail_window=5, polyx=1, complexity_filter=1),
    dict(cut_front=1, cut_tail=1, cut_front_window=5, cut_tail_window=4, cut_front_quality=30, cut_tail_quality=30, polyx=1,
         qualified_qual=33 + 60, unqualified_percent_limit=30, avg_qual_req=60),
    dict(cut_front=1, cut_tail=1, cut_front_window=7, cut_tail_window=3, cut_front_quality=60, cut_tail_quality=85,
         qualified_qual=33 + 93, unqualified_percent_limit=50, avg_qual_req=93)]])
def test_emulated_full_quality_byte_range(orc, monkeypatch, okw_all, sorted_stats, which):
    """quality bytes over '!'..'~' (synth.wide_qualities; the other generators stay within Q2..Q50) with thresholds up to the top of
    the reference's option ranges (src/options.cpp:133-181) and beyond -- the emulator builds the kernels' C forms, the -m gpu
    test of the same name the v_dot4 / inline-asm ones"""
    okw = okw_all[which]
    if sorted_stats:
        monkeypatch.setenv("FPL_STATS_MIN_BUCKET", "1")
    cfg = orc.Config(abi.FplOptions.default(**okw), synth.START_ADAPTER, synth.END_ADAPTER)
    a = synth.ont_like(30, seed=31, median_len=800, p_middle=0.1, p_polya=0.2)
    b = synth.adversarial(70, seed=32)
    reads = []
    for (s_, q_, o_) in (a, b):
        reads += [(s_[int(o_[i]):int(o_[i + 1])], q_[int(o_[i]):int(o_[i + 1])]) for i in range(len(o_) - 1)]
    seq, qual, off = synth.pack(reads)
    qual = synth.wide_qualities(qual, off, 33)
    assert qual.min() == 33 and qual.max() == 126
    C = int(np.diff(off.astype(np.int64)).max()) + 1
    want_res, want_cnt = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
    got_res, got_cnt = emu.process_batch(cfg, seq, qual, off, C)
    parity.assert_results_equal(got_res, want_res, seq, off)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)


def test_emulated_wave_per_read_trim_kernel(orc, monkeypatch):
    """k_trim_ends<1>, what batches of fewer than 65 536 reads take (the suite forces k_trim_ends_batched elsewhere)"""
    monkeypatch.delenv("FPL_TRIM_BATCH_MIN", raising=False)
    cfg = orc.Config(abi.FplOptions.default(cut_front=1, cut_tail=1, cut_front_window=5, cut_tail_window=5, polyx=1,
                                            complexity_filter=1), synth.START_ADAPTER, synth.END_ADAPTER)
    seq, qual, off = synth.adversarial(90, seed=77)
    C = int(np.diff(off.astype(np.int64)).max()) + 1
    want_res, want_cnt = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
    got_res, got_cnt = emu.process_batch(cfg, seq, qual, off, C)
    parity.assert_results_equal(got_res, want_res, seq, off)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)


@pytest.mark.parametrize("min_bucket,per", [(1, 0), (3, 64), (40, 0)])
def test_emulated_sorted_statistics_pass(orc, monkeypatch, min_bucket, per):
    """k_stats_sorted (the statistics pass over the reads sorted by front trim: bucket kernels, persistent blocks, the
    shifted reduce) is what batches of >= 150 000 reads take; FPL_STATS_MIN_BUCKET forces it for a small batch.
    min_bucket 1: every front trim gets slices of its own; 3 with slices of 64 reads: several slices per bucket, rare front
    trims handed to the EXTRA pass; 40: nearly everything handed over"""
    monkeypatch.setenv("FPL_STATS_MIN_BUCKET", str(min_bucket))
    if per:
        monkeypatch.setenv("FPL_STATS_PER", str(per))
    cfg = orc.Config(abi.FplOptions.default(cut_front=1, cut_tail=1, cut_front_window=5, cut_tail_window=5, polyx=1,
                                            complexity_filter=1), synth.START_ADAPTER, synth.END_ADAPTER)
    a = synth.ont_like(60, seed=12, median_len=700, p_middle=0.1, p_polya=0.2)
    b = synth.adversarial(40, seed=13)
    reads = []
    for (s_, q_, o_) in (a, b):
        reads += [(s_[int(o_[i]):int(o_[i + 1])], q_[int(o_[i]):int(o_[i + 1])]) for i in range(len(o_) - 1)]
    seq, qual, off = synth.pack(reads)
    C = int(np.diff(off.astype(np.int64)).max()) + 1
    want_res, want_cnt = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
    got_res, got_cnt = emu.process_batch(cfg, seq, qual, off, C)
    parity.assert_results_equal(got_res, want_res, seq, off)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)


@pytest.mark.parametrize("hi_tile,group,rows,per", [(2, 4, 70, 64)])
def test_emulated_sorted_statistics_pass_slice_groups(orc, monkeypatch, hi_tile, group, rows, per):
    """k_stats_sorted: from cycle tile hi_tile - 1 on an item is a GROUP of consecutive slices of one front trim that share a
    table set and a slab while their rows fit (FPL_STATS_GROUP_ROWS: a small limit, so that groups are handed over in pieces);
    reads of 300 .. 2500 bases in slices of 64: several tiles, several slices per front trim, leaders that differ from tile to tile"""
    monkeypatch.setenv("FPL_STATS_MIN_BUCKET", "2")
    monkeypatch.setenv("FPL_STATS_PER", str(per))
    monkeypatch.setenv("FPL_STATS_HI_TILE", str(hi_tile))
    monkeypatch.setenv("FPL_STATS_GROUP", str(group))
    if rows:
        monkeypatch.setenv("FPL_STATS_GROUP_ROWS", str(rows))
    cfg = orc.Config(abi.FplOptions.default(cut_front=1, cut_tail=1, cut_front_window=5, cut_tail_window=5, polyx=1,
                                            complexity_filter=1), synth.START_ADAPTER, synth.END_ADAPTER)
    a = synth.ont_like(150, seed=22, median_len=700, sigma_len=0.7, min_len=300, max_len=2500, p_middle=0.05)
    b = synth.adversarial(30, seed=23)
    reads = []
    for (s_, q_, o_) in (a, b):
        reads += [(s_[int(o_[i]):int(o_[i + 1])], q_[int(o_[i]):int(o_[i + 1])]) for i in range(len(o_) - 1)]
    seq, qual, off = synth.pack(reads)
    C = int(np.diff(off.astype(np.int64)).max()) + 1
    want_res, want_cnt = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
    got_res, got_cnt = emu.process_batch(cfg, seq, qual, off, C)
    parity.assert_results_equal(got_res, want_res, seq, off)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)


def _reads_for_row_setup(seed, n=120):
    """reads for the row set-up of k_stats_sorted: bytes that are no bases (N, lower case, X) and U's planted where the 5-mer
    stream crosses from one lane's eight bytes into the next and from one 512-cycle tile into the next (the four bases in
    front of a tile come through a register of their own), reads that END within a few bytes of those boundaries, clean
    reads (every row takes the constant-increment path) and reads with a single odd byte (one row takes the exact path)"""
    rng = np.random.default_rng(9100 + seed)
    reads = []
    odd = np.frombuffer(b"NnacgtXU", dtype=np.uint8)
    for i in range(n):
        kind = i % 6
        if kind == 0:
            L = int(rng.choice([512, 1024, 1536])) + int(rng.integers(-5, 10))
        elif kind == 1:
            L = int(rng.integers(300, 1800))
        else:
            L = int(rng.integers(1030, 1700))
        s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, L)].copy()
        q = rng.integers(33 + 8, 33 + 45, L).astype(np.uint8)
        if kind >= 2:
            spots = []
            for base in (512, 1024, 8 * int(rng.integers(1, L // 8))):
                spots += [base + d for d in rng.integers(-6, 6, int(rng.integers(1, 4)))]
            if kind == 4:
                spots = spots[:1]
            for p_ in spots:
                if 40 <= p_ < L - 40:
                    s[p_] = ord("U") if kind == 5 else int(rng.choice(odd))
        reads.append((s, q))
    return synth.pack(reads)


@pytest.mark.parametrize("opts", [dict(), dict(cut_front=1, cut_tail=1, cut_front_window=5, cut_tail_window=5, polyx=1)])
def test_emulated_sorted_statistics_pass_row_setup(orc, monkeypatch, opts):
    """k_stats_sorted's row set-up: the 5-mer stream of a lane from its own two v_dot4 packs and its neighbour's finished one
    (lane 0: the halo register of the group of four rows), the validity test without the neighbour's bytes"""
    monkeypatch.setenv("FPL_STATS_MIN_BUCKET", "1")
    monkeypatch.setenv("FPL_STATS_PER", "64")
    cfg = orc.Config(abi.FplOptions.default(**opts), synth.START_ADAPTER, synth.END_ADAPTER)
    seq, qual, off = _reads_for_row_setup(1, n=90)
    C = int(np.diff(off.astype(np.int64)).max()) + 1
    want_res, want_cnt = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
    got_res, got_cnt = emu.process_batch(cfg, seq, qual, off, C)
    parity.assert_results_equal(got_res, want_res, seq, off)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)


def _reads_for_kmer6(seed, n=100):
    """reads for the 6-mer table of k_stats_sorted (FPL_OPT_KMER6): N's at every offset of a lane's eight bytes (a window pair
    that loses its first or its second window takes the 5-mer table), runs of N, bytes of the base classes 0 and 2 (low three
    bits 000 / 010: '@' 'H' 'X' 'p' '*' 'B' 'R' 'j' -- counted with global atomics, their rows walked byte by byte) in rows
    inside r1, across both ends of r1 (adapters at both ends, low-quality tails for the quality cut) and in reads that fail
    (qualities below the limit), in the first tile (no bases in front of lane 0) and in ragged last rows"""
    rng = np.random.default_rng(4600 + seed)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    odd = np.frombuffer(b"@HXp*BRj08", dtype=np.uint8)
    sa = np.frombuffer(synth.START_ADAPTER.encode(), np.uint8)
    ea = np.frombuffer(synth.END_ADAPTER.encode(), np.uint8)
    reads = []
    for i in range(n):
        kind = i % 8
        L = int(rng.integers(520, 2200)) if kind else int(rng.choice([512, 1024, 1032])) + int(rng.integers(-3, 4))
        s = letters[rng.integers(0, 4, L)].copy()
        q = rng.integers(33 + 12, 33 + 45, L).astype(np.uint8)
        if kind in (1, 2, 5):  # adapters at the ends: r1 starts / ends inside a row
            a = int(rng.integers(0, 25))
            s[a:a + len(sa)] = sa
            b = L - int(rng.integers(0, 25)) - len(ea)
            s[b:b + len(ea)] = ea
        if kind in (2, 6):  # low-quality ends for cut_front / cut_tail
            q[:int(rng.integers(1, 60))] = 33 + 3
            q[L - int(rng.integers(1, 60)):] = 33 + 3
        if kind == 3:  # a read that fails the quality filter: counted pre-filter only
            q[:] = rng.integers(33 + 2, 33 + 9, L).astype(np.uint8)
        # N's: one per offset class, plus a run
        for j in range(int(rng.integers(2, 9))):
            p_ = 8 * int(rng.integers(4, L // 8 - 4)) + (j % 8)
            s[p_] = ord("N")
        if kind in (4, 5):
            p_ = int(rng.integers(60, L - 60))
            s[p_:p_ + int(rng.integers(2, 12))] = ord("N")
        if kind in (1, 4, 7):  # two N's a few bases apart: a lane with two windows that count alone
            p_ = 8 * int(rng.integers(8, L // 8 - 8)) + int(rng.integers(0, 8))
            s[p_] = ord("N")
            s[p_ + int(rng.choice([6, 7, 8, 9]))] = ord("N")
        if kind in (5, 6, 7, 0):  # bytes without cells in LDS, anywhere: the ends included
            for p_ in list(rng.integers(0, L, int(rng.integers(1, 5)))) + [0, L - 1][:int(rng.integers(0, 3))]:
                s[int(p_)] = int(rng.choice(odd))
        reads.append((s, q))
    return synth.pack(reads)


@pytest.mark.parametrize("opts", [dict(), dict(cut_front=1, cut_tail=1, cut_front_window=5, cut_tail_window=5, polyx=1, complexity_filter=1)])
@pytest.mark.parametrize("per", [0, 64])
def test_emulated_sorted_statistics_pass_kmer6(orc, monkeypatch, opts, per):
    """k_stats_sorted with the 6-mer table: pairs of 5-mer windows as one update, windows that count alone, rows with bytes of
    the base classes 0 and 2 (tests/test_kernels_emu.py::_reads_for_kmer6)"""
    monkeypatch.setenv("FPL_STATS_MIN_BUCKET", "1")
    if per:
        monkeypatch.setenv("FPL_STATS_PER", str(per))
    cfg = orc.Config(abi.FplOptions.default(**opts), synth.START_ADAPTER, synth.END_ADAPTER)
    seq, qual, off = _reads_for_kmer6(1 + per, n=96)
    C = int(np.diff(off.astype(np.int64)).max()) + 1
    want_res, want_cnt = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
    got_res, got_cnt = emu.process_batch(cfg, seq, qual, off, C)
    parity.assert_results_equal(got_res, want_res, seq, off)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)


def test_emulated_kernels_multi_adapter(orc):
    fasta = ["ACGTTGCAATGCCGTA", "TTGACCAGTAGGCATCAGGATCCA", "GATTACA", "CCCCGGGGAAAATTTTCCCCGGGGAAAATTTTCCCCGGGGAAAATTTTCCCCGGGGAAAATTTTCCCCGGGG"]
    cfg = orc.Config(abi.FplOptions.default(), synth.START_ADAPTER, synth.END_ADAPTER, fasta)
    seq, qual, off = synth.adversarial(70, seed=21, fasta=fasta)
    C = int(np.diff(off.astype(np.int64)).max()) + 1
    want_res, want_cnt = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
    got_res, got_cnt = emu.process_batch(cfg, seq, qual, off, C)
    parity.assert_results_equal(got_res, want_res, seq, off)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)


def test_emulated_kernels_empty_batch(orc):
    cfg = orc.Config(abi.FplOptions.default(), synth.START_ADAPTER, synth.END_ADAPTER)
    off = np.zeros(1, np.uint64)
    res, cnt = emu.process_batch(cfg, np.zeros(0, np.uint8), np.zeros(0, np.uint8), off, 8)
    assert len(res) == 0 and not cnt.any()


def test_emulated_kernels_byte_scan_fallback(orc):
    """adapters with bytes outside ACGT (or longer than 64) take the byte-wise scan in k_scan"""
    cfg = orc.Config(abi.FplOptions.default(), "AAGGATTCATTCCNACGGTAACAC", "GTGTTACCGTNGGAATGAATCCTT")
    seq, qual, off = synth.adversarial(80, seed=5, start_adapter="AAGGATTCATTCCNACGGTAACAC",
                                       end_adapter="GTGTTACCGTNGGAATGAATCCTT")
    C = int(np.diff(off.astype(np.int64)).max()) + 1
    want_res, want_cnt = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
    got_res, got_cnt = emu.process_batch(cfg, seq, qual, off, C)
    parity.assert_results_equal(got_res, want_res, seq, off)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)


@pytest.mark.parametrize("lens,ed_max,seed", [((30, 45), 0.25, 1), ((23, 24, 31, 32, 33, 64), 0.25, 2), ((23, 40, 64), 0.4, 3),
                                              ((32, 64), 1.0, 4), ((22, 23, 30), 0.25, 5), ((23, 64), 0.1, 6), ((30, 45), 0.0, 7)])
def test_emulated_fasta_filter_packed_scores(orc, lens, ed_max, seed):
    """k_trim_ends<2>: adapter sets whose every adapter has 23 bases or more take the packed-score form of the lane-per-adapter
    filter (fasta_may_trim32p).  The emulator build compares each of its verdicts with the plain form's and aborts on a
    difference; the records and counters are the oracle's.  ed_max 1.0 puts the thresholds beyond the packed fields' bias (plain
    form), the set with a 22-mer takes the plain form as a whole; adapters planted at both ends, whole and cut short"""
    rng = np.random.default_rng(1000 + seed)
    n_ad = 9 if seed not in (2, 7) else 70  # (70: two groups of 64 lanes)
    fasta = ["".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(lens[0], lens[-1] + 1) if len(lens) == 2 else rng.choice(lens))))
             for _ in range(n_ad)]
    start, end = fasta[0], synth.revcomp(fasta[0])
    cfg = orc.Config(abi.FplOptions.default(ed_max=ed_max, trimming_extension=5), start, end, fasta)
    seq, qual, off = synth.adversarial(150, seed=40 + seed, start_adapter=start, end_adapter=end, fasta=fasta)
    reads = [(seq[int(off[i]):int(off[i + 1])], qual[int(off[i]):int(off[i + 1])]) for i in range(len(off) - 1)]
    for k in range(120):  # an adapter (whole, or its partial pattern and a few bases more) with up to three errors at an end of a random read
        ad = fasta[int(rng.integers(0, len(fasta)))]
        body = "".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(40, 400))))
        piece = list(ad if k % 3 else (ad[-(16 + k % 9):] if k % 2 else ad[:16 + k % 9]))
        for _ in range(k % 4):
            piece[int(rng.integers(0, len(piece)))] = "ACGT"[int(rng.integers(0, 4))]
        piece = "".join(piece)
        r = np.frombuffer(((piece + body) if k % 2 else (body + piece)).encode(), np.uint8)
        reads.append((r, np.full(len(r), 70, np.uint8)))
    seq, qual, off = synth.pack(reads)
    C = int(np.diff(off.astype(np.int64)).max()) + 1
    want_res, want_cnt = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
    got_res, got_cnt = emu.process_batch(cfg, seq, qual, off, C)
    parity.assert_results_equal(got_res, want_res, seq, off)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)


@pytest.mark.parametrize("la,lb", [(32, 17), (16, 31)])
def test_emulated_batched_trim_kernel_adapter_lengths(orc, la, lb):
    """the lane-per-read window scans of k_trim_ends_batched (ham_scan_lanes) slide 32 one-hot nibbles: adapters at both ends of
    the 16..32 range, planted near both read ends by the adversarial generator"""
    rng = np.random.default_rng(la * 100 + lb)
    start = "".join("ACGT"[i] for i in rng.integers(0, 4, la))
    end = "".join("ACGT"[i] for i in rng.integers(0, 4, lb))
    cfg = orc.Config(abi.FplOptions.default(cut_front=1, cut_tail=1, polyx=1), start, end)
    seq, qual, off = synth.adversarial(150, seed=la + lb, start_adapter=start, end_adapter=end)
    C = int(np.diff(off.astype(np.int64)).max()) + 1
    want_res, want_cnt = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
    got_res, got_cnt = emu.process_batch(cfg, seq, qual, off, C)
    parity.assert_results_equal(got_res, want_res, seq, off)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)


@pytest.mark.parametrize("la,lb,nfa", [(40, 50, 0), (64, 33, 3)])
def test_emulated_batched_trim_kernel_adapters_beyond_32_bases(orc, la, lb, nfa):
    """k_trim_ends_batched<.., 8, ..>: command-line adapters of 33..64 bases (64-base one-hot windows, 64-column confirmations), alone
    and as the front end of a FASTA chain (k_trim_ends<2> then starts from the ReadState records).  Reads shorter than the adapter
    make a trim return a NEGATIVE count (pos = min(pos + ext, rlen - alen)): only reads whose total is positive are booked"""
    rng = np.random.default_rng(la * 100 + lb)
    start = "".join("ACGT"[i] for i in rng.integers(0, 4, la))
    end = "".join("ACGT"[i] for i in rng.integers(0, 4, lb))
    fasta = ["".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.choice([16, 24, 40, 64])))) for _ in range(nfa)]
    cfg = orc.Config(abi.FplOptions.default(cut_front=1, polyx=1, trimming_extension=10), start, end, fasta)
    seq, qual, off = synth.adversarial(120, seed=la + lb, start_adapter=start, end_adapter=end, fasta=fasta)
    reads = [(seq[int(off[i]):int(off[i + 1])], qual[int(off[i]):int(off[i + 1])]) for i in range(len(off) - 1)]
    for k in range(30):  # the start adapter's last 16 bases and a few more: a partial match in a read shorter than alen - 16
        r = np.frombuffer((start[-16:] + "".join("ACGT"[i] for i in rng.integers(0, 4, 2 + k % 6))).encode(), np.uint8)
        reads.append((r, np.full(len(r), 70, np.uint8)))
    seq, qual, off = synth.pack(reads)
    C = int(np.diff(off.astype(np.int64)).max()) + 1
    want_res, want_cnt = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
    got_res, got_cnt = emu.process_batch(cfg, seq, qual, off, C)
    parity.assert_results_equal(got_res, want_res, seq, off)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)
    if la >= 40:
        assert (want_res["r1_len"][-30:] == 0).any()  # (Read::trimFront with a negative count erased the read)


def _reads_with_long_scans(seed, n=40):
    """reads whose trimAndCut / polyX scans run long: low-quality heads and tails of 100..500 bases (a sliding window finds
    nothing good for hundreds of positions), poly-A tails and N runs beyond the lane-per-read forms' iteration cap, and a few
    reads that are ALL low quality or all one base"""
    rng = np.random.default_rng(seed)
    reads = []
    for i in range(n):
        L = int(rng.integers(700, 2500))
        sq = synth._ACGT[rng.integers(0, 4, L)].astype(np.uint8)
        ql = (np.clip(np.round(rng.normal(24, 5, L)), 2, 50) + 33).astype(np.uint8)
        k = i % 8
        if k == 0:
            ql[:int(rng.integers(100, 500))] = 35  # Q2 head
        elif k == 1:
            ql[-int(rng.integers(100, 500)):] = 35  # Q2 tail
        elif k == 2:
            sq[-int(rng.integers(170, 400)):] = ord("A")  # poly-A beyond the cap
        elif k == 3:
            a = int(rng.integers(170, 300))
            ql[:a] = 35
            sq[a - 3:a + int(rng.integers(165, 260))] = ord("N")  # a long N run behind the cut point
        elif k == 4:
            ql[:] = 36  # nothing good anywhere
        elif k == 5:
            sq[:] = ord("G")
        elif k == 6:
            b = int(rng.integers(170, 300))
            ql[-b:] = 35
            sq[-(b + int(rng.integers(165, 240))):-(b - 3)] = ord("N")
        reads.append((sq, ql))
    return synth.pack(reads)


@pytest.mark.parametrize("seed", [1, 2])
def test_emulated_batched_trim_kernel_long_scans_fall_back(orc, seed):
    """the lane-per-read trimAndCut / polyX of k_trim_ends_batched give up after LANE_SCAN_CAP rounds and hand the read to the
    wave-per-read forms: same results either way"""
    seq, qual, off = _reads_with_long_scans(seed)
    cfg = orc.Config(abi.FplOptions.default(cut_front=1, cut_tail=1, cut_front_window=5, cut_tail_window=4, polyx=1,
                                            n_base_percent_limit=60), synth.START_ADAPTER, synth.END_ADAPTER)
    C = int(np.diff(off.astype(np.int64)).max()) + 1
    want_res, want_cnt = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
    got_res, got_cnt = emu.process_batch(cfg, seq, qual, off, C)
    parity.assert_results_equal(got_res, want_res, seq, off)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)
    assert (want_res["dropped"] != 0).any() and (want_res["r1_len"][want_res["dropped"] == 0] < 600).any()


def _reads_ending_anywhere(seed, small=False):
    """every length modulo the scan's 32-byte chunks and 1984-byte tiles, tails of N / of one base / of lower-case letters: the
    ragged last tile of k_scan pads itself with the read's last byte"""
    rng = np.random.default_rng(seed)
    lens = list(range(1, 140)) + [1984 + d for d in range(-34, 35)] + [2 * 1984 + d for d in (-33, -32, -31, -1, 0, 1, 31, 32, 33)] + [2048, 4031, 4032]
    if small:  # (the emulator's share: every offset of a chunk once, the tile boundaries)
        lens = list(range(1, 67)) + [1984 + d for d in (-33, -32, -31, -2, -1, 0, 1, 2, 31, 32, 33)] + [2 * 1984 - 1, 2 * 1984 + 1, 2048]
    reads = []
    for i, L in enumerate(lens):
        sq = synth._ACGT[rng.integers(0, 4, L)].astype(np.uint8)
        ql = (np.clip(np.round(rng.normal(26, 6, L)), 2, 50) + 33).astype(np.uint8)
        r = int(rng.integers(0, min(L, 45) + 1))
        k = i % 5
        if k == 0 and r:
            sq[L - r:] = ord("N")
        elif k == 1 and r:
            sq[L - r:] = ord("G")
        elif k == 2 and r:
            sq[L - r:] = np.frombuffer(b"acgtn", dtype=np.uint8)[rng.integers(0, 5, r)]
        elif k == 3:
            sq[L - 1] = ord("N")
        reads.append((sq, ql))
    return synth.pack(reads)


@pytest.mark.parametrize("opts", [dict(), dict(complexity_filter=1, n_base_percent_limit=60), dict(adapter_enabled=0, complexity_filter=1)])
def test_emulated_scan_ragged_last_tiles(orc, opts):
    seq, qual, off = _reads_ending_anywhere(5, small=True)
    cfg = orc.Config(abi.FplOptions.default(**opts), synth.START_ADAPTER if opts.get("adapter_enabled", 1) else "",
                     synth.END_ADAPTER if opts.get("adapter_enabled", 1) else "")
    C = int(np.diff(off.astype(np.int64)).max()) + 1
    want_res, want_cnt = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
    got_res, got_cnt = emu.process_batch(cfg, seq, qual, off, C)
    parity.assert_results_equal(got_res, want_res, seq, off)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)


def _reads_for_pair_packing(seed, small=False):
    """k_scan's pair packing: the head of a chunk's NEXT read rides in the lanes a read's ragged last tile leaves empty.  Pairs
    (A, B) back to back: A's last tile needs 1 .. 62 lanes (a head takes what is left when >= 12 lanes are), B is just short of /
    just long enough for / far beyond what the head needs, B's head holds N runs, lower-case letters (the byte-wise tile), the
    middle adapters at its start, across the lane the head ends in and just behind it, B is dropped by trimAndCut, A has a long
    trimmed end (no hosting), reads of several tiles whose own tiles then start 32 * (62 - hb) bytes in."""
    rng = np.random.default_rng(seed)
    sa = np.frombuffer(synth.START_ADAPTER.encode(), np.uint8)
    ea = np.frombuffer(synth.END_ADAPTER.encode(), np.uint8)

    def read(L, mu=26.0):
        sq = synth._ACGT[rng.integers(0, 4, L)].astype(np.uint8)
        ql = (np.clip(np.round(rng.normal(mu, 6, L)), 2, 50) + 33).astype(np.uint8)
        return sq, ql

    rems = [1, 31, 32, 33, 500, 1023, 1024, 1025, 1599, 1600, 1601, 1983] if small else \
        [1, 2, 31, 32, 33, 63, 64, 65, 300, 500, 777, 1000, 1023, 1024, 1025, 1300, 1567, 1568, 1569, 1599, 1600, 1601, 1602, 1900, 1983, 1984]
    reads = []
    k = 0
    for rem in rems:
        la = (rem + 31) // 32
        need = 32 * (64 - la) + 64  # the shortest r1 of B that gets a head
        for dB in ((-1, 0, 1, 700, 2500) if not small else (-1, 0, 2500)):
            LA = rem + (1984 if (k % 3 == 0) else 0)
            a_s, a_q = read(LA)
            LB = max(40, need + dB)
            b_s, b_q = read(LB)
            hb_bytes = 32 * (62 - la)
            v = k % 8
            if v == 1 and hb_bytes > 40:
                b_s[5:5 + 20] = ord("N")
                b_s[hb_bytes - 3:hb_bytes + 3] = ord("N")
            elif v == 2 and hb_bytes > 40:
                b_s[hb_bytes // 2] = ord("a")  # one lower-case letter in the head: the whole tile takes the byte-wise sums
            elif v == 3 and hb_bytes > 100 and LB > hb_bytes + 100:
                p = hb_bytes - 15  # a middle adapter across the boundary between the head and B's own first tile
                b_s[p:p + len(sa)] = sa
            elif v == 4 and LB > 200:
                b_s[60:60 + len(ea)] = ea  # ... and one inside the head
            elif v == 5 and hb_bytes > 100 and LB > hb_bytes + 100:
                p = hb_bytes + 1
                b_s[p:p + len(ea)] = ea  # ... and one that starts right behind it
            elif v == 6:
                b_q[:] = 33 + 2  # everything below the cut threshold: with cut_front / cut_tail B is dropped
            elif v == 7 and LA > 400:
                a_q[:200] = 33 + 2  # A loses 200 bases at the front with cut_front: a trimmed end beyond the prefetch -> no hosting
            reads += [(a_s, a_q), (b_s, b_q)]
            k += 1
    # a run of reads that all host and are hosted (every read's tiles start inside it)
    for L in ([2100, 3100, 2500, 5000, 2048, 2300] if small else [2100, 3100, 2500, 5000, 2048, 2300, 4100, 6100, 2200, 9000, 2400]):
        reads.append(read(L))
    return synth.pack(reads)


@pytest.mark.parametrize("opts,chunk", [(dict(cut_front=1, cut_tail=1, complexity_filter=1, n_base_percent_limit=60), "3"), (dict(), "")])
def test_emulated_scan_pair_packing(orc, monkeypatch, opts, chunk):
    """(chunk "": the built-in chunk rule -- one read per dequeue for a batch this small, the plain scan)"""
    if chunk:
        monkeypatch.setenv("FPL_SCAN_CHUNK", chunk)
    else:
        monkeypatch.delenv("FPL_SCAN_CHUNK", raising=False)
    seq, qual, off = _reads_for_pair_packing(7, small=True)
    cfg = orc.Config(abi.FplOptions.default(**opts), synth.START_ADAPTER, synth.END_ADAPTER)
    C = int(np.diff(off.astype(np.int64)).max()) + 1
    want_res, want_cnt = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
    got_res, got_cnt = emu.process_batch(cfg, seq, qual, off, C)
    parity.assert_results_equal(got_res, want_res, seq, off)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)
    if opts.get("cut_front"):
        assert (want_res["dropped"] != 0).any()
    assert (want_res["n_frag"] == 2).any()  # (the planted middle adapters split their reads)


REPEAT_START = "GTCAGTTACGTATTGC" + "AC" * 8  # (the start trim's partial pattern = the LAST 16 bases)
REPEAT_END = "TG" * 8 + "AGCAATACGTAACTGA"   # (the end trim's = the FIRST 16)


def _reads_with_partial_adapters(seed, start, end, n=96):
    """reads for the partial-pattern searches of the end trims: noisy truncated copies of the adapters at the very ends (the last
    10..25 bases of the start adapter in front, the first 10..25 of the end adapter behind), several copies in a row, runs of
    the adapters' 16-base patterns' repeat unit (dozens of candidate windows per read), reads shorter than the 200-base
    windows, and plain reads"""
    rng = np.random.default_rng(seed)
    sa = np.frombuffer(start.encode(), dtype=np.uint8)
    ea = np.frombuffer(end.encode(), dtype=np.uint8)

    def noisy(x, err):
        x = x.copy()
        hit = rng.random(len(x)) < err
        x[hit] = synth._ACGT[rng.integers(0, 4, int(hit.sum()))]
        return x

    reads = []
    for i in range(n):
        L = int(rng.integers(40, 260)) if i % 6 == 5 else int(rng.integers(300, 1500))
        sq = synth._ACGT[rng.integers(0, 4, L)].astype(np.uint8)
        ql = (np.clip(np.round(rng.normal(24, 5, L)), 2, 50) + 33).astype(np.uint8)
        k = i % 8
        if k in (0, 1, 4):  # truncated start adapter(s) at the head
            at = int(rng.integers(0, 30))
            for _ in range(1 + (k == 4) * int(rng.integers(1, 4))):
                t = noisy(sa[-int(rng.integers(10, 26)):], float(rng.choice([0.0, 0.08, 0.2])))
                if at + len(t) < L:
                    sq[at:at + len(t)] = t
                at += len(t) + int(rng.integers(0, 12))
        if k in (1, 2, 4):  # truncated end adapter(s) at the tail
            at = L - int(rng.integers(0, 30))
            for _ in range(1 + (k == 4) * int(rng.integers(1, 4))):
                t = noisy(ea[:int(rng.integers(10, 26))], float(rng.choice([0.0, 0.08, 0.2])))
                if at - len(t) > 0:
                    sq[at - len(t):at] = t
                at -= len(t) + int(rng.integers(0, 12))
        if k == 3:  # runs of the patterns' repeat units at both ends
            a, b = int(rng.integers(30, 190)), int(rng.integers(30, 190))
            if a + b < L:
                sq[:a] = noisy(np.resize(sa[-16:], a), 0.06)
                sq[L - b:] = noisy(np.resize(ea[:16], b), 0.06)
        if k == 7 and L >= 300:  # the patterns with three bases inserted in the middle, several times over: columns the search
            # variant cannot rule out (score 3) whose exact windows do not qualify -- more candidates than the lanes' cap
            for side in (0, 1):
                pat = sa[-16:] if side == 0 else ea[:16]
                blocks = np.concatenate([np.concatenate([pat[:8], synth._ACGT[rng.integers(0, 4, 3)], pat[8:]]) for _ in range(7)])
                if side == 0:
                    sq[2:2 + len(blocks)] = blocks
                else:
                    sq[L - 2 - len(blocks):L - 2] = blocks
        reads.append((sq, ql))
    return synth.pack(reads)


@pytest.mark.parametrize("seed,start,end", [(1, synth.START_ADAPTER, synth.END_ADAPTER), (2, REPEAT_START, REPEAT_END),
                                            (3, REPEAT_START, synth.END_ADAPTER)])
def test_emulated_batched_trim_kernel_partial_pattern_searches(orc, seed, start, end):
    """k_trim_ends_batched's partial-pattern searches with lane = read (partial16_candidates / partial16_resolve_lanes): a few
    candidate columns per read as a rule; reads with more than PART_CAND_CAP of them (repeats) fall back to the wave-per-read
    search -- same results either way"""
    seq, qual, off = _reads_with_partial_adapters(seed, start, end)
    cfg = orc.Config(abi.FplOptions.default(), start, end)
    C = int(np.diff(off.astype(np.int64)).max()) + 1
    want_res, want_cnt = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
    got_res, got_cnt = emu.process_batch(cfg, seq, qual, off, C)
    parity.assert_results_equal(got_res, want_res, seq, off)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)
    assert (want_res["r1_start"] > 0).sum() > 10 and (want_res["r1_start"] + want_res["r1_len"] < np.diff(off.astype(np.int64))).sum() > 10


def test_emulated_long_reads_split_by_middle_adapters(orc):
    """reads beyond REDO_LONG (16 kb) with a middle adapter go to the FRONT of the REDO list, the others to its far end"""
    seq, qual, off = synth.ont_like(7, seed=3, median_len=19000, sigma_len=0.25, p_middle=0.9)
    cfg = orc.Config(abi.FplOptions.default(cut_front=1, cut_tail=1, polyx=1, complexity_filter=1), synth.START_ADAPTER, synth.END_ADAPTER)
    C = int(np.diff(off.astype(np.int64)).max()) + 1
    want_res, want_cnt = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
    got_res, got_cnt = emu.process_batch(cfg, seq, qual, off, C)
    parity.assert_results_equal(got_res, want_res, seq, off)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)
    split = want_res["n_frag"] == 2
    assert (split & (want_res["r1_len"] > 16384)).any() and (split & (want_res["r1_len"] <= 16384)).any()


def test_emulated_kernels_adapters_of_33_to_64_bases(orc):
    """A / C / G / T adapters beyond 32 bases: the bit-sliced scan with seven count planes in k_scan, and k_resolve's edit
    distances on 64-bit columns, one window per lane (lev_lanes64_acgt)"""
    rng = np.random.default_rng(77)
    start = "".join("ACGT"[i] for i in rng.integers(0, 4, 45))
    end = "".join("ACGT"[i] for i in rng.integers(0, 4, 38))
    cfg = orc.Config(abi.FplOptions.default(ed_max=0.25), start, end)
    seq, qual, off = synth.adversarial(120, seed=9, start_adapter=start, end_adapter=end)
    C = int(np.diff(off.astype(np.int64)).max()) + 1
    want_res, want_cnt = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
    got_res, got_cnt = emu.process_batch(cfg, seq, qual, off, C)
    parity.assert_results_equal(got_res, want_res, seq, off)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)
    assert (want_res["n_frag"] == 2).any()  # some read is split by a middle adapter


@pytest.mark.parametrize("acc", [0, 64])
def test_emulated_stats_extra_pass_accumulates_and_overflows(orc, monkeypatch, acc):
    """k_stats<EXTRA>: one block per tile walks many short slices of the split-fragment list into ONE slab;
    with a tiny accumulation limit it must take the atomic overflow path -- same tables either way"""
    monkeypatch.setenv("FPL_STATS_EXTRA_PER", "32")
    monkeypatch.setenv("FPL_STATS_EXTRA_BLOCKS", "1")
    if acc:
        monkeypatch.setenv("FPL_STATS_EXTRA_ACC", str(acc))
    cfg = orc.Config(abi.FplOptions.default(), synth.START_ADAPTER, synth.END_ADAPTER)
    seq, qual, off = synth.ont_like(90, seed=8, median_len=500, p_middle=0.9, p_polya=0.0)
    C = int(np.diff(off.astype(np.int64)).max()) + 1
    want_res, want_cnt = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
    assert 2 * int((want_res["n_frag"] == 2).sum()) > 3 * 32  # more than three slices of EXTRA items
    got_res, got_cnt = emu.process_batch(cfg, seq, qual, off, C)
    parity.assert_results_equal(got_res, want_res, seq, off)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)


def _lowq_batch(n, seed, with_adapters=True):
    """reads whose qualities dip far below the --break / --mask thresholds in stretches"""
    rng = np.random.default_rng(seed)
    seq, qual, off = synth.ont_like(n, seed=seed, median_len=900, p_middle=0.3, p_polya=0.1)
    qual = qual.copy()
    for i in range(n):
        a, b = int(off[i]), int(off[i + 1])
        pos = a + int(rng.integers(0, 200))
        while pos < b:
            run = int(rng.integers(10, 150))
            if rng.random() < 0.4:
                qual[pos:min(b, pos + run)] = np.clip(np.round(rng.normal(6, 3, min(b, pos + run) - pos)), 2, 40) + 33
            pos += run + int(rng.integers(20, 300))
    return seq, qual, off


@pytest.mark.parametrize("be,me", [(1, 0), (0, 1), (1, 1)])
def test_emulated_break_mask(orc, be, me):
    """--break / --mask: k_break_mask + the cycle-offset / masked items of k_stats<EXTRA> against the oracle"""
    # (the reference masks from the first low-quality window to the end of the read, see orc_detect_low_quality_regions:
    #  generous N / quality limits keep some masked reads passing, so that their pieces reach the post-filter tables)
    opt = abi.FplOptions.default(cut_front=1, cut_tail=1, polyx=1, complexity_filter=1, break_enabled=be, break_window=30,
                                 break_quality=12, mask_enabled=me, mask_window=15, mask_quality=13,
                                 n_base_percent_limit=95, unqualified_percent_limit=90, complexity_percent=5)
    cfg = orc.Config(opt, synth.START_ADAPTER, synth.END_ADAPTER)
    seq, qual, off = _lowq_batch(40, seed=31 + 2 * be + me)
    C = int(np.diff(off.astype(np.int64)).max()) + 1
    want_res, want_cnt, want_f, want_r = orc.process_batch_ex(cfg, seq, qual, off, max_cycles=C)
    got_res, got_cnt, got_f, got_r = emu.process_batch(cfg, seq, qual, off, C, with_fragments=True)
    parity.assert_fragments_equal(got_f, got_r, want_f, want_r)
    parity.assert_results_equal(got_res, want_res, seq, off)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)
    assert (want_f["break_no"] > 0).any() == bool(be) and (want_f["region_count"] > 0).any() == bool(me)
    assert (want_f["code"] == abi.FPL_PASS_FILTER).any() and (want_f["kind"] > 0).any()
    if me:  # a masked read that passes: its unmasked / all-N pieces feed the post-filter tables at their own cycles
        assert ((want_f["region_count"] > 0) & (want_f["code"] == abi.FPL_PASS_FILTER)).any()


def test_emulated_mask_piece_starting_just_before_a_tile_boundary(orc):
    """--mask: the unmasked piece behind a masked stretch is counted post-filter from the cycle it starts at; when that
    cycle lies 1..3 bases in front of a 512-cycle tile boundary, the first 5-mer windows of the next tile reach back
    across it (read 175 of fuzz case 13107: body starts at cycle 1021)"""
    from tests.test_gpu_parity import random_case
    okw, start, end, seq, qual, off = random_case(13107)
    a, b = int(off[175]), int(off[176])
    seq, qual, off1 = seq[a:b], qual[a:b], np.array([0, b - a], np.uint64)
    cfg = orc.Config(abi.FplOptions.default(**okw), start, end)
    want_res, want_cnt, want_f, want_r = orc.process_batch_ex(cfg, seq, qual, off1, max_cycles=b - a)
    assert len(want_r) == 1 and (int(want_r[0]["start"]) + int(want_r[0]["len"]) - int(want_f[0]["start"])) % 512 in (509, 510, 511)
    got = emu.process_batch(cfg, seq, qual, off1, b - a, with_fragments=True)
    parity.assert_counters_equal(got[1], want_cnt, b - a, cfg.n_adapters)


def _random_fasta_case(seed):
    from tests.test_gpu_parity import random_case
    okw, start, end, _, _, _ = random_case(seed)
    rng = np.random.default_rng(77000 + seed)
    lens = [6, 7, 12, 15, 16, 17, 24, 31, 32, 33, 45, 63, 64, 65, 100, 199, 200, 201, 250]
    fasta = ["".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.choice(lens)))) for _ in range(int(rng.integers(0, 5)))]
    if rng.random() < 0.3:
        start = "".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.choice(lens))))
    if rng.random() < 0.3:
        end = "".join("ACGTN"[i] for i in rng.integers(0, 5, int(rng.choice(lens))))
    a = synth.adversarial(int(os.environ.get("FPL_EMU_FUZZ_READS", "60")), seed=seed,
                          start_adapter=start or synth.START_ADAPTER, end_adapter=end or synth.END_ADAPTER, fasta=fasta)
    # a few longer reads (several cycle tiles / scan tiles, middle adapters) with the command-line pair or one FASTA adapter at the
    # ends (FPL_EMU_FUZZ_LONG=0 leaves them out)
    ends = (fasta[0], synth.revcomp(fasta[0])) if (fasta and rng.random() < 0.5) else (start or synth.START_ADAPTER, end or synth.END_ADAPTER)
    b = synth.ont_like(4 if os.environ.get("FPL_EMU_FUZZ_LONG", "1") != "0" else 0, seed=seed, median_len=int(rng.choice([600, 1500, 4000])), start_adapter=ends[0][:120],
                       end_adapter=ends[1].replace("N", "A")[:120], p_middle=0.4, p_polya=0.2)
    reads = []
    for (s_, q_, o_) in (a, b):
        reads += [(s_[int(o_[i]):int(o_[i + 1])], q_[int(o_[i]):int(o_[i + 1])]) for i in range(len(o_) - 1)]
    seq, qual, off = synth.pack(reads)
    return okw, start, end, fasta, seq, qual, off


# 408: a 7-base FASTA adapter whose partial match sits at the last of its 193 end positions (beyond three rounds of 64)
@pytest.mark.parametrize("seed", list(range(int(os.environ.get("FPL_EMU_FUZZ_FROM", "0")),
                                            int(os.environ.get("FPL_EMU_FUZZ_FROM", "0")) + int(os.environ.get("FPL_EMU_FUZZ", "40")))) + [408])
def test_emulated_random_fasta_cases(orc, seed):
    """random options x random command-line / FASTA adapter sets (every length class, so all instantiations of
    k_trim_ends / k_scan) on the emulator, 40 seeds by default (a quarter of a second each since the emulator's lanes are fibers);
    FPL_EMU_FUZZ=<n> FPL_EMU_FUZZ_FROM=<first> widen it for a soak on the CPU"""
    okw, start, end, fasta, seq, qual, off = _random_fasta_case(seed)
    cfg = orc.Config(abi.FplOptions.default(**okw), start, end, fasta)
    C = max(1, int(np.diff(off.astype(np.int64)).max()))
    if okw["break_enabled"] or okw["mask_enabled"]:
        want_res, want_cnt, want_f, want_r = orc.process_batch_ex(cfg, seq, qual, off, max_cycles=C)
        got = emu.process_batch(cfg, seq, qual, off, C, with_fragments=True)
        parity.assert_fragments_equal(got[2], got[3], want_f, want_r)
    else:
        want_res, want_cnt = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
        got = emu.process_batch(cfg, seq, qual, off, C)
    parity.assert_results_equal(got[0], want_res, seq, off)
    parity.assert_counters_equal(got[1], want_cnt, C, cfg.n_adapters)


def test_emulated_rna_batch(orc):
    """direct-RNA data: U for T in reads and adapters (U adapters are not ACGT-only: the byte-wise scan; U counts as a
    valid base for the 5-mers, Stats::base2val)"""
    cfg = orc.Config(abi.FplOptions.default(cut_front=1, polyx=1, complexity_filter=1), synth.START_ADAPTER.replace("T", "U"),
                     synth.END_ADAPTER.replace("T", "U"))
    seq, qual, off = synth.ont_like(14, seed=4, median_len=900, p_middle=0.3, p_polya=0.2)
    seq = seq.copy()
    seq[seq == ord("T")] = ord("U")
    C = int(np.diff(off.astype(np.int64)).max())
    want_res, want_cnt = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
    got_res, got_cnt = emu.process_batch(cfg, seq, qual, off, C)
    parity.assert_results_equal(got_res, want_res, seq, off)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)
    assert int(abi.CountersView(want_cnt, C, 2).post.kmer.sum()) > 5000
