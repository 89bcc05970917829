"""The oracle's adapter code (oracle/fpl_oracle.c: orc_search_adapter, orc_trim_start / _end, orc_find_middle) against a second,
independently written reading of src/adaptertrimmer.cpp (tests/second_reading.py: numpy whole-array forms, a textbook edit
distance) on seeded cases -- ties between windows, adapters longer than the read, empty adapters, reads shorter than the 16-base
pattern, N / lower-case bytes, every ed_max.  The reference's translation unit cannot be compiled here (Google Highway), so two
readings that agree are what stands in for it; the reference's own four KATs are replayed on both."""
import numpy as np
import pytest

from fastplong_amd import synth
from tests import second_reading as sr

ED_MAX = [0.0, 0.1, 0.25, 0.3, 0.4, 0.5]


def _rnd(rng, n, alphabet=b"ACGT"):
    return bytes(np.frombuffer(alphabet, np.uint8)[rng.integers(0, len(alphabet), int(n))])


def _mutated(rng, ad, err):
    return bytes(synth._mutate(rng, np.frombuffer(ad, np.uint8), err)) if len(ad) else b""


def _adapter(rng):
    k = int(rng.integers(0, 10))
    if k == 0:
        return b""
    if k == 1:
        return _rnd(rng, rng.integers(1, 8))
    if k == 2:
        return _rnd(rng, 1) * int(rng.integers(4, 40))  # homopolymer: every window ties
    if k == 3:
        return _rnd(rng, rng.integers(2, 5)) * int(rng.integers(3, 12))  # short period: ties between shifted windows
    if k == 4:
        return _rnd(rng, rng.integers(16, 70), b"ACGTN")
    if k == 5:
        return _rnd(rng, rng.integers(60, 260))
    return _rnd(rng, rng.choice([15, 16, 17, 24, 24, 31, 32, 33, 45, 64]))


def _read_with(rng, ad, rlen_max=700):
    """a read that holds 0..3 noisy / truncated / exact copies of `ad` (so that several windows tie), N runs, lower case"""
    L = int(rng.choice([0, 1, 5, 15, 16, 17, 31, 40, 199, 200, 201, 216, 217])) if rng.random() < 0.3 else int(rng.integers(0, rlen_max))
    if rng.random() < 0.15 and len(ad):
        L = int(rng.integers(0, len(ad) + 2))  # around alen: alen > rlen, alen == rlen, alen == rlen - 1
    body = bytearray(_rnd(rng, L))
    if rng.random() < 0.15 and len(ad):
        unit = ad[:max(1, len(ad) // 2)]
        body = bytearray((unit * (L // len(unit) + 1))[:L])  # the adapter's own period all over the read
    for _ in range(int(rng.integers(0, 4))):
        if not len(ad) or not L:
            break
        c = _mutated(rng, ad, float(rng.choice([0.0, 0.0, 0.05, 0.1, 0.2, 0.35])))
        if rng.random() < 0.35 and len(c) > 2:
            cut = int(rng.integers(1, len(c)))
            c = c[cut:] if rng.random() < 0.5 else c[:cut]
        where = rng.random()
        at = int(rng.integers(0, 30)) if where < 0.35 else (max(0, L - len(c) - int(rng.integers(0, 30))) if where < 0.7 else int(rng.integers(0, L)))
        body[at:at + len(c)] = c
        body = body[:L] if rng.random() < 0.8 else body
    if rng.random() < 0.1 and len(body):
        a = int(rng.integers(0, len(body)))
        body[a:a + int(rng.integers(1, 20))] = b"N" * min(len(body) - a, int(rng.integers(1, 20)))
    if rng.random() < 0.05 and len(body):
        a = int(rng.integers(0, len(body)))
        body[a:a + 10] = bytes(body[a:a + 10]).lower()
    return bytes(body)


def test_reference_kats_on_the_second_reading():
    """test/adaptertrimmer_test.cpp: the four vectors the reference holds for this file"""
    ad = b"GCGCATACTTTTCCACGGGGATACTACTG"
    s = b"AGGTGCTGCGCATACTTTTCCACGGGGATACTACTGGGTGTTACCGTGGGAATGAATCCTTTTAACCTTAGCAATACGTAAAGGTGCT"
    assert sr.trim_start(s, ad, 0.3, 0)[0] == b"GGTGTTACCGTGGGAATGAATCCTTTTAACCTTAGCAATACGTAAAGGTGCT"
    s = b"TTTTAACCCCCCCCCCCCCCCCCCCCCCCCCCCCAATTTTAAAAGCGCATACTTTTCCACGGGGA"
    assert sr.trim_end(s, ad, 0.3, 0)[0] == b"TTTTAACCCCCCCCCCCCCCCCCCCCCCCCCCCCAATTTTAAAA"
    s = b"TTTTAACCCCCCCCCCCCCCCCCCCCCCCCCCCCAATTTTAAAATTTTCCCCGGGGAAATTTCCCGGGAAATTTCCCGGGATCGATCGATCGATCGAATTCC"
    assert sr.search_adapter(s, b"TTTT", 0.3, 0, -1, True, False) == 0
    assert sr.search_adapter(s, b"AACC", 0.3, 0, -1, True, False) == 4


def test_second_reading_edit_distance_matches_the_pinned_one(orc):
    """the textbook table against orc_edit_distance, which IS pinned against the real editdistance.o (tests/test_oracle_vs_ref.py)"""
    rng = np.random.default_rng(5)
    for _ in range(1500):
        a = _rnd(rng, rng.integers(0, 80), b"ACGTN")
        b = _mutated(rng, a, 0.2) if rng.random() < 0.6 else _rnd(rng, rng.integers(0, 80))
        assert sr.levenshtein(a, b) == orc.edit_distance(a, b), (a, b)


@pytest.mark.parametrize("seed", range(6))
def test_search_adapter_default_mode_and_find_middle_two_readings_agree(orc, seed):
    """searchAdapter's default mode (first strict minimum, last position never visited, one edit-distance confirmation) and
    findMiddleAdapters' combination of the two hits: 6 x 1 300 seeded (read, adapter pair, ed_max, extension) cases"""
    rng = np.random.default_rng(7000 + seed)
    found = both = 0
    for _ in range(1300):
        sa, ea = _adapter(rng), _adapter(rng)
        if rng.random() < 0.2:
            ea = bytes(synth.revcomp(sa.decode()).encode()) if sa else ea
        seq = _read_with(rng, sa if rng.random() < 0.5 else ea)
        if rng.random() < 0.3 and len(seq) > 40 and len(ea):
            at = int(rng.integers(0, len(seq)))
            seq = seq[:at] + _mutated(rng, ea, 0.1) + seq[at:]
        ed, ext = float(rng.choice(ED_MAX)), int(rng.choice([0, 0, 5, 10, 30, 500]))
        for ad in (sa, ea):
            assert orc.search_adapter(seq, ad, ed) == sr.search_adapter(seq, ad, ed), (seq, ad, ed)
        want = sr.find_middle(seq, sa, ea, ed, ext)
        got = orc.find_middle(seq, sa, ea, ed, ext)
        assert got[0] == want[0] and (not want[0] or got[1:] == want[1:]), (seq, sa, ea, ed, ext, got, want)
        found += want[0]
        both += sr.search_adapter(seq, sa, ed) >= 0 and sr.search_adapter(seq, ea, ed) >= 0
    assert found > 200 and both > 50


@pytest.mark.parametrize("seed", range(4))
def test_search_adapter_window_modes_two_readings_agree(orc, seed):
    """asLeftAsPossible / asRightAsPossible with search windows as the end trims pass them and as nobody passes them"""
    rng = np.random.default_rng(7100 + seed)
    hits = 0
    for _ in range(1500):
        ad = _adapter(rng)
        seq = _read_with(rng, ad, rlen_max=450)
        ed = float(rng.choice(ED_MAX))
        start = int(rng.choice([0, 0, max(0, len(seq) - 200), int(rng.integers(0, len(seq) + 5))]))
        length = int(rng.choice([-1, 0, 200, 200, int(rng.integers(1, 400))]))
        for left, right in ((True, False), (False, True), (True, True), (False, False)):
            want = sr.search_adapter(seq, ad, ed, start, length, left, right)
            assert orc.search_adapter(seq, ad, ed, start, length, left, right) == want, (seq, ad, ed, start, length, left, right)
            hits += want >= 0
    assert hits > 500


@pytest.mark.parametrize("seed", range(4))
def test_end_trims_two_readings_agree(orc, seed):
    """trimBySequenceStart / trimBySequenceEnd: what is left of the read, the value returned (it feeds adapter_trimmed_bases
    unclamped) and the length of the key handed to addAdapterTrimmed"""
    rng = np.random.default_rng(7200 + seed)
    full = partial = 0
    for _ in range(900):
        ad = _adapter(rng)
        seq = _read_with(rng, ad, rlen_max=500)
        ed, ext = float(rng.choice(ED_MAX)), int(rng.choice([0, 0, 5, 10, 30, 500]))
        for name in ("trim_start", "trim_end"):
            want = getattr(sr, name)(seq, ad, ed, ext)
            got = getattr(orc, name)(seq, ad, ed, ext)
            assert (got[0].encode("latin-1"), got[1], got[2]) == want, (name, seq, ad, ed, ext, got, want)
            full += want[2] == len(ad) and want[1] > 0
            partial += 0 < want[2] < len(ad)
    assert full > 100 and partial > 20


# ---- adapter auto-detection (src/evaluator.cpp:166-404): the checker's literal loops (oracle/evaluator_oracle.c) against the
# ---- whole-array second reading


def _eval_lib(orc):
    import ctypes as C

    L = orc.lib()
    L.orc_eval_top_key.restype = C.c_int
    L.orc_eval_top_key.argtypes = [C.c_void_p, C.c_int]
    L.orc_eval_extend_key.restype = C.c_int
    L.orc_eval_extend_key.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p]
    L.orc_eval_count_end_kmers.restype = None
    L.orc_eval_count_end_kmers.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    return L


def _planted(rng, n_reads, alphabet=b"ACGT", side=0):
    ad = _rnd(rng, rng.integers(14, 70), alphabet)
    reads = []
    for _ in range(n_reads):
        n = int(rng.choice([0, 5, 9, 10, 11, 12, 137, 138, 139, 140])) if rng.random() < 0.1 else int(rng.integers(20, 420))
        s = bytearray(_rnd(rng, n, alphabet))
        if rng.random() < 0.75 and n:
            c = _mutated(rng, ad, float(rng.choice([0.0, 0.0, 0.03, 0.1])))
            lead = int(rng.integers(0, 4))
            if side == 0:
                s[lead:lead + len(c)] = c[:max(0, n - lead)]
            else:
                c = c[:n]
                s[max(0, n - lead - len(c)):max(0, n - lead - len(c)) + len(c)] = c
            s = s[:n]
        if rng.random() < 0.1 and n:
            s[int(rng.integers(0, n))] = ord("N")
        if rng.random() < 0.03 and n:
            s[int(rng.integers(0, n))] = ord("a")
        reads.append(bytes(s))
    return reads


@pytest.mark.parametrize("seed", range(10))
def test_detection_counting_seed_and_growth_two_readings_agree(orc, seed):
    """the counting loops of evalAdapterAndReadNum (both sides, every tail shift), getTopKey and extendKeyToAdapter (DNA and RNA
    letters, left- and right-first) on reads with a planted adapter"""
    import ctypes as C

    L = _eval_lib(orc)
    rng = np.random.default_rng(7300 + seed)
    rna = seed % 3 == 0
    side = seed % 2
    reads = _planted(rng, int(rng.integers(150, 400)), b"ACGU" if rna else b"ACGT", side)
    shift = int(rng.choice([1, 1, 2, 7]))
    seq = np.frombuffer(b"".join(reads) + b"\0" * 16, np.uint8).copy()
    off = np.zeros(len(reads) + 1, np.uint64)
    off[1:] = np.cumsum([len(r) for r in reads])
    cnt = np.zeros(1 << 20, np.uint32)
    acc = np.zeros(1 << 20, np.uint64)
    tot = C.c_uint64(0)
    L.orc_eval_count_end_kmers(seq.ctypes.data, off.ctypes.data, len(reads), side, shift, cnt.ctypes.data, acc.ctypes.data, C.byref(tot))
    c2, a2, t2 = sr.count_end_kmers(reads, side, shift)
    assert tot.value == t2 > 0 and np.array_equal(cnt, c2) and np.array_equal(acc, a2)
    if seed >= 5:  # sparse noise over the table: barred keys with large counts, ties, chains that break
        k = rng.integers(0, 1 << 20, 4000)
        cnt[k] += rng.integers(1, 300, 4000).astype(np.uint32)
        acc[k] += cnt[k].astype(np.uint64) * rng.integers(0, 120, 4000).astype(np.uint64)
    cnt[0] = 0  # src/evaluator.cpp:191
    key = L.orc_eval_top_key(cnt.ctypes.data, 10)
    assert key == sr.top_key(cnt) and key >= 0
    for left_first in (1, 0):
        out = C.create_string_buffer(80)
        L.orc_eval_extend_key(key, cnt.ctypes.data, acc.ctypes.data, 10, int(rna), left_first, out)
        assert out.value.decode() == sr.extend_key(key, cnt, acc, rna, bool(left_first)), (seed, left_first)
    if seed < 5:
        assert len(sr.extend_key(key, cnt, acc, rna, True)) > 12  # the planted adapter grows out of its seed


def test_top_key_rules_two_readings_agree(orc):
    """every rule that bars a key, and the rule that reads the COUNT's digits (`val`, src/evaluator.cpp:293-299), on small tables:
    300 tables of 1..40 random keys with random counts"""
    L = _eval_lib(orc)
    rng = np.random.default_rng(7400)
    none = 0
    for _ in range(300):
        cnt = np.zeros(1 << 20, np.uint32)
        n = int(rng.integers(1, 41))
        keys = rng.integers(0, 1 << 20, n)
        if rng.random() < 0.5:  # keys the rules are about: runs, two-letter keys, repeats of five, GC-rich, GGGG...
            a, b = int(rng.integers(0, 4)), int(rng.integers(0, 4))
            half = int(rng.integers(0, 1 << 10))
            keys[:5] = [sum(a << (2 * i) for i in range(10)), sum((a if i % 2 else b) << (2 * i) for i in range(10)),
                        (half << 10) | half, (0xff << 12) | int(rng.integers(0, 1 << 12)),
                        sum(int(rng.choice([2, 3, 2, 3, 0])) << (2 * i) for i in range(10))][:min(5, n)]
        cnt[keys] = rng.choice([1, 3, 0b111001, 0x15555, 1000, 65535, 1 << 20, 0xFFFFFFFF], n) + rng.integers(0, 50, n).astype(np.uint32)
        got = L.orc_eval_top_key(cnt.ctypes.data, 10)
        assert got == sr.top_key(cnt), (keys, cnt[keys])
        none += got < 0
    assert 0 < none < 300


def test_key_coder_round_trip_reference_kat():
    """test/evaluator_test.cpp: int2seq(seq2int(s)) == s for the reference's own strings, on the second reading's coder"""
    for s in ("ATCGATCGAT", "GGGGGGGGGG", "AAAAAAAAAA", "TTTTCCCCGG"):
        c, _, t = sr.count_end_kmers([s.encode() + b"A"], 0, 1)
        key = int(np.nonzero(c)[0][0])
        assert t == 1 and sr.int2seq(key, 10) == s
