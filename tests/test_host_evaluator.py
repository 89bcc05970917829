"""Adapter auto-detection on the host (fastplong_amd/host/evaluator.cpp, restated from the reference's
Evaluator::evalAdapterAndReadNum).  The reference object cannot be built here (its FASTQ reader needs ISA-L), so
this is pinned by the reference's own known-answer test plus behavioural checks on seeded reads."""
import ctypes as C

import numpy as np
import pytest

from fastplong_amd import build, synth
from tests import hostio


@pytest.fixture(scope="module")
def hostlib():
    build.build_host()
    L = C.CDLL(build.HOST_LIB)
    L.fplh_seq2int.restype = C.c_int
    L.fplh_seq2int.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.fplh_int2seq.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_char_p]
    L.fplh_detect_adapters.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p]
    return L


def test_int2seq_roundtrip_reference_kat(hostlib):
    """reference test/evaluator_test.cpp: int2seq(seq2int(s, 0, 10, -1), 10) == s"""
    s = b"ATCGATCGAT"
    out = C.create_string_buffer(64)
    key = hostlib.fplh_seq2int(s, len(s), 0, 10, -1)
    assert key == int("".join("%d%d" % divmod("ATCG".index(chr(c)), 2) for c in s), 2)
    hostlib.fplh_int2seq(key, 10, 0, out)
    assert out.value == s
    # rolling form == direct form; anything but A/T/U/C/G invalidates the key
    t = b"ACGTTGCANACGTACGTACGTT"
    last = -1
    for pos in range(len(t) - 10 + 1):
        direct = hostlib.fplh_seq2int(t, len(t), pos, 10, -1)
        last = hostlib.fplh_seq2int(t, len(t), pos, 10, last)
        assert last == direct and (direct < 0) == (b"N" in t[pos:pos + 10])
    hostlib.fplh_int2seq(key, 10, 1, out)
    assert out.value == s.replace(b"T", b"U")


def _detect(hostlib, tmp_path, reads):
    seq, qual, off = synth.pack(reads)
    text, _, _ = hostio.make_fastq(seq, qual, off)
    p = tmp_path / "in.fq"
    p.write_bytes(text)
    a, b = C.create_string_buffer(128), C.create_string_buffer(128)
    hostlib.fplh_detect_adapters(str(p).encode(), 0, 0, a, b)
    return a.value.decode(), b.value.decode()


def test_detects_the_adapters_most_reads_carry(hostlib, tmp_path):
    rng = np.random.default_rng(3)
    sa = np.frombuffer(synth.START_ADAPTER.encode(), np.uint8)
    ea = np.frombuffer(synth.END_ADAPTER.encode(), np.uint8)
    reads = []
    for _ in range(400):
        body = synth._ACGT[rng.integers(0, 4, int(rng.integers(400, 900)))]
        s = np.concatenate([sa, body, ea, synth._ACGT[rng.integers(0, 4, 1)]]) if rng.random() < 0.8 else body
        reads.append((s.astype(np.uint8), np.full(len(s), 33 + 20, np.uint8)))
    start, end = _detect(hostlib, tmp_path, reads)
    # the k-mer walk reconstructs the adapter up to where the counts thin out at its borders
    assert len(start) > 16 and (start in synth.START_ADAPTER or synth.START_ADAPTER in start)
    assert len(end) > 16 and (end in synth.END_ADAPTER or synth.END_ADAPTER in end)


def test_nothing_detected_in_random_reads_or_small_files(hostlib, tmp_path):
    rng = np.random.default_rng(4)
    reads = [(synth._ACGT[rng.integers(0, 4, 600)].astype(np.uint8), np.full(600, 33 + 20, np.uint8)) for _ in range(300)]
    assert _detect(hostlib, tmp_path, reads) == ("auto", "auto")
    assert _detect(hostlib, tmp_path, reads[:50]) == ("auto", "auto")  # fewer than 100 records: no evaluation


def _write_reads(path, n, length, seed=0, gz=False):
    import gzip
    rng = np.random.default_rng(seed)
    body = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (n, length))]
    rec = np.empty((n, 17 + 2 * length), np.uint8)
    names = np.array([list(b"@r%010d" % i) for i in range(n)], np.uint8)
    rec[:, :12] = names
    rec[:, 12] = 10
    rec[:, 13:13 + length] = body
    rec[:, 13 + length] = 10
    rec[:, 14 + length] = ord("+")
    rec[:, 15 + length] = 10
    rec[:, 16 + length:16 + 2 * length] = 33 + 20
    rec[:, 16 + 2 * length] = 10
    data = rec.tobytes()
    if gz:
        with gzip.GzipFile(path, "wb", compresslevel=1, mtime=0) as f:
            f.write(data)
    else:
        open(path, "wb").write(data)
    return len(data), 17 + 2 * length


def test_read_number_estimate(hostlib, tmp_path):
    """Evaluator::evaluateReadNum / the estimate of evalAdapterAndReadNum (src/evaluator.cpp:62-103,141-150): exact when
    the evaluated prefix reaches the end of the file, else size * 1.01 / (bytes per read), with `bytes` counted in the
    8 MiB buffers the reference's reader had pulled (FastqReader::getBytes)."""
    L = hostlib
    L.fplh_evaluate_read_num.restype = C.c_long
    L.fplh_evaluate_read_num.argtypes = [C.c_char_p]
    L.fplh_detect_read_num.restype = C.c_long
    L.fplh_detect_read_num.argtypes = [C.c_char_p]
    p = str(tmp_path / "small.fq")
    _write_reads(p, 3000, 150)
    assert L.fplh_evaluate_read_num(p.encode()) == 3000 and L.fplh_detect_read_num(p.encode()) == 3000
    # 80 000 reads of 200 bases: the adapter evaluation stops after 65 536 reads = 27.3 MB, inside the 4th buffer
    p = str(tmp_path / "mid.fq")
    size, rec = _write_reads(p, 80000, 200)
    buf = 1 << 23
    pulled = lambda u: min(size, ((u - 1) // buf + 1) * buf)  # noqa: E731
    want = int(size * 1.01 / ((pulled(65536 * rec) - pulled(rec)) / 65536))
    assert L.fplh_detect_read_num(p.encode()) == want and abs(want - 80000) < 20000
    assert L.fplh_evaluate_read_num(p.encode()) == 80000  # 512 Ki reads / 77 Mbases are not reached: exact
    # everything evaluated sits in the first buffer: bytes per read = 0, the reference's (long)(+inf)
    p = str(tmp_path / "tiny_reads.fq")
    _write_reads(p, 70000, 30)
    assert L.fplh_detect_read_num(p.encode()) == -(1 << 63)
    # gzip: compressed bytes behind each 8 MiB inflate call
    p = str(tmp_path / "mid.fq.gz")
    _write_reads(p, 80000, 200, gz=True)
    got = L.fplh_detect_read_num(p.encode())
    assert abs(got - 80000) < 25000, got


def _kmer_case(seed):
    """reads of 0..600 bases (incl. shorter than a key), with N / lower-case bytes and U, planted adapters at both ends"""
    import numpy as np
    from fastplong_amd import synth
    rng = np.random.default_rng(seed)
    reads = []
    ad = np.frombuffer(b"AAGGATTCATTCCCACGGTAACAC", np.uint8)
    for i in range(300):
        n = int(rng.choice([0, 5, 9, 10, 11, 12, 137, 138, 139, 140])) if rng.random() < 0.3 else int(rng.integers(0, 600))
        s = synth._ACGT[rng.integers(0, 4, n)].astype(np.uint8)
        if n > 60 and rng.random() < 0.5:
            s[:24] = ad
        if n > 60 and rng.random() < 0.5:
            s[-24:] = ad[::-1]
        for _ in range(int(rng.integers(0, 3))):
            if n:
                s[int(rng.integers(0, n))] = int(rng.choice([ord("N"), ord("a"), ord("U"), ord("t")]))
        reads.append((s, np.full(n, 40, np.uint8)))
    return synth.pack(reads)


def _host_counts(L, seq, off, side, shift):
    import ctypes as C
    import numpy as np
    cnt = np.zeros(1 << 20, np.uint32)
    pos = np.zeros(1 << 20, np.uint64)
    tot = C.c_uint64(0)
    L.fplh_count_end_kmers_host.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    L.fplh_count_end_kmers_host(seq.ctypes.data, off.ctypes.data, len(off) - 1, side, shift, cnt.ctypes.data, pos.ctypes.data, C.byref(tot))
    return cnt, pos, tot.value


def _oracle_counts(seq, off, side, shift):
    """the checker's literal restatement of the reference's two counting loops (oracle/evaluator_oracle.c,
    src/evaluator.cpp:166-183, :207-225 with Evaluator::seq2int rolling the key)"""
    import ctypes as C
    import numpy as np
    from oracle import oracle
    L = oracle.lib()
    L.orc_eval_count_end_kmers.restype = None
    L.orc_eval_count_end_kmers.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    cnt = np.zeros(1 << 20, np.uint32)
    pos = np.zeros(1 << 20, np.uint64)
    tot = C.c_uint64(0)
    seq = np.ascontiguousarray(np.concatenate([seq, np.zeros(16, np.uint8)]))
    off = np.ascontiguousarray(off.astype(np.uint64))
    L.orc_eval_count_end_kmers(seq.ctypes.data, off.ctypes.data, len(off) - 1, side, shift, cnt.ctypes.data, pos.ctypes.data, C.byref(tot))
    return cnt, pos, tot.value


def test_oracle_key_coder_reference_kat(orc):
    """reference test/evaluator_test.cpp on the checker's own seq2int: the key of ATCGATCGAT spells it back (int2seq is the host's,
    pinned by the same KAT above), rolling == direct, N invalidates"""
    import ctypes as C
    L = orc.lib()
    L.orc_eval_seq2int.restype = C.c_int
    L.orc_eval_seq2int.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
    s = b"ATCGATCGAT"
    key = L.orc_eval_seq2int(s, 0, 10, -1)
    assert key == int("".join("%d%d" % divmod("ATCG".index(chr(c)), 2) for c in s), 2)
    t = b"ACGTTGCANACGUACGTACGTT"
    last = -1
    for pos in range(len(t) - 10 + 1):
        direct = L.orc_eval_seq2int(t, pos, 10, -1)
        last = L.orc_eval_seq2int(t, pos, 10, last)
        assert last == direct and (direct < 0) == (b"N" in t[pos:pos + 10])


@pytest.mark.parametrize("side,shift", [(0, 1), (1, 1), (0, 7), (1, 30)])
def test_host_kmer_counting_equals_oracle(orc, side, shift):
    """the host form (csrc/adapter_pick.h's window + key coder, one key per position) against the literal rolling loops"""
    import ctypes as C
    from fastplong_amd import build
    build.build_host()
    H = C.CDLL(build.HOST_LIB)
    for seed in (100, 101, 102):
        seq, _, off = _kmer_case(seed)
        want = _oracle_counts(seq, off, side, shift)
        got = _host_counts(H, np.ascontiguousarray(np.concatenate([seq, np.zeros(16, np.uint8)])), np.ascontiguousarray(off.astype(np.uint64)), side, shift)
        assert got[2] == want[2] > 0
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


@pytest.mark.parametrize("side,shift", [(0, 1), (1, 1), (0, 7), (1, 30)])
def test_emulated_kmer_counting_kernel_equals_host(side, shift):
    """k_count_end_kmers (what fpl_count_end_kmers runs on the GPU for the adapter auto-detection) on the CPU emulator against
    the host's counting loops (Evaluator::evalAdapterAndReadNum, src/evaluator.cpp:300-345): counters, position sums, total"""
    import ctypes as C
    import numpy as np
    from fastplong_amd import build
    from tests.emu import emu
    build.build_host()
    H = C.CDLL(build.HOST_LIB)
    seq, _, off = _kmer_case(100 + side)
    seq = np.ascontiguousarray(np.concatenate([seq, np.zeros(16, np.uint8)]))
    off = np.ascontiguousarray(off.astype(np.uint64))
    want = _host_counts(H, seq, off, side, shift)
    E = emu.lib()
    cnt = np.zeros(1 << 20, np.uint32)
    pos = np.zeros(1 << 20, np.uint64)
    tot = C.c_uint64(0)
    E.emu_count_end_kmers.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    E.emu_count_end_kmers(seq.ctypes.data, off.ctypes.data, len(off) - 1, side, shift, cnt.ctypes.data, pos.ctypes.data, C.byref(tot))
    assert tot.value == want[2] > 0
    assert np.array_equal(cnt, want[0]) and np.array_equal(pos, want[1])
    orc_want = _oracle_counts(seq[:-16], off, side, shift)
    assert tot.value == orc_want[2] and np.array_equal(cnt, orc_want[0]) and np.array_equal(pos, orc_want[1])


@pytest.mark.gpu
@pytest.mark.parametrize("side,shift", [(0, 1), (1, 1), (0, 7), (1, 30)])
def test_device_kmer_counting_equals_oracle(orc, side, shift):
    """fpl_count_end_kmers through the C-ABI on the GPU against the CHECKER's counting loops (oracle/evaluator_oracle.c: the
    literal restatement of src/evaluator.cpp:166-183, :207-225), not against the product's own host form"""
    import ctypes as C
    import numpy as np
    from fastplong_amd import engine
    L = engine.load_library()
    seq, _, off = _kmer_case(200 + side)
    off = np.ascontiguousarray(off.astype(np.uint64))
    seq = np.ascontiguousarray(seq)
    want = _oracle_counts(seq, off, side, shift)
    cnt = np.zeros(1 << 20, np.uint32)
    pos = np.zeros(1 << 20, np.uint64)
    tot = C.c_uint64(0)
    rc = L.fpl_count_end_kmers(0, seq.ctypes.data, off.ctypes.data, len(off) - 1, side, shift, cnt.ctypes.data, pos.ctypes.data, C.byref(tot))
    assert rc == 0 and tot.value == want[2] > 0
    assert np.array_equal(cnt, want[0]) and np.array_equal(pos, want[1])


# ---- seed choice and growth (csrc/adapter_pick.h) against the literal restatement of Evaluator::getTopKey /
# ---- extendKeyToAdapter kept with the checker (oracle/evaluator_oracle.c); parity with the reference itself: unpinned


def _pick_tables(seed):
    """counter tables as the detection sees them: reads with one adapter planted at the start of most of them (noisy copies),
    plus -- for odd seeds -- random sparse counts sprinkled over the table so that inadmissible keys with large counts, ties and
    broken chains occur"""
    import numpy as np
    from fastplong_amd import build, synth
    import ctypes as C
    rng = np.random.default_rng(seed)
    ad = synth._ACGT[rng.integers(0, 4, int(rng.integers(18, 70)))].astype(np.uint8)
    reads = []
    for i in range(int(rng.integers(150, 500))):
        n = int(rng.integers(150, 500))
        s_ = synth._ACGT[rng.integers(0, 4, n)].astype(np.uint8)
        if rng.random() < 0.8:
            a = ad.copy()
            for _ in range(int(rng.integers(0, 3))):
                a[int(rng.integers(0, len(a)))] = synth._ACGT[int(rng.integers(0, 4))]
            lead = int(rng.integers(0, 3))
            s_[lead:lead + len(a)] = a[:n - lead]
        reads.append((s_, np.full(n, 40, np.uint8)))
    seq, _, off = synth.pack(reads)
    build.build_host()
    H = C.CDLL(build.HOST_LIB)
    cnt, pos, tot = _host_counts(H, np.concatenate([seq, np.zeros(16, np.uint8)]), np.ascontiguousarray(off.astype(np.uint64)), 0, 1)
    if seed % 2:
        k = rng.integers(0, 1 << 20, 3000)
        cnt[k] += rng.integers(1, 400, 3000).astype(np.uint32)
        pos[k] += (cnt[k].astype(np.uint64) * rng.integers(0, 120, 3000).astype(np.uint64))
        cnt[0] += 5000  # poly-A: counted as seen, never a seed
    return cnt, pos, H


def _oracle_pick(cnt, pos, is_rna):
    import ctypes as C
    from oracle import oracle
    L = oracle.lib()
    L.orc_eval_top_key.restype = C.c_int
    L.orc_eval_top_key.argtypes = [C.c_void_p, C.c_int]
    L.orc_eval_extend_key.restype = C.c_int
    L.orc_eval_extend_key.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p]
    c0 = cnt.copy()
    total_key = int((c0 > 0).sum())
    c0[0] = 0  # src/evaluator.cpp:191
    key = L.orc_eval_top_key(c0.ctypes.data, 10)
    if key < 0:
        return key, 0, total_key, b""
    out = C.create_string_buffer(80)
    L.orc_eval_extend_key(key, c0.ctypes.data, pos.ctypes.data, 10, int(is_rna), 1, out)
    return key, int(c0[key]), total_key, out.value


def _run_pick(fn, cnt, pos, is_rna):
    import ctypes as C
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_char_p]
    c, tk = C.c_uint32(0), C.c_uint32(0)
    out = C.create_string_buffer(80)
    key = fn(cnt.ctypes.data, pos.ctypes.data, int(is_rna), C.byref(c), C.byref(tk), out)
    return key, int(c.value), int(tk.value), out.value


@pytest.mark.parametrize("seed", range(12))
def test_host_seed_and_growth_equal_the_literal_restatement(orc, seed):
    cnt, pos, H = _pick_tables(seed)
    want = _oracle_pick(cnt, pos, seed % 3 == 0)
    got = _run_pick(H.fplh_pick_adapter, cnt, pos, seed % 3 == 0)
    assert got == want, (got, want)
    assert want[0] >= 0
    if seed % 2 == 0:
        assert len(want[3]) > 10  # the planted adapter is found and grown (the noise of the odd seeds may stop the walk)


def test_pick_rules_on_single_keys(orc):
    """every rule that bars a key from being the seed, one table each: the rule's key holds the largest count, a plain key the
    second largest"""
    import numpy as np
    import ctypes as C
    from fastplong_amd import build
    build.build_host()
    H = C.CDLL(build.HOST_LIB)
    code = {"A": 0, "T": 1, "C": 2, "G": 3}
    enc = lambda s_: sum(code[c] << (2 * (9 - i)) for i, c in enumerate(s_))  # noqa: E731
    plain = enc("ACGTTGCAAC")
    for barred in ["AAAAAAAAAA", "AAAAAACGTC", "ACACACACAC", "ACGTCACGTC", "GGCCGGCCAT", "GGGGACTAGC", "ATATATATCG"]:
        for val in (1000, 0x15555, 3, 0b111001):  # the count's own digits decide too (count_digits_vary)
            cnt = np.zeros(1 << 20, np.uint32)
            pos = np.zeros(1 << 20, np.uint64)
            cnt[enc(barred)] = val + 7
            cnt[plain] = val
            want = _oracle_pick(cnt, pos, False)
            got = _run_pick(H.fplh_pick_adapter, cnt, pos, False)
            assert got == want, (barred, val, got, want)


@pytest.mark.parametrize("seed", [1, 4, 7])
def test_emulated_pick_kernel_equals_host(seed):
    """k_pick_adapter (what fpl_pick_adapter runs behind the counting) on the CPU emulator against the host form"""
    from tests.emu import emu
    cnt, pos, H = _pick_tables(seed)
    want = _run_pick(H.fplh_pick_adapter, cnt, pos, seed == 4)
    got = _run_pick(emu.lib().emu_pick_adapter, cnt, pos, seed == 4)
    assert got == want and want[0] >= 0


def _planted_reads(seed, side, rna, n_reads=400, noisy=False):
    import numpy as np
    from fastplong_amd import synth
    rng = np.random.default_rng(seed)
    alen = int(rng.integers(20, 60)) if noisy else 34
    ad = synth._ACGT[rng.integers(0, 4, alen)].astype(np.uint8)
    if rna:
        ad[ad == ord("T")] = ord("U")
    reads = []
    for i in range(n_reads):
        n = int(rng.integers(200, 700))
        s_ = synth._ACGT[rng.integers(0, 4, n)].astype(np.uint8)
        if rna:
            s_[s_ == ord("T")] = ord("U")
        if rng.random() < 0.85:
            a = ad.copy()
            if noisy:
                for _ in range(int(rng.integers(0, 3))):
                    a[int(rng.integers(0, alen))] = ord("ACGN"[int(rng.integers(0, 4))])
            lead = int(rng.integers(0, 3)) if noisy else 0
            if side == 0:
                s_[lead:lead + alen] = a
            else:
                s_[n - alen - 1 - lead:n - 1 - lead] = a
        reads.append((s_, np.full(n, 40, np.uint8)))
    return synth.pack(reads)


@pytest.mark.gpu
@pytest.mark.parametrize("side,rna,seed,noisy", [(0, 0, 31, False), (1, 0, 32, False), (1, 1, 32, False)] +
                         [(s_ % 2, int(s_ % 3 == 0), 500 + s_, True) for s_ in range(12)])
def test_device_pick_adapter_equals_oracle(orc, side, rna, seed, noisy):
    """fpl_pick_adapter through the C-ABI on the GPU (counting + seed + growth in HBM) against the CHECKER on the same reads: the
    oracle's counting loops, then its literal getTopKey / extendKeyToAdapter (src/evaluator.cpp:268-404) on those tables"""
    import ctypes as C
    import numpy as np
    from fastplong_amd import abi, engine
    L = engine.load_library()
    seq, _, off = _planted_reads(seed, side, rna, noisy=noisy)
    off = np.ascontiguousarray(off.astype(np.uint64))
    cnt, pos, tot = _oracle_counts(seq, off, side, 1)
    want = _oracle_pick(cnt, pos, rna)
    p = abi.FplAdapterPick()
    rc = L.fpl_pick_adapter(0, np.ascontiguousarray(seq).ctypes.data, off.ctypes.data, len(off) - 1, side, 1, rna, C.byref(p))
    assert rc == 0
    assert (p.key, p.count, p.total_key, p.seq) == want and p.total == tot and p.len == len(want[3])
    if not noisy:
        assert p.len > 20
