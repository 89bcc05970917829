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
