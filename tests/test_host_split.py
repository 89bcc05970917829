"""--split / --split_by_lines and --adapter_fasta on the host, pinned against the REAL reference objects
(oracle/_ref: ThreadConfig + Writer, FastaReader compiled in place from /root/reference/src).

The split writer of the CLI (fastplong_amd/host/split.cpp) replays, in one thread, what the reference's workers do
with their private writers; here both are driven with the same pack sequences -- packs dealt round-robin to the
workers, every pack handed over (the reference run in which the reader thread stays ahead of the workers) -- and must
leave the same files with the same bytes."""
import ctypes as C
import gzip
import os

import numpy as np
import pytest

from fastplong_amd import build


@pytest.fixture(scope="module")
def hostlib():
    build.build_host()
    L = C.CDLL(build.HOST_LIB)
    L.fplh_split_replay.restype = C.c_int
    L.fplh_split_replay.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_long, C.c_int, C.c_uint,
                                    C.POINTER(C.c_int), C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_char_p)]
    L.fplh_load_fasta.restype = C.c_int
    L.fplh_load_fasta.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_ulonglong)]
    L.fplh_free.argtypes = [C.c_void_p]
    return L


def _listing(d, gz):
    out = {}
    for f in sorted(os.listdir(d)):
        raw = open(os.path.join(d, f), "rb").read()
        # a .gz file nothing was written to: the reference leaves 0 bytes, this host an empty gzip member
        out[f] = (gzip.decompress(raw) if raw else b"") if gz else raw
    return out


def _scenarios():
    rng = np.random.default_rng(2024)
    sc = []
    for i in range(60):
        T = int(rng.integers(1, 6))
        by_lines = bool(rng.integers(0, 2))
        number = int(rng.integers(max(2, T), 12)) if not by_lines else 0
        size = int(rng.integers(1, 60))
        digits = int(rng.choice([0, 1, 4, 6]))
        n_packs = int(rng.integers(0, 80))
        reads = [16] * n_packs
        if n_packs and rng.random() < 0.7:
            reads[-1] = int(rng.integers(1, 17))  # the last pack of an input is short
        passed = [int(rng.integers(0, r + 1)) for r in reads]
        sc.append(dict(T=T, by_lines=by_lines, number=number, size=size, digits=digits, reads=reads, passed=passed,
                       gz=(i % 5 == 4), out=(i % 11 != 10)))
    # the corner the reference's mCanBeStopped is about: files used up, number % threads != 0
    sc.append(dict(T=3, by_lines=False, number=4, size=16, digits=4, reads=[16] * 30, passed=[16] * 30, gz=False, out=True))
    sc.append(dict(T=4, by_lines=False, number=6, size=5, digits=2, reads=[16] * 25, passed=[9] * 25, gz=False, out=True))
    return sc


@pytest.mark.parametrize("k", range(62))
def test_split_writer_equals_reference_threadconfig(ref, hostlib, tmp_path, k):
    s = _scenarios()[k]
    n = len(s["reads"])
    name = "out.fq.gz" if s["gz"] else "out.fq"
    d_ref, d_got = tmp_path / "ref", tmp_path / "got"
    d_ref.mkdir()
    d_got.mkdir()
    texts = [("@p%d_%d_%d;" % (k, i, s["passed"][i])) * (1 + s["passed"][i]) for i in range(n)]
    per = [s["passed"][i] if s["by_lines"] else s["reads"][i] for i in range(n)]
    lines = ["S_BEGIN %d %d %d %d %d 4 %s" % (s["T"], int(s["by_lines"]), s["number"], s["size"], s["digits"],
                                              ref.s(str(d_ref / name) if s["out"] else ""))]
    lines += ["S_PACK %d %d %s" % (i % s["T"], per[i], ref.s(texts[i])) for i in range(n)]
    lines += ["S_END"]
    out = ref.run(lines).split("\n")
    stopped = [int(x) for x in out[1:1 + n]]
    W = (C.c_int * max(n, 1))(*[i % s["T"] for i in range(n)])
    R = (C.c_long * max(n, 1))(*s["reads"])
    P = (C.c_long * max(n, 1))(*s["passed"])
    X = (C.c_char_p * max(n, 1))(*[t.encode() for t in texts])
    nfiles = hostlib.fplh_split_replay(str(d_got / name).encode() if s["out"] else b"", s["digits"], s["T"],
                                       int(s["by_lines"]), s["number"], s["size"], 4, n, W, R, P, X)
    want, got = _listing(d_ref, s["gz"]), _listing(d_got, s["gz"])
    assert list(got) == list(want), (s, sorted(got), sorted(want))
    assert got == want, s
    if s["out"]:
        assert nfiles == len(want)
        if not s["by_lines"]:
            assert len(want) == max(s["number"], s["T"])  # every file of --split exists, however short the input
    if k == 60:  # workers 1 and 2 may be stopped by the reference once their files are used up; worker 0 never
        assert any(stopped) and not stopped[0]


def test_fasta_reader_equals_reference(ref, hostlib, tmp_path):
    """--adapter_fasta through the real FastaReader and through the host's restatement: '>' inside header / sequence
    lines, lower case, CRLF, blank lines, characters str_keep_valid_sequence drops, text before the first '>', a last
    line without terminator, duplicate headers"""
    files = {
        "plain.fa": b">ad1 first\nACGTACGTAC\n>ad0\nTTTTGGGGCC\nAACC\n",
        "gt_inside.fa": b">ad1 desc with > inside\nACGT>ACGT\nGG>TT\n>ad2>x\nacgtnn\n",
        "lower_crlf.fa": b">a\r\nacgtACGT\r\nttgg\r\n>b\r\nGGCC**--\r\n",
        "blank_lines.fa": b">a\nACGT\n\nTTTT\n\n>b\n\nGGGG\n",
        "junk_first.fa": b"some text\nmore >a\nACGTAC\n>b\nTTGGAA",
        "odd_chars.fa": b">a\n1ACGT 23N-*x\n-ACGT\n>short\nACG\n>a\nGGGGGGGG\n",
        "empty.fa": b"",
        "no_records.fa": b"ACGT\nTTTT\n",
    }
    for name, data in files.items():
        p = tmp_path / name
        p.write_bytes(data)
        out = ref.run(["FA %s" % ref.s(str(p))]).split("\n")
        n = int(out[0])
        want = [tuple(bytes.fromhex(x[1:]) for x in line.split(" ")) for line in out[1:1 + n]]
        buf, ln = C.c_void_p(), C.c_ulonglong()
        got_n = hostlib.fplh_load_fasta(str(p).encode(), C.byref(buf), C.byref(ln))
        text = C.string_at(buf, ln.value)
        hostlib.fplh_free(buf)
        got = []
        rest = text
        # "header\tsequence\n" records; headers and sequences may themselves hold tabs / line feeds, so walk by the
        # reference's own values
        for h, sq in want:
            rec = h + b"\t" + sq + b"\n"
            assert rest.startswith(rec), (name, rest[:80], rec)
            rest = rest[len(rec):]
            got.append((h, sq))
        assert rest == b"" and got_n == n, (name, rest, got_n, n)
