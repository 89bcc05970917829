"""Host side of the path on the CPU: FASTQ -> CSR batches, result records -> output FASTQ text
(fastplong_amd/host/fastq.cpp), and the record formats pinned against the real reference
Read::breakByGap / appendToString / appendToStringWithTag (oracle/_ref)."""
import ctypes as C
import gzip

import numpy as np
import pytest

from fastplong_amd import abi, build, synth
from tests import hostio


@pytest.fixture(scope="module")
def hostlib():
    build.build_host()
    L = C.CDLL(build.HOST_LIB)
    L.fplh_batch_read.restype = C.c_void_p
    L.fplh_batch_read.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32]
    L.fplh_batch_n.restype = C.c_uint32
    L.fplh_batch_n.argtypes = [C.c_void_p]
    L.fplh_batch_bytes.restype = C.c_uint64
    L.fplh_batch_bytes.argtypes = [C.c_void_p]
    for f in (L.fplh_batch_seq, L.fplh_batch_qual, L.fplh_batch_off):
        f.restype = C.c_void_p
        f.argtypes = [C.c_void_p]
    L.fplh_batch_free.argtypes = [C.c_void_p]
    L.fplh_format_batch.restype = C.c_int
    L.fplh_format_batch.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                    C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.fplh_format_batch_parallel.restype = C.c_int
    L.fplh_format_batch_parallel.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                             C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.fplh_free.argtypes = [C.c_void_p]
    return L


def read_batch(L, path):
    b = L.fplh_batch_read(str(path).encode(), 2 ** 62, 2 ** 30)
    assert b
    n, nb = L.fplh_batch_n(b), L.fplh_batch_bytes(b)
    seq = np.ctypeslib.as_array(C.cast(L.fplh_batch_seq(b), C.POINTER(C.c_uint8)), (max(nb, 1),))[:nb].copy()
    qual = np.ctypeslib.as_array(C.cast(L.fplh_batch_qual(b), C.POINTER(C.c_uint8)), (max(nb, 1),))[:nb].copy()
    off = np.ctypeslib.as_array(C.cast(L.fplh_batch_off(b), C.POINTER(C.c_uint64)), (n + 1,)).copy()
    return b, seq, qual, off


@pytest.mark.parametrize("variant", ["plain", "crlf", "gz", "noeol", "junk"])
def test_fastq_to_csr(hostlib, tmp_path, variant):
    seq, qual, off = synth.adversarial(200, seed=3)
    # the reader cannot represent a record whose sequence line is empty differently from a blank
    # line between records; the reference has the same property -- keep empty reads out of this file
    keep = np.nonzero(np.diff(off.astype(np.int64)) > 0)[0]
    reads = [(seq[int(off[i]):int(off[i + 1])], qual[int(off[i]):int(off[i + 1])]) for i in keep]
    seq, qual, off = synth.pack(reads)
    text, names, strands = hostio.make_fastq(seq, qual, off, crlf=(variant == "crlf"), strand_names=True)
    if variant == "noeol":
        text = text[:-1]
    if variant == "junk":  # blank lines and stray non-@ lines between records are skipped
        text = b"\n\ngarbage line\n" + text
    p = tmp_path / ("in.fq.gz" if variant == "gz" else "in.fq")
    if variant == "gz":
        with gzip.open(p, "wb") as f:
            f.write(text)
    else:
        p.write_bytes(text)
    b, s2, q2, o2 = read_batch(hostlib, p)
    hostlib.fplh_batch_free(b)
    assert np.array_equal(o2, off) and np.array_equal(s2, seq) and np.array_equal(q2, qual)


@pytest.mark.parametrize("window", [16, 100, 4096])
@pytest.mark.parametrize("eol", [b"\n", b"\r\n", b"\r"])
def test_fastq_reader_refills_anywhere(hostlib, tmp_path, monkeypatch, window, eol):
    """the reader scans one window of the input in place: records that straddle, or exceed, the window and
    terminators split across two refills ("\\r" | "\\n") must parse exactly like the whole file; a lone
    "\\r" ends a line as in the reference's getLine"""
    monkeypatch.setenv("FPLH_READ_WINDOW", str(window))
    seq, qual, off = synth.adversarial(60, seed=12)
    keep = np.nonzero(np.diff(off.astype(np.int64)) > 0)[0]
    seq, qual, off = synth.pack([(seq[int(off[i]):int(off[i + 1])], qual[int(off[i]):int(off[i + 1])]) for i in keep])
    text, names, strands = hostio.make_fastq(seq, qual, off, strand_names=True)
    text = b"\n" + text.replace(b"\n", eol)
    if eol != b"\r":
        text = text[:-len(eol)]  # no terminator after the last line
    p = tmp_path / "in.fq"
    p.write_bytes(text)
    b, s2, q2, o2 = read_batch(hostlib, p)
    hostlib.fplh_batch_free(b)
    assert np.array_equal(o2, off) and np.array_equal(s2, seq) and np.array_equal(q2, qual)


def _read_all(L, path, mb, mr):
    L.fplh_batch_read_all.restype = C.c_void_p
    L.fplh_batch_read_all.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32]
    b = L.fplh_batch_read_all(str(path).encode(), mb, mr)
    assert b
    n, nb = L.fplh_batch_n(b), L.fplh_batch_bytes(b)
    seq = np.ctypeslib.as_array(C.cast(L.fplh_batch_seq(b), C.POINTER(C.c_uint8)), (max(nb, 1),))[:nb].copy()
    qual = np.ctypeslib.as_array(C.cast(L.fplh_batch_qual(b), C.POINTER(C.c_uint8)), (max(nb, 1),))[:nb].copy()
    off = np.ctypeslib.as_array(C.cast(L.fplh_batch_off(b), C.POINTER(C.c_uint64)), (n + 1,)).copy()
    L.fplh_batch_free(b)
    return seq, qual, off


@pytest.mark.parametrize("variant", ["clean", "crlf", "junk", "malformed", "at_quals"])
@pytest.mark.parametrize("caps", [(10 ** 9, 2 ** 30), (30000, 2 ** 30), (5000, 2 ** 30)])
def test_parallel_scan_equals_sequential(hostlib, tmp_path, monkeypatch, variant, caps):
    """mapped files are scanned by several threads, each starting at a header it had to guess; the pieces are
    only joined where they line up with the one-thread scan, so the result can never differ from it"""
    rng = np.random.default_rng(21)
    reads = []
    for _ in range(300):
        n = int(rng.integers(1, 700))
        q = rng.integers(33, 75, n).astype(np.uint8)
        if variant == "at_quals" or rng.random() < 0.3:
            q[0] = ord("@")  # a quality line that starts like a header
            if n > 1 and rng.random() < 0.5:
                q[1] = ord("+")
        reads.append((synth._ACGT[rng.integers(0, 4, n)].astype(np.uint8), q))
    seq, qual, off = synth.pack(reads)
    text, _, _ = hostio.make_fastq(seq, qual, off, crlf=(variant == "crlf"), strand_names=True)
    if variant == "junk":
        lines = text.split(b"\n")
        for i in range(40, len(lines) - 8, 41 * 4):  # stray lines between records
            lines[i:i] = [b"stray line", b"", b"+not a record"]
        text = b"\n".join(lines)
    if variant == "malformed":
        lines = text.split(b"\n")
        lines[4 * 150 + 2] = b"-"  # the '+' line of record 150
        text = b"\n".join(lines)
    p = tmp_path / "in.fq"
    p.write_bytes(text)
    monkeypatch.setenv("FPLH_PARSE_THREADS", "1")
    want = _read_all(hostlib, p, *caps)
    monkeypatch.setenv("FPLH_PARSE_MIN", "512")
    for threads in ("2", "7"):
        monkeypatch.setenv("FPLH_PARSE_THREADS", threads)
        hostlib.fplh_parallel_records.restype = C.c_uint64
        hostlib.fplh_parallel_records()
        got = _read_all(hostlib, p, *caps)
        for g, w in zip(got, want):
            assert np.array_equal(g, w)
        if variant in ("clean", "crlf"):  # nearly every record came from the threads, not from the sequential tail
            assert hostlib.fplh_parallel_records() > 0.8 * (len(want[2]) - 1)
    if variant == "malformed":
        assert len(want[2]) - 1 == 150
    elif variant != "junk":
        assert np.array_equal(want[2], off)


def test_format_batch_matches_python_composition(orc, hostlib, tmp_path):
    cfg = orc.Config(abi.FplOptions.default(cut_front=1, cut_tail=1, polyx=1, complexity_filter=1),
                     synth.START_ADAPTER, synth.END_ADAPTER)
    seq, qual, off = synth.adversarial(400, seed=8)
    keep = np.nonzero(np.diff(off.astype(np.int64)) > 0)[0]
    seq, qual, off = synth.pack([(seq[int(off[i]):int(off[i + 1])], qual[int(off[i]):int(off[i + 1])]) for i in keep])
    text, names, strands = hostio.make_fastq(seq, qual, off, strand_names=True)
    p = tmp_path / "in.fq"
    p.write_bytes(text)
    res, _ = orc.process_batch(cfg, seq, qual, off)
    assert (res["n_frag"] == 2).any() and (res["dropped"] == 1).any()
    b, *_ = read_batch(hostlib, p)
    out, failed = C.c_void_p(), C.c_void_p()
    ol, fl = C.c_uint64(), C.c_uint64()
    assert hostlib.fplh_format_batch(b, res.ctypes.data, C.byref(out), C.byref(ol), C.byref(failed), C.byref(fl)) == 0
    got_out, got_failed = C.string_at(out, ol.value), C.string_at(failed, fl.value)
    hostlib.fplh_free(out)
    hostlib.fplh_free(failed)
    for threads in (1, 3, 16):  # the sliced formatter the CLI uses writes the same bytes
        assert hostlib.fplh_format_batch_parallel(b, res.ctypes.data, threads, C.byref(out), C.byref(ol), C.byref(failed),
                                                  C.byref(fl)) == 0
        assert C.string_at(out, ol.value) == got_out and C.string_at(failed, fl.value) == got_failed
        hostlib.fplh_free(out)
        hostlib.fplh_free(failed)
    hostlib.fplh_batch_free(b)
    want_out, want_failed = hostio.expected_outputs(seq, qual, off, names, strands, res)
    assert got_out == want_out
    assert got_failed == want_failed
    assert len(want_failed) > 0 and b"split-by-adapter-right-" in want_out


def test_format_batch_with_fragment_list(orc, hostlib, tmp_path):
    """--break / --mask: the formatter works from the fragment / region lists (names through both insert(1, ..),
    N over the masked stretches, --failed_out only for single-fragment reads, masked only when it is r1 itself)"""
    hostlib.fplh_format_batch_fragments.restype = C.c_int
    hostlib.fplh_format_batch_fragments.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                                    C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                                    C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    opt = abi.FplOptions.default(cut_front=1, cut_tail=1, polyx=1, break_enabled=1, break_window=30, break_quality=12,
                                 mask_enabled=1, mask_window=15, mask_quality=13, n_base_percent_limit=95,
                                 unqualified_percent_limit=90)
    cfg = orc.Config(opt, synth.START_ADAPTER, synth.END_ADAPTER)
    rng = np.random.default_rng(5)
    seq, qual, off = synth.ont_like(80, seed=9, median_len=700, p_middle=0.3, p_polya=0.1)
    qual = qual.copy()
    for i in range(len(off) - 1):
        lo, hi = int(off[i]), int(off[i + 1])
        if rng.random() < 0.7:
            a = lo + int(rng.integers(0, hi - lo))
            qual[a:min(hi, a + int(rng.integers(20, 200)))] = 36
    text, names, strands = hostio.make_fastq(seq, qual, off, strand_names=True)
    p = tmp_path / "in.fq"
    p.write_bytes(text)
    res, _, frags, regs = orc.process_batch_ex(cfg, seq, qual, off)
    b, *_ = read_batch(hostlib, p)
    want_out, want_failed = hostio.expected_outputs_fragments(seq, qual, off, names, strands, res, frags, regs)
    for threads in (1, 5):
        out, failed = C.c_void_p(), C.c_void_p()
        ol, fl = C.c_uint64(), C.c_uint64()
        assert hostlib.fplh_format_batch_fragments(b, res.ctypes.data, frags.ctypes.data, len(frags), regs.ctypes.data, len(regs),
                                                   threads, C.byref(out), C.byref(ol), C.byref(failed), C.byref(fl)) == 0
        assert C.string_at(out, ol.value) == want_out and C.string_at(failed, fl.value) == want_failed
        hostlib.fplh_free(out)
        hostlib.fplh_free(failed)
    hostlib.fplh_batch_free(b)
    assert b"@r" in want_out and b"N" * 15 in want_out and len(want_failed) > 0


def test_record_formats_match_reference(ref):
    """names of split fragments and the failed-read tag line, from the reference's own Read code"""
    out = ref.run(["BG 3 2 =@r1_desc =ACGTACGTAC =+xyz =IIIIIJJJJJ"])
    head, body = out.split("\n", 1)
    assert head.split()[0] == "2"
    assert body == "@split-by-adapter-left-r1_desc\nACG\n+xyz\nIII\n@split-by-adapter-right-r1_desc\nCGTAC\n+xyz\nJJJJJ\n"
    for code, tag in ((12, "failed_too_many_n_bases"), (16, "failed_too_short"), (17, "failed_too_long"),
                      (20, "failed_quality_filter"), (24, "failed_low_complexity")):
        assert abi.FAILED_TYPES[code] == tag
        out = ref.run(["TAG %d =@r1 =ACGT =+ =IIII" % code])
        assert out.split("\n", 1)[1] == "@r1 %s\nACGT\n+\nIIII\n" % tag


def _gz_member(data, level=6):
    import gzip
    import io
    b = io.BytesIO()
    with gzip.GzipFile(fileobj=b, mode="wb", compresslevel=level, mtime=0) as f:
        f.write(data)
    return b.getvalue()


@pytest.mark.parametrize("variant", ["members", "false_header", "big_member", "trailing_zeros", "single"])
def test_multi_member_gzip_input(hostlib, tmp_path, monkeypatch, variant):
    """a gzip file that is a concatenation of members is inflated member by member on the worker pool; headers that only
    LOOK like member starts (inside stored blocks), a member too large to buffer, empty members and trailing padding
    all end up with exactly what the single zlib stream delivers"""
    rng = np.random.default_rng(5)
    reads = [(synth._ACGT[rng.integers(0, 4, n)].astype(np.uint8), rng.integers(35, 70, n).astype(np.uint8))
             for n in rng.integers(50, 3000, 400)]
    seq, qual, off = synth.pack(reads)
    text, _, _ = hostio.make_fastq(seq, qual, off)
    lines = text.split(b"\n")
    if variant == "false_header":  # gzip magic inside a read name, kept verbatim by a stored (level 0) member
        lines[4 * 37] = b"@read37 \x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03 looks like a member"
        lines[4 * 201] = b"@read201 \x1f\x8b\x08\x08 again"
        text = b"\n".join(lines)
    cuts = sorted(set(int(x) for x in rng.integers(0, len(text), 9)) | {0, len(text)})
    parts = [text[a:b] for a, b in zip(cuts[:-1], cuts[1:])]  # members cut anywhere, also inside records
    if variant == "single":
        blob = _gz_member(text)
    else:
        level = 0 if variant == "false_header" else 6
        blob = b"".join(_gz_member(p_, level) for p_ in parts[:4]) + _gz_member(b"") + b"".join(_gz_member(p_, level) for p_ in parts[4:])
    if variant == "trailing_zeros":
        blob += b"\0" * 1000
    if variant == "big_member":
        monkeypatch.setenv("FPLH_GZ_MEMBER_CAP", str(64 << 10))  # most members exceed 64 KiB: stream from the first such
    p = tmp_path / "in.fq.gz"
    p.write_bytes(blob)
    hostlib.fplh_gz_members.restype = C.c_uint64
    monkeypatch.setenv("FPLH_NO_GZ_MEMBERS", "1")
    want = _read_all(hostlib, p, 10 ** 9, 2 ** 30)
    assert len(want[2]) - 1 == 400 and np.array_equal(want[0], seq)
    monkeypatch.delenv("FPLH_NO_GZ_MEMBERS")
    hostlib.fplh_gz_members()
    for caps in ((10 ** 9, 2 ** 30), (20000, 2 ** 30)):
        got = _read_all(hostlib, p, *caps)
        for g, w in zip(got, want):
            assert np.array_equal(g, w)
    n = hostlib.fplh_gz_members()
    if variant in ("members", "false_header", "trailing_zeros"):
        assert n >= 2 * 9
    elif variant == "single":
        assert n == 0


@pytest.mark.parametrize("variant", ["members", "false_header", "big_member", "trailing_zeros", "single", "single_padded", "truncated", "cap"])
def test_gzip_members_inflated_into_memory(hostlib, tmp_path, monkeypatch, variant):
    """the CLI's fast lane for multi-member gzip input: the members are inflated side by side into anonymous memory (the
    chunk parsers then take it like a mapped file).  The text must be exactly what zlib's stream delivers; whatever cannot
    be taken that way -- a single member, a member too large to buffer, a damaged tail, more text than allowed -- is
    refused and left to the sequential reader; a file that is ONE member is inflated in one piece (libdeflate) and taken too"""
    import gzip
    rng = np.random.default_rng(6)
    text = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(synth._ACGT[rng.integers(0, 4, n)]), bytes(rng.integers(35, 70, n).astype(np.uint8)))
                    for i, n in enumerate(rng.integers(50, 3000, 300)))
    if variant == "false_header":
        text = text.replace(b"@r37\n", b"@r37 \x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03 looks like a member\n", 1)
    cuts = sorted(set(int(x) for x in rng.integers(0, len(text), 11)) | {0, len(text)})
    parts = [text[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    level = 0 if variant == "false_header" else 6
    blob = _gz_member(text) if variant.startswith("single") else b"".join(_gz_member(p_, level) for p_ in parts[:5]) + _gz_member(b"") + b"".join(
        _gz_member(p_, level) for p_ in parts[5:])
    if variant in ("trailing_zeros", "single_padded"):
        blob += b"\0" * 1000
    if variant == "truncated":
        blob = blob[:-20]
    if variant == "big_member":
        monkeypatch.setenv("FPLH_GZ_MEMBER_CAP", str(16 << 10))
    p = tmp_path / "in.fq.gz"
    p.write_bytes(blob)
    hostlib.fplh_gunzip_to_memory.restype = C.c_void_p
    hostlib.fplh_gunzip_to_memory.argtypes = [C.c_char_p, C.c_int, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    hostlib.fplh_gunzip_release.argtypes = [C.c_void_p, C.c_uint64]
    size, reserved = C.c_uint64(0), C.c_uint64(0)
    cap = 1000 if variant == "cap" else 2 ** 32
    for threads in (1, 3, 16):
        base = hostlib.fplh_gunzip_to_memory(str(p).encode(), threads, cap, C.byref(size), C.byref(reserved))
        if variant in ("truncated", "big_member", "cap"):
            assert not base
            continue
        if variant.startswith("single") and not base:  # (one member goes through libdeflate in one piece: without it, the stream)
            hostlib.fplh_have_libdeflate.restype = C.c_int
            assert not hostlib.fplh_have_libdeflate()
            continue
        assert base and size.value == len(text)
        got = C.string_at(base, size.value)
        hostlib.fplh_gunzip_release(base, reserved.value)
        assert got == text == gzip.decompress(blob[:-1000] if variant in ("trailing_zeros", "single_padded") else blob)


@pytest.mark.parametrize("source", ["file", "memory"])
@pytest.mark.parametrize("variant", ["clean", "crlf", "junk", "at_quals", "malformed", "long_record", "noeol"])
def test_chunked_reader_equals_sequential(hostlib, tmp_path, monkeypatch, variant, source):
    """the chunk-parallel reader of the CLI (FastqReader::parse_chunk + ChunkedReader): parser threads guess where
    the first record of their chunk starts, the sequencer checks every guess against the chunk in front and parses
    again where they differ -- whatever the chunk size, the records are those of the sequential reader.  source =
    memory: the parsers take the bytes in place from a mapping (how inflated gzip members reach them)"""
    if source == "memory":
        monkeypatch.setenv("FPLH_CHUNK_MEM", "1")
    rng = np.random.default_rng(33)
    reads = []
    for i in range(300):
        n = int(rng.integers(1, 700))
        if variant == "long_record" and i % 60 == 7:
            n = 9000  # a record that runs across several (tiny) chunks
        q = rng.integers(33, 75, n).astype(np.uint8)
        if variant == "at_quals" or rng.random() < 0.3:
            q[0] = ord("@")  # a quality line that starts like a header
            if n > 1 and rng.random() < 0.5:
                q[1] = ord("+")
        reads.append((synth._ACGT[rng.integers(0, 4, n)].astype(np.uint8), q))
    seq, qual, off = synth.pack(reads)
    text, _, _ = hostio.make_fastq(seq, qual, off, crlf=(variant == "crlf"), strand_names=True)
    if variant == "junk":
        lines = text.split(b"\n")
        for i in range(40, len(lines) - 8, 41 * 4):  # stray lines between records
            lines[i:i] = [b"stray line", b"", b"+not a record"]
        text = b"\n".join(lines)
    if variant == "malformed":
        lines = text.split(b"\n")
        lines[4 * 150 + 2] = b"-"  # the '+' line of record 150
        text = b"\n".join(lines)
    if variant == "noeol":
        text = text[:-1]
    p = tmp_path / "in.fq"
    p.write_bytes(text)
    want = _read_all(hostlib, p, 2 ** 62, 2 ** 30)
    hostlib.fplh_batch_read_chunked.restype = C.c_void_p
    hostlib.fplh_batch_read_chunked.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]
    redo_total = 0
    for chunk, threads in ((997, 3), (4096, 5), (50_000, 2), (10 ** 9, 4)):
        redo = C.c_uint64(0)
        b = hostlib.fplh_batch_read_chunked(str(p).encode(), chunk, threads, C.byref(redo))
        assert b
        n, nb = hostlib.fplh_batch_n(b), hostlib.fplh_batch_bytes(b)
        s2 = np.ctypeslib.as_array(C.cast(hostlib.fplh_batch_seq(b), C.POINTER(C.c_uint8)), (max(nb, 1),))[:nb].copy()
        q2 = np.ctypeslib.as_array(C.cast(hostlib.fplh_batch_qual(b), C.POINTER(C.c_uint8)), (max(nb, 1),))[:nb].copy()
        o2 = np.ctypeslib.as_array(C.cast(hostlib.fplh_batch_off(b), C.POINTER(C.c_uint64)), (n + 1,)).copy()
        hostlib.fplh_batch_free(b)
        assert np.array_equal(o2, want[2]) and np.array_equal(s2, want[0]) and np.array_equal(q2, want[1]), (variant, chunk)
        redo_total += redo.value
    if variant == "malformed":
        assert len(want[2]) - 1 == 150
    elif variant not in ("junk",):
        assert np.array_equal(want[2], off)
    if variant == "clean":
        assert redo_total < 40  # guesses are right for well-formed input (a cut inside a header line aside)


@pytest.mark.parametrize("source", ["file", "memory"])
@pytest.mark.parametrize("variant", ["clean", "crlf", "at_quals", "long_record", "noeol"])
def test_chunk_loader_cuts_the_text_at_records(hostlib, tmp_path, monkeypatch, variant, source):
    """the chunk LOADER of the device-parse path (FastqReader::load_chunk_text + ChunkedReader as_text): every chunk's text starts
    at the '@' of a record and ends behind the line break of one -- whatever the chunk size, with records longer than a chunk,
    quality lines that start with '@' / '+', "\r\n" line ends and a missing last line break -- and the chunks follow one another
    without a gap: concatenated they are the file"""
    if source == "memory":
        monkeypatch.setenv("FPLH_CHUNK_MEM", "1")
    rng = np.random.default_rng(35)
    reads = []
    for i in range(300):
        n = int(rng.integers(1, 700))
        if variant == "long_record" and i % 50 == 7:
            n = int(rng.choice([9000, 31000, 2600]))  # records that run across several (tiny) chunks
        q = rng.integers(33, 75, n).astype(np.uint8)
        if variant == "at_quals" or rng.random() < 0.3:
            q[0] = ord("@")
            if n > 1 and rng.random() < 0.5:
                q[1] = ord("+")
        reads.append((synth._ACGT[rng.integers(0, 4, n)].astype(np.uint8), q))
    seq, qual, off = synth.pack(reads)
    text, _, _ = hostio.make_fastq(seq, qual, off, crlf=(variant == "crlf"), strand_names=True)
    if variant == "noeol":
        text = text[:-1]
    p = tmp_path / "in.fq"
    p.write_bytes(text)
    starts = set()
    pos = 0
    for i, line in enumerate(text.split(b"\n")):
        if i % 4 == 0:
            starts.add(pos)
        pos += len(line) + 1
    hostlib.fplh_text_chunk_ranges.restype = C.c_int64
    hostlib.fplh_text_chunk_ranges.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.POINTER(C.c_uint64), C.c_uint64]
    for chunk, threads in ((997, 3), (4096, 5), (50_000, 2), (10 ** 9, 4)):
        cap = len(text) // 500 + 64
        buf = (C.c_uint64 * (2 * cap))()
        n = hostlib.fplh_text_chunk_ranges(str(p).encode(), chunk, threads, buf, cap)
        assert 0 < n <= cap, (variant, chunk, n)
        r = np.frombuffer(buf, np.uint64)[:2 * n].reshape(n, 2).astype(np.int64)
        assert r[0, 0] == 0 and r[-1, 1] == len(text), (variant, chunk, r[0], r[-1])
        assert np.array_equal(r[1:, 0], r[:-1, 1])  # no gap, no overlap
        assert all(int(a) in starts for a in r[:, 0]), (variant, chunk)  # every chunk starts at a record
        assert (r[:, 1] > r[:, 0]).all()
        if chunk < 10 ** 6:
            assert n > 3


def test_truncated_gzip_is_an_error(hostlib, tmp_path):
    """a .fq.gz cut short (an incomplete transfer) must not pass for a complete input: the reference aborts with
    "igzip: unexpected eof" (src/fastqreader.cpp:133-137); the reader here ends the input and reports the error"""
    hostlib.fplh_read_error.restype = C.c_int
    hostlib.fplh_read_error.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    seq, qual, off = synth.ont_like(60, seed=5, median_len=2000)
    text, _, _ = hostio.make_fastq(seq, qual, off)
    whole = gzip.compress(text)
    msg = C.create_string_buffer(512)
    good = tmp_path / "good.fq.gz"
    good.write_bytes(whole)
    assert hostlib.fplh_read_error(str(good).encode(), msg, 512) == 0
    cut = tmp_path / "cut.fq.gz"
    cut.write_bytes(whole[:len(whole) * 2 // 3])
    assert hostlib.fplh_read_error(str(cut).encode(), msg, 512) == 1
    assert b"igzip" in msg.value
    # the same for a file of several members whose last member is cut (the member-parallel path)
    members = b"".join(gzip.compress(text[i:i + len(text) // 5 + 1]) for i in range(0, len(text), len(text) // 5 + 1))
    multi = tmp_path / "multi.fq.gz"
    multi.write_bytes(members[:-40])
    assert hostlib.fplh_read_error(str(multi).encode(), msg, 512) == 1
    bad = bytearray(whole)
    bad[len(bad) // 2] ^= 0xFF  # damaged in the middle
    dmg = tmp_path / "dmg.fq.gz"
    dmg.write_bytes(bytes(bad))
    assert hostlib.fplh_read_error(str(dmg).encode(), msg, 512) == 1
