"""-m gpu: FASTQ text in, records out (fpl_process_text_async / fpl_wait_text, ABI v7): the parse runs on the device.
The line starts against a Python line scan of the same bytes, records and counters against the oracle run on the reads that
scan finds; irregular text (what the reference's sequential reader treats specially, src/fastqreader.cpp:219-347) must be
refused WHOLE -- status FPL_TEXT_IRREGULAR, nothing counted -- so that the host's reader takes the chunk."""
import numpy as np
import pytest

from fastplong_amd import abi, synth
from tests import hostio, parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine_mod():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from fastplong_amd import engine

    return engine


def _line_starts(text):
    """(n, 4) offsets of the four lines of every record of REGULAR text"""
    nl = np.flatnonzero(np.frombuffer(text, np.uint8) == 10)
    assert len(nl) % 4 == 0
    starts = np.concatenate([[0], nl[:-1] + 1]).astype(np.uint32)
    return starts.reshape(-1, 4)


def _pinned(eng, data):
    a = eng.pinned_array(len(data))
    a[:] = np.frombuffer(data, np.uint8)
    return a


OPTS = dict(cut_front=1, cut_tail=1, cut_front_window=5, cut_tail_window=5, polyx=1, complexity_filter=1)


def _check_text(orc, engine_mod, text, seq, qual, off, opts=OPTS):
    cfg = orc.Config(abi.FplOptions.default(**opts), synth.START_ADAPTER, synth.END_ADAPTER)
    C = max(1, int(np.diff(off.astype(np.int64)).max()))
    want_res, want_cnt = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
    eng = engine_mod.Engine(cfg.opt, synth.START_ADAPTER, synth.END_ADAPTER, device=0, max_cycles=C)
    buf = _pinned(eng, text)
    eng.submit_text(buf)
    info, res, lines = eng.wait_text()
    assert info["status"] == abi.FPL_TEXT_OK and info["n_reads"] == len(off) - 1, info
    assert info["n_bases"] == int(off[-1]) and info["max_read_len"] == C and info["n_lines"] == 4 * (len(off) - 1)
    assert np.array_equal(lines, _line_starts(text))
    got_cnt = eng.counters()
    eng.close()
    parity.assert_results_equal(res, want_res, seq, off)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)


@pytest.mark.parametrize("crlf", [False, True])
@pytest.mark.parametrize("strand_names", [False, True])
def test_text_batch_bit_exact(orc, engine_mod, crlf, strand_names):
    seq, qual, off = synth.ont_like(3000, seed=41, median_len=1500, p_middle=0.05)
    text, _, _ = hostio.make_fastq(seq, qual, off, crlf=crlf, strand_names=strand_names)
    _check_text(orc, engine_mod, text, seq, qual, off)


def test_text_batch_short_and_empty_reads(orc, engine_mod):
    """reads of 0, 1, 15, 16, 17 ... bases (the gather's 16-byte steps and tails), a '@' or '+' as first quality byte, a read
    whose name is just '@x'"""
    rng = np.random.default_rng(5)
    reads = []
    for L in [0, 1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 1000, 4095, 4096, 4097, 0, 5] * 8:
        s = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)].copy()
        q = rng.integers(33, 127, L).astype(np.uint8)
        if L:
            q[0] = ord("@") if len(reads) % 2 else ord("+")
        reads.append((s, q))
    seq, qual, off = synth.pack(reads)
    parts = []
    for i, (s, q) in enumerate(reads):
        parts += [b"@x" if i % 7 == 0 else b"@r%d some text" % i, b"\n", s.tobytes(), b"\n+\n", q.tobytes(), b"\n"]
    _check_text(orc, engine_mod, b"".join(parts), seq, qual, off, opts=dict())


def _status(engine_mod, text):
    eng = engine_mod.Engine(abi.FplOptions.default(), synth.START_ADAPTER, synth.END_ADAPTER, device=0, max_cycles=4096)
    eng.submit_text(_pinned(eng, text))
    info, res, lines = eng.wait_text()
    cnt = eng.counters()
    eng.close()
    return info, len(res), cnt


def test_irregular_text_is_refused_whole(engine_mod):
    seq, qual, off = synth.ont_like(50, seed=3, median_len=400)
    good, _, _ = hostio.make_fastq(seq, qual, off)
    recs = good.split(b"\n")[:-1]
    lines4 = [recs[i:i + 4] for i in range(0, len(recs), 4)]

    def join(ls, end=b"\n"):
        return b"".join(l + b"\n" for rec in ls for l in rec)[:-1] + end

    cases = {}
    cases["no final line break"] = good[:-1]
    cases["blank line between records"] = join(lines4[:10]) + b"\n" + join(lines4[10:])
    cases["lone carriage return"] = good.replace(b"\n", b"\r", 1)
    cases["carriage return inside a line"] = good[:40] + b"\r" + good[40:]
    bad = [list(r) for r in lines4]
    bad[7][2] = b"-"
    cases["third line does not start with +"] = join(bad)
    bad = [list(r) for r in lines4]
    bad[20][3] = bad[20][3][:-1]
    cases["fewer qualities than bases"] = join(bad)
    bad = [list(r) for r in lines4]
    bad[0][0] = b"read0 without the at sign"
    cases["header without @"] = join(bad)
    bad = [list(r) for r in lines4]
    del bad[30][1]
    cases["a record of three lines"] = join(bad)
    for name, text in cases.items():
        info, n, cnt = _status(engine_mod, text)
        assert info["status"] == abi.FPL_TEXT_IRREGULAR and n == 0 and info["n_reads"] == 0, (name, info)
        assert not cnt.any(), name  # nothing of the chunk was counted
    info, n, _ = _status(engine_mod, join(bad := [list(r) for r in lines4]))
    assert info["status"] == abi.FPL_TEXT_OK and n == 50
    bad = [list(r) for r in lines4]
    bad[13][3] = bad[13][3] + b"I"
    info, _, _ = _status(engine_mod, join(bad))
    assert info["status"] == abi.FPL_TEXT_IRREGULAR and info["bad_record"] == 13


def test_text_too_many_records(engine_mod):
    """reads of ten bases: more than n_bytes / 64 + 16 records -- the caller's own reader takes such a chunk"""
    text = b"".join(b"@r%d\nACGTACGTAC\n+\nIIIIIIIIII\n" % i for i in range(2000))
    info, n, cnt = _status(engine_mod, text)
    assert info["status"] == abi.FPL_TEXT_TOO_MANY and n == 0 and not cnt.any()


def test_text_and_csr_batches_share_the_pipeline(orc, engine_mod):
    """two batches in flight, text and CSR submissions in turn, an irregular chunk in between: every batch's records and the
    counters of all of them together are the oracle's"""
    cfg = orc.Config(abi.FplOptions.default(**OPTS), synth.START_ADAPTER, synth.END_ADAPTER)
    batches = [synth.ont_like(800 + 100 * k, seed=60 + k, median_len=900 + 200 * k, p_middle=0.05) for k in range(6)]
    C = max(int(np.diff(b[2].astype(np.int64)).max()) for b in batches)
    eng = engine_mod.Engine(cfg.opt, synth.START_ADAPTER, synth.END_ADAPTER, device=0, max_cycles=C)
    want_cnt = None
    pending = []
    got = []

    def collect():
        kind, k, keep = pending.pop(0)
        if kind == "text":
            info, res, lines = eng.wait_text()
            got.append((k, info, res))
        else:
            eng.wait()
            got.append((k, None, keep[3].copy()))

    for k, (seq, qual, off) in enumerate(batches):
        r, c = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
        if k != 3:
            want_cnt = c if want_cnt is None else want_cnt + c
        if eng.in_flight() == abi.FPL_MAX_IN_FLIGHT:
            collect()
        if k % 2 == 0 or k == 3:
            text, _, _ = hostio.make_fastq(seq, qual, off, crlf=(k == 4))
            if k == 3:
                text = text[:-1]  # irregular: refused, not counted
            buf = _pinned(eng, text)
            eng.submit_text(buf)
            pending.append(("text", k, buf))
        else:
            ps, pq = _pinned(eng, seq.tobytes()), _pinned(eng, qual.tobytes())
            po = eng.pinned_array(len(off), np.uint64)
            po[:] = off
            rr = np.zeros(len(off) - 1, dtype=abi.RESULT_DTYPE)
            eng.submit_host(ps, pq, po, rr)
            pending.append(("csr", k, (ps, pq, po, rr)))
    while pending:
        collect()
    got_cnt = eng.counters()
    eng.close()
    for k, info, res in got:
        seq, qual, off = batches[k]
        if k == 3:
            assert info["status"] == abi.FPL_TEXT_IRREGULAR and len(res) == 0
            continue
        if info is not None:
            assert info["status"] == abi.FPL_TEXT_OK
        want_res, _ = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
        parity.assert_results_equal(res, want_res, seq, off)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)


def test_text_batch_of_bench_size_properties(engine_mod):
    """a 300 MB chunk (the gather at bandwidth, 32-bit line positions far from their limit): every record's line starts are
    increasing, the lengths add up, the counters say every read and base went in"""
    seq, qual, off = synth.ont_like(20000, seed=9, median_len=7000)
    text, _, _ = hostio.make_fastq(seq, qual, off)
    C = int(np.diff(off.astype(np.int64)).max())
    eng = engine_mod.Engine(abi.FplOptions.default(**OPTS), synth.START_ADAPTER, synth.END_ADAPTER, device=0, max_cycles=C)
    buf = _pinned(eng, text)
    eng.submit_text(buf)
    info, res, lines = eng.wait_text()
    v = abi.CountersView(eng.counters(), C, 2)
    eng.close()
    assert info["status"] == abi.FPL_TEXT_OK and info["n_reads"] == 20000 and info["n_bases"] == int(off[-1])
    assert np.array_equal(lines, _line_starts(text))
    assert int(v.pre.reads) == 20000 and int(v.pre.length_sum) == int(off[-1])


def test_peek_start_and_cancel(orc, engine_mod):
    """fpl_peek_text gives the parse's verdict of the next pending batch with nothing counted yet; fpl_cancel_text keeps it from
    running (its wait says FPL_TEXT_CANCELLED); fpl_start_text enqueues a batch's kernels ahead of its wait; a batch that is
    waited for after a peek is complete and exact"""
    seq, qual, off = synth.ont_like(1500, seed=77, median_len=1200, p_middle=0.05)
    text, _, _ = hostio.make_fastq(seq, qual, off)
    cfg = orc.Config(abi.FplOptions.default(**OPTS), synth.START_ADAPTER, synth.END_ADAPTER)
    C = int(np.diff(off.astype(np.int64)).max())
    want_res, want_cnt = orc.process_batch(cfg, seq, qual, off, max_cycles=C)
    eng = engine_mod.Engine(cfg.opt, synth.START_ADAPTER, synth.END_ADAPTER, device=0, max_cycles=C)
    buf = _pinned(eng, text)
    bad = _pinned(eng, text[:-1])
    eng.submit_text(buf)
    eng.submit_text(bad)
    eng.submit_text(buf)
    info = eng.peek_text()  # the first
    assert info["status"] == abi.FPL_TEXT_OK and info["n_reads"] == 1500 and info["n_bases"] == int(off[-1])
    assert not eng.counters().any() and eng.in_flight() == 3  # peeked at, not run
    eng.cancel_text()  # the first will not run
    assert eng.peek_text()["status"] == abi.FPL_TEXT_IRREGULAR  # the second is the next pending one
    eng.cancel_text()
    assert eng.peek_text()["n_reads"] == 1500  # the third
    eng.start_text()  # its kernels, ahead of the waits for the two in front of it
    for _ in range(2):
        info, res, lines = eng.wait_text()
        assert info["status"] == abi.FPL_TEXT_CANCELLED and len(res) == 0
    info, res, lines = eng.wait_text()
    got_cnt = eng.counters()
    assert eng.in_flight() == 0
    eng.close()
    assert info["status"] == abi.FPL_TEXT_OK
    parity.assert_results_equal(res, want_res, seq, off)
    parity.assert_counters_equal(got_cnt, want_cnt, C, cfg.n_adapters)  # the one batch that ran, once
