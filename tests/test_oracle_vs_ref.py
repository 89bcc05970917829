"""Differential pin of the oracle against the REAL reference objects in oracle/_ref
(built in place from /root/reference/src by oracle/Makefile; skipped where that binary is
absent).  Covers every hot-path function of the reference that builds with this image's
toolchain: edit_distance, Filter::trimAndCut, PolyX::trimPolyX, Filter::passFilter,
Read::trimFront/resize."""
import numpy as np
import pytest

from fastplong_amd import abi, synth


def rand_seq(rng, n, alphabet="ACGT", p_n=0.0):
    s = "".join(alphabet[i] for i in rng.integers(0, len(alphabet), n))
    if p_n > 0 and n > 0:
        a = list(s)
        for i in np.nonzero(rng.random(n) < p_n)[0]:
            a[i] = "N"
        s = "".join(a)
    return s


def rand_qual(rng, n, mu=20, sigma=10):
    q = np.clip(np.rint(rng.normal(mu, sigma, n)), 0, 60).astype(int) + 33
    return "".join(chr(c) for c in q)


def mutate(rng, s, rate):
    out = []
    for ch in s:
        x = rng.random()
        if x < rate / 3:
            continue
        if x < 2 * rate / 3:
            out.append("ACGT"[rng.integers(4)])
            out.append(ch)
        elif x < rate:
            out.append("ACGT"[rng.integers(4)])
        else:
            out.append(ch)
    return "".join(out)


def test_edit_distance_vs_reference(orc, ref):
    rng = np.random.default_rng(11)
    cases = []
    lens = [0, 1, 2, 15, 16, 17, 24, 63, 64, 65, 127, 128, 129, 151, 191, 192, 193, 256, 320, 640, 641, 700]
    for la in lens:
        for _ in range(6):
            a = rand_seq(rng, la, "ACGTN")
            k = rng.integers(0, 4)
            if k == 0:
                b = mutate(rng, a, float(rng.choice([0.02, 0.1, 0.3])))
            elif k == 1:
                b = rand_seq(rng, int(rng.choice(lens)), "ACGTN")
            elif k == 2:
                b = a[int(rng.integers(0, la + 1)):] + rand_seq(rng, int(rng.integers(0, 5)))
            else:
                b = a
            cases.append((a, b))
    # equal-length pairs as the hot path calls it (adapter x read window)
    for _ in range(400):
        n = int(rng.integers(1, 80))
        a = rand_seq(rng, n)
        cases.append((a, mutate(rng, a, 0.25)[:n].ljust(n, "A")))
    out = ref.run(["ED %s %s" % (ref.s(a), ref.s(b)) for a, b in cases]).split()
    assert len(out) == len(cases)
    for (a, b), want in zip(cases, out):
        assert orc.edit_distance(a, b) == int(want), (a, b)


def _tac_cases(rng, n):
    for _ in range(n):
        L = int(rng.choice([0, 1, 2, 3, 4, 5, 8, 9, 10, 20, 50, 120])) if rng.random() < 0.5 else int(rng.integers(0, 300))
        seq = rand_seq(rng, L, "ACGT", p_n=float(rng.choice([0, 0.05, 0.5])))
        mu = float(rng.choice([5, 15, 20, 25, 40]))
        qual = rand_qual(rng, L, mu, 8)
        if L > 4 and rng.random() < 0.3:  # N / low-quality ends
            a = int(rng.integers(0, L // 2 + 1))
            seq = "N" * a + seq[a:]
            qual = "".join("#" for _ in range(a)) + qual[a:]
        if L > 4 and rng.random() < 0.3:
            a = int(rng.integers(0, L // 2 + 1))
            seq = seq[:L - a] + "N" * a
        front = int(rng.choice([0, 0, 1, 3, 10]))
        tail = int(rng.choice([0, 0, 1, 2, 7]))
        cf, ct = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        wf, wt = int(rng.choice([1, 2, 4, 5, 10, 50])), int(rng.choice([1, 2, 4, 5, 10, 50]))
        qf, qt = int(rng.choice([1, 10, 15, 20, 30])), int(rng.choice([1, 10, 15, 20, 30]))
        yield (front, tail, cf, ct, wf, qf, wt, qt, seq, qual)


def test_trim_and_cut_vs_reference(orc, ref):
    rng = np.random.default_rng(12)
    cases = list(_tac_cases(rng, 3000))
    lines = ["TAC %d %d %d %d %d %d %d %d %s %s" % (c[:8] + (ref.s(c[8]), ref.s(c[9]))) for c in cases]
    out = ref.run(lines).splitlines()
    assert len(out) == len(cases)
    n_null = 0
    for c, want in zip(cases, out):
        front, tail, cf, ct, wf, qf, wt, qt, seq, qual = c
        opt = abi.FplOptions.default(trim_front=front, trim_tail=tail, cut_front=cf, cut_tail=ct,
                                     cut_front_window=wf, cut_front_quality=qf,
                                     cut_tail_window=wt, cut_tail_quality=qt)
        got = orc.trim_and_cut(seq, qual, opt)
        if want == "NULL":
            n_null += 1
            assert got is None, c
        else:
            ft, s, q = want.split(" ")
            assert got is not None, c
            assert (got[0], got[1], got[2]) == (int(ft), s[1:], q[1:]), c
    assert 0 < n_null < len(cases)


def test_trim_polyx_vs_reference(orc, ref):
    rng = np.random.default_rng(13)
    cases = []
    for _ in range(3000):
        L = int(rng.integers(1, 200))
        seq = rand_seq(rng, L, "ACGT", p_n=float(rng.choice([0, 0.02])))
        k = int(rng.integers(0, L))  # polyX tail of k bases with a few mismatches, read not entirely polyX
        if rng.random() < 0.8 and k > 0:
            base = "ACGTN"[rng.integers(5)]
            tail = [base] * k
            for i in np.nonzero(rng.random(k) < float(rng.choice([0, 0.05, 0.15])))[0]:
                tail[i] = "ACGTN"[rng.integers(5)]
            # keep the first base of the read different from every tail base so the reference's
            # scan always terminates before index -1 (which is undefined behaviour there)
            seq = seq[:L - k] + "".join(tail)
        # guard base: a lower-case char never counts for any poly base (src/polyx.cpp:24-44)
        seq = "x" + "c" * 6 + seq
        cases.append((int(rng.choice([2, 5, 8, 10, 10, 15, 30])), seq))
    out = ref.run(["PX %d %s %s" % (m, ref.s(s), ref.s("I" * len(s))) for m, s in cases]).splitlines()
    n_called = 0
    for (m, s), want in zip(cases, out):
        wseq, wreads, wbases = want.split(" ")
        got_seq, called, poly, tl = orc.trim_polyx(s, None, m)
        assert got_seq == wseq[1:], (m, s)
        assert called == int(wreads), (m, s)
        assert (tl if called else 0) == int(wbases), (m, s)
        n_called += called
    assert n_called > 100


def test_pass_filter_vs_reference(orc, ref):
    rng = np.random.default_rng(14)
    cases = []
    for _ in range(4000):
        L = int(rng.choice([0, 1, 2, 10, 19, 20, 21, 100])) if rng.random() < 0.4 else int(rng.integers(0, 400))
        kind = rng.integers(0, 4)
        if kind == 0:
            seq = rand_seq(rng, L, "ACGT", p_n=float(rng.choice([0, 0.05, 0.1, 0.12, 0.3])))
        elif kind == 1:  # low complexity
            seq = ("".join(ch * int(rng.integers(1, 8)) for ch in rand_seq(rng, L)))[:L]
        else:
            seq = rand_seq(rng, L)
        qual = rand_qual(rng, L, float(rng.choice([5, 12, 15, 17, 20, 30])), float(rng.choice([1, 5, 10])))
        o = dict(
            qual_filter=int(rng.random() < 0.8), qualified_qual=33 + int(rng.choice([0, 10, 15, 20, 30])),
            unqualified_percent_limit=int(rng.choice([0, 10, 40, 50, 100])),
            n_base_limit=int(rng.choice([1000000, 0, 1, 5])), n_base_percent_limit=int(rng.choice([0, 5, 10, 100])),
            avg_qual_req=int(rng.choice([0, 0, 10, 15, 20])),
            length_filter=int(rng.random() < 0.8), required_length=int(rng.choice([0, 15, 20, 50])),
            max_length=int(rng.choice([0, 0, 100, 350])),
            complexity_filter=int(rng.random() < 0.5), complexity_percent=int(rng.choice([0, 10, 30, 33, 50, 100])),
        )
        cases.append((o, seq, qual))
    lines = []
    for o, seq, qual in cases:
        lines.append("PF %d %d %d %d %d %d %d %d %d %d %d %s %s" % (
            o["qual_filter"], o["qualified_qual"], o["unqualified_percent_limit"], o["n_base_limit"],
            o["n_base_percent_limit"], o["avg_qual_req"], o["length_filter"], o["required_length"],
            o["max_length"], o["complexity_filter"], o["complexity_percent"], ref.s(seq), ref.s(qual)))
    out = ref.run(lines).split()
    seen = set()
    for (o, seq, qual), want in zip(cases, out):
        got = orc.pass_filter(seq, qual, abi.FplOptions.default(**o))
        assert got == int(want), (o, seq, qual)
        seen.add(got)
    assert seen == {0, 12, 16, 17, 20, 24}


def test_read_offset_ops_vs_reference(ref):
    """Read::trimFront / Read::resize (src/read.cpp:62-73) against the window arithmetic the
    oracle and the kernels use."""
    rng = np.random.default_rng(15)
    cases = []
    for _ in range(300):
        L = int(rng.integers(0, 40))
        cases.append((rng.choice(["TF", "RS"]), int(rng.integers(-5, 50)), rand_seq(rng, L), rand_qual(rng, L)))
    out = ref.run(["%s %d %s %s" % (op, n, ref.s(s), ref.s(q)) for op, n, s, q in cases]).splitlines()
    for (op, n, s, q), want in zip(cases, out):
        ws, wq = want.split(" ")
        L = len(s)
        if op == "TF":
            k = min(L - 1, n)
            exp = ("", "") if k < 0 else (s[k:], q[k:])
        else:
            exp = (s, q) if (n > L or n < 0) else (s[:n], q[:n])
        assert (ws[1:], wq[1:]) == exp, (op, n, s)


# ---- --break / --mask (src/seprocessor.cpp:234-262) -----------------------------------------------
def _lowq_read(rng, n, p_low=0.3):
    """qualities with stretches far below the threshold, so that regions start, stop and chain"""
    q = np.clip(np.round(rng.normal(30, 4, n)), 5, 50)
    pos = 0
    while pos < n:
        run = int(rng.integers(5, 120))
        if rng.random() < p_low:
            q[pos:pos + run] = np.clip(np.round(rng.normal(6, 3, min(run, n - pos))), 2, 40)
        pos += run
    return (q + 33).astype(np.uint8)


@pytest.mark.parametrize("window,quality", [(5, 10), (20, 15), (50, 10), (100, 10), (7, 30)])
def test_detect_low_quality_regions_matches_reference(orc, ref, window, quality):
    rng = np.random.default_rng(window * 100 + quality)
    quals = [_lowq_read(rng, int(n)) for n in list(rng.integers(1, 400, 40)) + [window - 1, window, window + 1, 2 * window]]
    lines = ["LQR %d %d %s %s" % (window, quality, ref.s(b"A" * len(q)), ref.s(q.tobytes())) for q in quals]
    out = ref.run(lines).splitlines()
    n_regions = 0
    for q, line in zip(quals, out):
        t = [int(x) for x in line.split()]
        want = list(zip(t[1::2], t[2::2]))
        assert len(want) == t[0]
        assert orc.detect_low_quality_regions(q.tobytes(), window, quality) == want
        n_regions += len(want)
    assert n_regions > 10


@pytest.mark.parametrize("be,me", [(1, 0), (0, 1), (1, 1)])
def test_break_mask_stage_matches_reference(orc, ref, be, me):
    """the oracle's fragment list, rendered the way the host formats it, against the real
    detectLowQualityRegions + breakByRegions + maskRegionWithN + appendToString"""
    rng = np.random.default_rng(7 + 2 * be + me)
    bw, bq, mw, mq = 20, 12, 8, 14
    opt = abi.FplOptions.default(adapter_enabled=0, qual_filter=0, length_filter=0, break_enabled=be, break_window=bw,
                                 break_quality=bq, mask_enabled=me, mask_window=mw, mask_quality=mq)
    cfg = orc.Config(opt)
    reads = []
    for n in list(rng.integers(1, 500, 60)):
        reads.append((synth._ACGT[rng.integers(0, 4, int(n))].astype(np.uint8), _lowq_read(rng, int(n))))
    seq, qual, off = synth.pack(reads)
    res, cnt, frags, regs = orc.process_batch_ex(cfg, seq, qual, off)
    lines = ["BRK %d %d %d %d %d %d %s %s %s %s" % (be, bw, bq, me, mw, mq, ref.s(b"@read%d" % i), ref.s(s.tobytes()),
                                                    ref.s(b"+"), ref.s(q.tobytes())) for i, (s, q) in enumerate(reads)]
    raw = ref.run(lines).encode("latin-1")
    pos, total_frag, total_masked = 0, 0, 0
    for i, (s, q) in enumerate(reads):
        nl = raw.index(b"\n", pos)
        n_out, nbytes = [int(x) for x in raw[pos:nl].split()]
        want = raw[nl + 1:nl + 1 + nbytes]
        pos = nl + 1 + nbytes
        mine = frags[frags["read"] == i]
        assert len(mine) == n_out and list(mine["seq_no"]) == list(range(n_out))
        got = []
        for f in mine:
            sb = bytearray(s[f["start"]:f["start"] + f["len"]].tobytes())
            for r in regs[f["region_first"]:f["region_first"] + f["region_count"]]:
                a = int(r["start"]) - int(f["start"])
                sb[a:a + int(r["len"])] = b"N" * int(r["len"])
                total_masked += int(r["len"])
            name = b"@" + (b"r%d-" % f["break_no"] if f["break_no"] else b"") + b"read%d" % i
            got += [name, b"\n", bytes(sb), b"\n+\n", q[f["start"]:f["start"] + f["len"]].tobytes(), b"\n"]
        assert b"".join(got) == want
        total_frag += n_out
    assert (frags["break_no"] > 0).any() if be else total_frag == len(reads)
    assert (total_masked > 0) == bool(me)
