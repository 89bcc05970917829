"""Seeded synthetic long-read generators.

numpy generators build CSR batches (seq, qual, off) on the host for tests and small runs;
`device_batch` builds the same kind of batch directly in HBM with torch for bench.py.
The README adapter pair of the reference (README.md:145) is the default.
"""
import numpy as np

START_ADAPTER = "AAGGATTCATTCCCACGGTAACAC"
END_ADAPTER = "GTGTTACCGTGGGAATGAATCCTT"
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def revcomp(s):
    """Sequence::reverseComplement (reference src/sequence.cpp:29-77): A<->T, C<->G, else N."""
    m = {"A": "T", "T": "A", "C": "G", "G": "C", "a": "T", "t": "A", "c": "G", "g": "C"}
    return "".join(m.get(c, "N") for c in reversed(s))


def _mutate(rng, ad, err):
    """substitution / insertion / deletion errors at rate `err` per base"""
    out = []
    for ch in ad:
        x = rng.random()
        if x < err / 3:
            continue  # deletion
        if x < 2 * err / 3:
            out.append(int(_ACGT[rng.integers(4)]))  # insertion
            out.append(ch)
        elif x < err:
            out.append(int(_ACGT[rng.integers(4)]))  # substitution (may be silent)
        else:
            out.append(ch)
    return np.array(out, dtype=np.uint8)


def _qual(rng, n, mu, sigma):
    q = np.clip(np.rint(rng.normal(mu, sigma, n)), 2, 50).astype(np.uint8) + 33
    return q


def wide_qualities(qual, off, seed, share=0.7):
    """Overwrite the qualities of `share` of the reads with bytes from the WHOLE FASTQ range '!'..'~' (33..126; `_qual` stays
    within Q2..Q50): per read one of -- uniform over 33..126, all '~' (what PacBio HiFi reads mostly carry), all '!', the top of
    the range (84..126), the bottom (33..35), runs of 1..300 bytes alternating between the two ends, '~' with a sprinkle of lower
    bytes.  Returns a new array; the bases and offsets are untouched."""
    rng = np.random.default_rng(seed)
    q = qual.copy()
    for i in range(len(off) - 1):
        a, b = int(off[i]), int(off[i + 1])
        L = b - a
        if L == 0 or rng.random() >= share:
            continue
        kind = int(rng.integers(0, 7))
        if kind == 0:
            q[a:b] = rng.integers(33, 127, L)
        elif kind == 1:
            q[a:b] = 126
        elif kind == 2:
            q[a:b] = 33
        elif kind == 3:
            q[a:b] = rng.integers(84, 127, L)
        elif kind == 4:
            q[a:b] = rng.integers(33, 36, L)
        elif kind == 5:
            p, hi = a, bool(rng.integers(2))
            while p < b:
                run = int(rng.integers(1, 301))
                q[p:min(b, p + run)] = rng.integers(120, 127) if hi else rng.integers(33, 36)
                p, hi = p + run, not hi
        else:
            q[a:b] = 126
            m = rng.random(L) < 0.05
            q[a:b][m] = rng.integers(33, 127, int(m.sum()))
    return q


def make_read(rng, length, mu=18.0, sigma=8.0, start_ad=None, end_ad=None, p_start=0.7, p_end=0.6,
              err=0.10, p_polya=0.05, p_middle=0.01, n_rate=0.001, lead_max=30):
    """One ONT-like read: body of iid ACGT (+N), optional noisy adapters at the ends after
    0..lead_max random bases, optional polyA tail, optional adapter in the middle."""
    body = _ACGT[rng.integers(0, 4, length)]
    if n_rate > 0:
        body = body.copy()
        body[rng.random(length) < n_rate] = ord("N")
    parts = []
    if start_ad is not None and rng.random() < p_start:
        parts.append(_ACGT[rng.integers(0, 4, rng.integers(0, lead_max + 1))])
        parts.append(_mutate(rng, start_ad, err))
    if p_middle > 0 and rng.random() < p_middle and length > 200:
        cut = int(rng.integers(100, length - 100))
        ad = start_ad if (rng.random() < 0.5 or end_ad is None) else end_ad
        if ad is not None:
            parts += [body[:cut], _mutate(rng, ad, err), body[cut:]]
        else:
            parts.append(body)
    else:
        parts.append(body)
    if rng.random() < p_polya:
        parts.append(np.full(rng.integers(10, 41), ord("A"), dtype=np.uint8))
    if end_ad is not None and rng.random() < p_end:
        parts.append(_mutate(rng, end_ad, err))
        parts.append(_ACGT[rng.integers(0, 4, rng.integers(0, lead_max + 1))])
    seq = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
    return seq, _qual(rng, len(seq), mu, sigma)


def pack(reads):
    """list of (seq, qual) -> CSR (seq, qual, off)."""
    n = len(reads)
    lens = np.array([len(s) for s, _ in reads], dtype=np.int64)
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    seq = np.concatenate([s for s, _ in reads]) if n else np.zeros(0, np.uint8)
    qual = np.concatenate([q for _, q in reads]) if n else np.zeros(0, np.uint8)
    return seq.astype(np.uint8), qual.astype(np.uint8), off


def ont_like(n_reads, seed=1, median_len=8000, sigma_len=0.5, min_len=50, max_len=None,
             start_adapter=START_ADAPTER, end_adapter=END_ADAPTER, **kw):
    """Config 2/3 style batch: lognormal lengths (median 8 kb, sigma 0.5 => N50 ~ 10 kb)."""
    rng = np.random.default_rng(seed)
    sa = np.frombuffer(start_adapter.encode(), np.uint8) if start_adapter else None
    ea = np.frombuffer(end_adapter.encode(), np.uint8) if end_adapter else None
    lens = np.rint(rng.lognormal(np.log(median_len), sigma_len, n_reads)).astype(np.int64)
    lens = np.clip(lens, min_len, max_len if max_len else None)
    reads = [make_read(rng, int(L), start_ad=sa, end_ad=ea, **kw) for L in lens]
    return pack(reads)


def adversarial(n_reads, seed=7, start_adapter=START_ADAPTER, end_adapter=END_ADAPTER, fasta=()):
    """Short nasty reads: lengths 0..700 incl. 15/16/17 and 199..217, homopolymers, N runs,
    truncated / mutated adapters at 0..190 bases from either end, middle adapters, low and
    high quality stretches, lower-case and non-ACGT bytes."""
    rng = np.random.default_rng(seed)
    sa = np.frombuffer(start_adapter.encode(), np.uint8) if start_adapter else None
    ea = np.frombuffer(end_adapter.encode(), np.uint8) if end_adapter else None
    fas = [np.frombuffer(a.encode() if isinstance(a, str) else a, np.uint8) for a in fasta]
    special = [0, 1, 2, 3, 4, 5, 15, 16, 17, 18, 31, 32, 33, 63, 64, 65] + list(range(196, 220))
    reads = []
    for i in range(n_reads):
        kind = rng.integers(0, 12)
        L = int(special[rng.integers(len(special))]) if rng.random() < 0.35 else int(rng.integers(0, 700))
        seq = _ACGT[rng.integers(0, 4, L)].copy()
        mu = float(rng.choice([5, 12, 18, 25, 35]))
        qual = _qual(rng, L, mu, 8.0)
        if kind == 0 and L > 0:  # homopolymer / polyX tail with a few mismatches
            k = int(rng.integers(1, L + 1))
            seq[L - k:] = _ACGT[rng.integers(4)]
            mm = rng.random(k) < 0.08
            seq[L - k:][mm] = _ACGT[rng.integers(0, 4, int(mm.sum()))]
        elif kind == 1 and L > 0:  # N runs at the ends / inside
            for _ in range(int(rng.integers(1, 4))):
                a = int(rng.integers(0, L))
                seq[a:a + int(rng.integers(1, 30))] = ord("N")
        elif kind in (2, 3, 4, 5):  # adapters (maybe truncated / mutated) near the ends
            pool = [x for x in [sa, ea] + fas if x is not None]
            if pool and L > 0:
                for side in range(2):
                    if rng.random() < 0.7:
                        ad = pool[rng.integers(len(pool))]
                        ad = _mutate(rng, ad, float(rng.choice([0.0, 0.05, 0.1, 0.2, 0.35])))
                        if rng.random() < 0.4 and len(ad) > 4:
                            cut = int(rng.integers(1, len(ad)))
                            ad = ad[cut:] if side == 0 else ad[:cut]
                        d = int(rng.integers(0, 191)) if rng.random() < 0.5 else int(rng.integers(0, 12))
                        if side == 0:
                            seq = np.concatenate([_ACGT[rng.integers(0, 4, d)], ad, seq])
                        else:
                            seq = np.concatenate([seq, ad, _ACGT[rng.integers(0, 4, d)]])
                L = len(seq)
                qual = _qual(rng, L, mu, 8.0)
        elif kind == 6 and L > 60:  # adapter in the middle
            pool = [x for x in [sa, ea] if x is not None]
            if pool:
                ad = _mutate(rng, pool[rng.integers(len(pool))], float(rng.choice([0.0, 0.1, 0.2])))
                cut = int(rng.integers(0, L))
                seq = np.concatenate([seq[:cut], ad, seq[cut:]])
                L = len(seq)
                qual = _qual(rng, L, mu, 8.0)
        elif kind == 7 and L > 0:  # low-quality ends (exercise cut_front/cut_tail), high inside
            a, b = int(rng.integers(0, L // 2 + 1)), int(rng.integers(0, L // 2 + 1))
            qual[:a] = 33 + rng.integers(2, 10, a)
            qual[L - b:] = 33 + rng.integers(2, 10, b)
        elif kind == 8 and L > 0:  # odd bytes: lower case, U, other letters
            m = rng.random(L) < 0.1
            seq[m] = np.frombuffer(b"acgtnURYKMSWBDHV*-", np.uint8)[rng.integers(0, 18, int(m.sum()))]
        elif kind == 9 and L > 0:  # low complexity
            unit = _ACGT[rng.integers(0, 4, int(rng.integers(1, 4)))]
            seq = np.resize(np.repeat(unit, int(rng.integers(2, 9))), L).copy()
        reads.append((seq.astype(np.uint8), qual.astype(np.uint8)))
    return pack(reads)


def hifi_like(n_reads, seed=5, mean_len=20000, sd_len=2000, n_adapters=64):
    """Config 5 style: N(20 kb, 2 kb) Q30-ish reads and a FASTA of random 30-45-mers (names
    ad00..adNN); 30 % of reads carry one adapter (10 % errors) at an end, 1 % in the middle."""
    rng = np.random.default_rng(seed)
    ads = ["".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(30, 46)))) for _ in range(n_adapters)]
    adb = [np.frombuffer(a.encode(), np.uint8) for a in ads]
    reads = []
    for _ in range(n_reads):
        L = max(100, int(rng.normal(mean_len, sd_len)))
        seq = _ACGT[rng.integers(0, 4, L)]
        x = rng.random()
        if x < 0.30:
            ad = _mutate(rng, adb[rng.integers(n_adapters)], 0.10)
            seq = np.concatenate([ad, seq]) if rng.random() < 0.5 else np.concatenate([seq, ad])
        elif x < 0.31:
            cut = int(rng.integers(100, L - 50))
            seq = np.concatenate([seq[:cut], _mutate(rng, adb[rng.integers(n_adapters)], 0.10), seq[cut:]])
        reads.append((seq.astype(np.uint8), _qual(rng, len(seq), 35.0, 6.0)))
    s, q, off = pack(reads)
    return s, q, off, ads


def to_fastq(seq, qual, off, prefix="read"):
    """CSR batch -> FASTQ text (bytes) with names @<prefix><i>."""
    out = []
    for i in range(len(off) - 1):
        a, b = int(off[i]), int(off[i + 1])
        out.append(b"@%s%d some comment\n" % (prefix.encode(), i))
        out.append(seq[a:b].tobytes() + b"\n+\n" + qual[a:b].tobytes() + b"\n")
    return b"".join(out)


def _variant_pool(rng, ad, err, n_variants):
    """n_variants noisy copies of an adapter as a padded matrix + lengths"""
    vs = [_mutate(rng, ad, err) for _ in range(n_variants)]
    lmax = max(1, max(len(v) for v in vs))
    mat = np.zeros((n_variants, lmax), np.uint8)
    lens = np.zeros(n_variants, np.int64)
    for i, v in enumerate(vs):
        mat[i, :len(v)] = v
        lens[i] = len(v)
    return mat, lens


def device_batch(n_reads, seed=1, median_len=8000, sigma_len=0.5, min_len=50, max_len=None,
                 start_adapter=START_ADAPTER, end_adapter=END_ADAPTER, p_start=0.7, p_end=0.6,
                 err=0.10, p_polya=0.05, p_middle=0.01, mu=18.0, sigma=8.0, device="cuda",
                 chunk=1 << 27, n_variants=2048):
    """Build an ONT-like CSR batch directly in HBM: lognormal
    lengths, iid ACGT + 0.1 % N, quality clamp(round(N(mu, sigma)), 2, 50) + 33, noisy start /
    end adapters after 0..30 leading bases, polyA tails, adapters in the middle.  Bodies and
    qualities are generated on the device in chunks; decorations OVERWRITE bases (lengths stay
    as drawn) and are scattered in with vectorised index_put.
    Returns (seq u8 [n_bytes], qual u8 [n_bytes], off int64 [n+1], max_len)."""
    import torch

    rng = np.random.default_rng(seed)
    lens = np.rint(rng.lognormal(np.log(median_len), sigma_len, n_reads)).astype(np.int64)
    lens = np.clip(lens, max(min_len, 300), max_len if max_len else None)
    off = np.zeros(n_reads + 1, dtype=np.int64)
    off[1:] = np.cumsum(lens)
    total = int(off[-1])
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    seq = torch.empty(total, dtype=torch.uint8, device=device)
    qual = torch.empty(total, dtype=torch.uint8, device=device)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    for a in range(0, total, chunk):
        b = min(total, a + chunk)
        idx = torch.randint(0, 4, (b - a,), generator=g, device=device, dtype=torch.int64)
        s = lut[idx]
        del idx
        s[torch.rand(b - a, generator=g, device=device) < 0.001] = ord("N")
        seq[a:b] = s
        del s
        q = torch.randn(b - a, generator=g, device=device) * sigma + mu
        qual[a:b] = (q.round_().clamp_(2, 50) + 33).to(torch.uint8)
        del q

    def scatter(read_idx, pos_in_read, mat, vlen, variant):
        """seq[off[r] + pos + j] = mat[variant, j] for j < vlen[variant]"""
        if len(read_idx) == 0:
            return
        base = torch.from_numpy(off[read_idx] + pos_in_read).to(device)
        var = torch.from_numpy(variant).to(device)
        m = torch.from_numpy(mat).to(device)
        vl = torch.from_numpy(vlen).to(device)
        j = torch.arange(m.shape[1], device=device)
        flat = base[:, None] + j[None, :]
        mask = j[None, :] < vl[var][:, None]
        seq[flat[mask]] = m[var][mask]

    u = rng.random((n_reads, 4))
    lead = rng.integers(0, 31, (n_reads, 2))
    tail_used = np.zeros(n_reads, np.int64)
    if end_adapter:
        mat, vlen = _variant_pool(rng, np.frombuffer(end_adapter.encode(), np.uint8), err, n_variants)
        ri = np.nonzero(u[:, 1] < p_end)[0]
        var = rng.integers(0, n_variants, len(ri))
        pos = lens[ri] - lead[ri, 1] - vlen[var]
        scatter(ri, pos, mat, vlen, var)
        tail_used[ri] = lens[ri] - pos
    if p_polya > 0:
        ri = np.nonzero(u[:, 2] < p_polya)[0]
        k = rng.integers(10, 41, len(ri))
        mat = np.full((1, 40), ord("A"), np.uint8)
        pos = lens[ri] - tail_used[ri] - k
        # one "variant" per distinct length: reuse scatter with per-row lengths
        scatter(ri, pos, np.repeat(mat, 31, axis=0), np.arange(10, 41, dtype=np.int64), (k - 10).astype(np.int64))
    if start_adapter:
        mat, vlen = _variant_pool(rng, np.frombuffer(start_adapter.encode(), np.uint8), err, n_variants)
        ri = np.nonzero(u[:, 0] < p_start)[0]
        var = rng.integers(0, n_variants, len(ri))
        scatter(ri, lead[ri, 0].astype(np.int64), mat, vlen, var)
        ri = np.nonzero((u[:, 3] < p_middle) & (lens > 600))[0]
        if len(ri):
            var = rng.integers(0, n_variants, len(ri))
            pos = 250 + (rng.random(len(ri)) * (lens[ri] - 550)).astype(np.int64)
            scatter(ri, pos, mat, vlen, var)
    off_t = torch.from_numpy(off).to(device)
    return seq, qual, off_t, int(lens.max())


def device_batch_hifi(n_reads, seed=5, mean_len=20000, sd_len=2000, n_adapters=64, mu=35.0, sigma=6.0, device="cuda",
                      chunk=1 << 27, n_variants=256, err=0.10):
    """Config 5 style batch built in HBM: N(20 kb, 2 kb) lengths, iid ACGT (+0.1 % N), quality
    clamp(round(N(35, 6)), 2, 50) + 33, and a FASTA of `n_adapters` random 30-45-mers (names ad00..): 30 % of the
    reads carry one adapter (10 % errors) at an end, 1 % in the middle (decorations overwrite bases).
    Returns (seq, qual, off int64, max_len, adapters) -- adapters in the order trimByMultiSequences visits them
    (sorted by FASTA header = index order)."""
    import torch

    rng = np.random.default_rng(seed)
    ads = ["".join("ACGT"[i] for i in rng.integers(0, 4, int(rng.integers(30, 46)))) for _ in range(n_adapters)]
    lens = np.maximum(300, np.rint(rng.normal(mean_len, sd_len, n_reads))).astype(np.int64)
    off = np.zeros(n_reads + 1, dtype=np.int64)
    off[1:] = np.cumsum(lens)
    total = int(off[-1])
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    seq = torch.empty(total, dtype=torch.uint8, device=device)
    qual = torch.empty(total, dtype=torch.uint8, device=device)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    for a in range(0, total, chunk):
        b = min(total, a + chunk)
        idx = torch.randint(0, 4, (b - a,), generator=g, device=device, dtype=torch.int64)
        s_ = lut[idx]
        del idx
        s_[torch.rand(b - a, generator=g, device=device) < 0.001] = ord("N")
        seq[a:b] = s_
        del s_
        q = torch.randn(b - a, generator=g, device=device) * sigma + mu
        qual[a:b] = (q.round_().clamp_(2, 50) + 33).to(torch.uint8)
        del q
    # one pool of noisy copies per adapter, flattened: variant id = adapter * n_variants + copy
    per = max(1, n_variants // max(1, n_adapters // 8))
    mats, vlens = [], []
    for a in ads:
        m, vl = _variant_pool(rng, np.frombuffer(a.encode(), np.uint8), err, per)
        mats.append(m)
        vlens.append(vl)
    width = max(m.shape[1] for m in mats)
    mat = np.zeros((n_adapters * per, width), np.uint8)
    for i, m in enumerate(mats):
        mat[i * per:(i + 1) * per, :m.shape[1]] = m
    vlen = np.concatenate(vlens)
    u = rng.random(n_reads)
    which = rng.integers(0, n_adapters, n_reads) * per + rng.integers(0, per, n_reads)
    side = rng.random(n_reads) < 0.5
    pos = np.zeros(n_reads, np.int64)
    at_end = (u < 0.30) & ~side
    pos[at_end] = lens[at_end] - vlen[which[at_end]]
    mid = (u >= 0.30) & (u < 0.31)
    pos[mid] = 100 + (rng.random(int(mid.sum())) * (lens[mid] - 250)).astype(np.int64)
    ri = np.nonzero(u < 0.31)[0]
    if len(ri):
        base = torch.from_numpy(off[ri] + pos[ri]).to(device)
        var = torch.from_numpy(which[ri]).to(device)
        m = torch.from_numpy(mat).to(device)
        vl = torch.from_numpy(vlen).to(device)
        j = torch.arange(m.shape[1], device=device)
        flat = base[:, None] + j[None, :]
        mask = j[None, :] < vl[var][:, None]
        seq[flat[mask]] = m[var][mask]
    return seq, qual, torch.from_numpy(off).to(device), int(lens.max()), ads
