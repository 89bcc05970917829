"""A SECOND, independently written reading of the reference's adapter code -- test infrastructure only.

`oracle/fpl_oracle.c` restates `src/adaptertrimmer.cpp`, and that one translation unit cannot be compiled in this image
(it includes Google Highway), so the restatement is pinned by the reference's own four known-answer tests and by
nothing else the reference holds.  This file is the same text read a second time, by other means, so that a misreading
has to be made twice to go unnoticed: written from `/root/reference/src/adaptertrimmer.cpp:13-40` (findMiddleAdapters),
`:59-166` (searchAdapter), `:168-302` (the two end trims) and `src/read.cpp:62-73` (resize / trimFront) -- NOT from
`oracle/` -- in numpy whole-array form where the oracle loops (every window's mismatch count at once, the loops'
first / last / strict-minimum rules as argmin / masks over that array) and with a textbook Wagner-Fischer table
where the oracle has its own edit distance.  `tests/test_second_reading.py` compares the two on seeded cases.

Conventions of the reference kept on purpose:
  * `threshold = round(edMax * alen)` is C's round() on a double (half away from zero): libm's own, through ctypes;
  * a mismatch count is a `size_t` compared with `int`s that are never negative here;
  * `std::string::erase(0, n)` with a negative `int n` erases everything (n converts to a huge size_t).
"""
import ctypes

import numpy as np

_libm = ctypes.CDLL("libm.so.6")
_libm.round.restype = ctypes.c_double
_libm.round.argtypes = [ctypes.c_double]


def c_round(x):
    return int(_libm.round(float(x)))


def levenshtein(a, b):
    """exact global edit distance, one row of the Wagner-Fischer table at a time (edit_distance's contract,
    src/editdistance.cpp:100-126: n == 0 -> m, m == 0 -> n)"""
    a, b = bytes(a), bytes(b)
    if not a:
        return len(b)
    if not b:
        return len(a)
    bb = np.frombuffer(b, np.uint8)
    row = np.arange(len(b) + 1, dtype=np.int64)
    j = np.arange(len(b) + 1)
    for i, ch in enumerate(a, 1):
        # cell (i, j) = min(diagonal + mismatch, above + 1, left + 1); the `left` term chains along the row:
        # new[j] = min over k <= j of (cand[k] + j - k) with cand[0] = i and cand[j] = min(diagonal + mismatch, above + 1)
        cand = np.concatenate(([i], np.minimum(row[:-1] + (bb != ch), row[1:] + 1)))
        row = np.minimum.accumulate(cand - j) + j
    return int(row[-1])


def window_mismatches(seq, adapter):
    """mm[p] = #(seq[p + i] != adapter[i]) for every p with p + alen <= rlen (raw byte inequality)"""
    s = np.frombuffer(bytes(seq), np.uint8)
    a = np.frombuffer(bytes(adapter), np.uint8)
    if len(a) == 0:
        return np.zeros(len(s) + 1, np.int64)
    if len(s) < len(a):
        return np.zeros(0, np.int64)
    w = np.lib.stride_tricks.sliding_window_view(s, len(a))
    return (w != a[None, :]).sum(axis=1).astype(np.int64)


def search_adapter(seq, adapter, ed_max, search_start=0, search_len=-1, as_left=False, as_right=False):
    """src/adaptertrimmer.cpp:59-166"""
    seq, adapter = bytes(seq), bytes(adapter)
    rlen, alen = len(seq), len(adapter)
    thr = c_round(ed_max * alen)
    search_end = rlen
    if search_len > 0:
        search_end = min(rlen, search_len + search_start)
    if search_start + alen > rlen:
        return -1
    mm = window_mismatches(seq, adapter)
    pos = -1
    if as_left:
        # p = searchStart .. searchEnd - alen - 1 ascending; the first p with mm <= thr is returned at once (no edit
        # distance); otherwise `<=` keeps the LAST of the smallest counts
        ps = np.arange(search_start, max(search_start, search_end - alen))
        if len(ps):
            m = mm[ps]
            hit = np.nonzero(m <= thr)[0]
            if len(hit):
                return int(ps[hit[0]])
            pos = int(ps[len(m) - 1 - int(np.argmin(m[::-1]))])
    elif as_right and search_end > alen:
        # p = searchEnd - alen .. searchStart DESCENDING, so here searchEnd - alen IS visited; `<=` keeps the last
        # visited = the LEFTMOST of the smallest counts
        ps = np.arange(search_end - alen, search_start - 1, -1)
        if len(ps):
            m = mm[ps]
            hit = np.nonzero(m <= thr)[0]
            if len(hit):
                return int(ps[hit[0]])
            pos = int(ps[len(m) - 1 - int(np.argmin(m[::-1]))])
    else:
        # strict `<` against 99999: the FIRST of the smallest counts, and position searchEnd - alen is never visited
        ps = np.arange(search_start, max(search_start, search_end - alen))
        if len(ps):
            m = mm[ps]
            if int(m.min()) < 99999:
                pos = int(ps[int(np.argmin(m))])
    if pos < 0:
        return -1
    return pos if levenshtein(seq[pos:pos + alen], adapter) <= thr else -1


def find_middle(seq, start_ad, end_ad, ed_max, ext):
    """src/adaptertrimmer.cpp:13-40 -> (found, start, len)"""
    seq, start_ad, end_ad = bytes(seq), bytes(start_ad), bytes(end_ad)
    rlen = len(seq)
    sp = search_adapter(seq, start_ad, ed_max)
    ep = search_adapter(seq, end_ad, ed_max)
    cover = []
    if sp >= 0:
        cover.append((sp, sp + len(start_ad)))
    if ep >= 0:
        cover.append((ep, ep + len(end_ad)))
    if not cover:
        return False, -1, -1
    lo = max(0, min(c[0] for c in cover) - ext)
    hi = min(rlen, max(c[1] for c in cover) + ext)
    return True, lo, hi - lo


def _erase_front(seq, n, length):
    """Read::trimFront, src/read.cpp:69-73: len = min(length() - 1, len); erase(0, len) -- a negative len erases everything"""
    n = min(length - 1, n)
    return b"" if n < 0 else seq[n:]


def trim_start(seq, adapter, ed_max, ext):
    """src/adaptertrimmer.cpp:168-236 -> (seq after, returned value, length of the key handed to addAdapterTrimmed or 0)"""
    seq, adapter = bytes(seq), bytes(adapter)
    rlen, alen = len(seq), len(adapter)
    if rlen < 16:
        return seq, 0, 0
    plen = min(16, alen)
    mpos = search_adapter(seq, adapter, ed_max, 0, 200, False, True)
    if mpos >= 0:
        mpos = min(mpos + ext, rlen - alen)
        return _erase_front(seq, mpos + alen, rlen), mpos + alen, alen
    thr_p = c_round(ed_max * plen)
    tail = adapter[alen - plen:]
    eds = np.array([levenshtein(seq[p:p + plen], tail) for p in range(0, max(0, min(rlen - plen, 200 - plen)))], np.int64)
    ok = np.nonzero(eds <= thr_p)[0]
    if len(ok) == 0:
        return seq, 0, 0
    # the first hit sets (pos, mined); a later hit replaces it only with a strictly smaller distance: the FIRST of the minima
    pos = int(ok[int(np.argmin(eds[ok]))])
    cmplen = min(pos + plen, alen)
    ed = levenshtein(seq[pos + plen - cmplen:pos + plen], adapter[alen - cmplen:])
    if ed <= c_round(ed_max * cmplen):
        pos = min(pos + ext, rlen - alen)
        return _erase_front(seq, pos + plen, rlen), pos + plen, cmplen
    return seq, 0, 0


def _resize(seq, n):
    """Read::resize, src/read.cpp:62-67: ignored when n > length() or n < 0"""
    return seq if (n > len(seq) or n < 0) else seq[:n]


def trim_end(seq, adapter, ed_max, ext):
    """src/adaptertrimmer.cpp:238-302"""
    seq, adapter = bytes(seq), bytes(adapter)
    rlen, alen = len(seq), len(adapter)
    if rlen < 16:
        return seq, 0, 0
    plen = min(16, alen)
    ss = max(0, rlen - 200)
    mpos = search_adapter(seq, adapter, ed_max, ss, 200, True, False)
    if mpos >= 0:
        mpos = max(0, mpos - ext)
        return _resize(seq, mpos), rlen - mpos, alen
    thr_p = c_round(ed_max * plen)
    head = adapter[:plen]
    pos, mined = -1, -1
    for p in range(0, max(0, min(rlen - plen, 200 - plen))):
        ed = levenshtein(seq[rlen - plen - p:rlen - p], head)
        if ed <= thr_p:
            if pos < 0 or ed <= mined:  # ties move on to the LATER hit ...
                pos, mined = p, ed
            else:  # ... and the first worse hit ends the walk
                break
    if pos > 0:  # strictly: a hit at p == 0 never trims
        cmplen = min(pos + plen, alen)
        a0 = rlen - plen - pos
        if levenshtein(seq[a0:a0 + cmplen], adapter[:cmplen]) <= c_round(ed_max * cmplen):
            pos = min(pos + ext, rlen - plen)
            return _resize(seq, rlen - plen - pos), pos + plen, cmplen
    return seq, 0, 0


# ---- adapter auto-detection: src/evaluator.cpp:166-183 / :207-225 (the counting loops), :268-326 (getTopKey),
# ---- :328-404 (extendKeyToAdapter), :484-560 (int2seq / seq2int) -- read a second time, in whole-array form.
# ---- oracle/evaluator_oracle.c is the checker's literal loop-by-loop reading; the reference object itself cannot be
# ---- compiled here (fastqreader.h -> ISA-L), so again two readings that agree stand in for it.

KEYLEN = 10
_CODE = np.full(256, -1, np.int64)
for _ch, _v in ((ord("A"), 0), (ord("T"), 1), (ord("U"), 1), (ord("C"), 2), (ord("G"), 3)):
    _CODE[_ch] = _v


def count_end_kmers(reads, side, shift_tail):
    """-> (counts[4^10] u32, positionAcc[4^10] u64, total).  seq2int rolls the previous key and looks at the NEW base only, and
    recodes all ten bases after a -1: a window has a key exactly when its ten bases are all A/T/U/C/G, and then the key is the
    base-4 number they spell -- no state needed."""
    size = 1 << (2 * KEYLEN)
    counts = np.zeros(size, np.int64)
    acc = np.zeros(size, np.int64)
    total = 0
    w4 = 4 ** np.arange(KEYLEN - 1, -1, -1)
    for r in reads:
        s = np.frombuffer(bytes(r), np.uint8)
        length = len(s)
        last = length - KEYLEN - shift_tail  # pos <= last
        if side == 0:
            ps = np.arange(0, min(last + 1, 128)) if last >= 0 else np.zeros(0, np.int64)
        else:
            ps = np.arange(max(0, last - 128), last + 1) if last >= 0 else np.zeros(0, np.int64)
        if len(ps) == 0:
            continue
        codes = _CODE[s]
        win = np.lib.stride_tricks.sliding_window_view(codes, KEYLEN)[ps]
        ok = (win >= 0).all(axis=1)
        keys = (win[ok] * w4[None, :]).sum(axis=1)
        np.add.at(counts, keys, 1)
        np.add.at(acc, keys, ps[ok] if side == 0 else length - ps[ok])
        total += int(ok.sum())
    return counts.astype(np.uint32), acc.astype(np.uint64), total


def top_key(counts):
    """getTopKey: the first key (ascending) that holds the largest count among the keys no rule bars; -1 when every admissible
    key counts zero (`val > topCount` against 0)"""
    k = np.nonzero(counts)[0].astype(np.int64)  # (a key that counts zero never beats topCount = 0)
    val = counts[k].astype(np.int64)
    digits = (k[:, None] >> (2 * np.arange(KEYLEN))[None, :]) & 3
    atcg = np.stack([(digits == b).sum(axis=1) for b in range(4)], axis=1)
    low = (atcg >= KEYLEN - 4).any(axis=1) | ((atcg == 0).sum(axis=1) >= 2)
    low |= (k >> KEYLEN) == (k & ((1 << KEYLEN) - 1))
    # the `diff` test reads base pairs out of the COUNT (val), not the key -- :293-299, as written
    sh = 2 * (KEYLEN - np.arange(KEYLEN - 1))
    cur = (val[:, None] >> sh[None, :]) & 3
    lastb = (val[:, None] >> (sh[None, :] - 2)) & 3
    diff = (cur != lastb).sum(axis=1)
    ok = (diff >= 3) & ~low & (atcg[:, 2] + atcg[:, 3] < KEYLEN - 2) & ((k >> 12) != 0xff) & (k != 0) & (val > 0)
    if not ok.any():
        return -1
    best = val[ok].max()
    return int(k[ok & (val == best)][0])


def int2seq(val, n, is_rna=False):
    letters = "AUCG" if is_rna else "ATCG"
    return "".join(letters[(val >> (2 * (n - 1 - i))) & 3] for i in range(n))


def extend_key(key, counts, acc, is_rna=False, left_first=True):
    """extendKeyToAdapter: alternate sides until both have stopped; a side grows by the first base b (A, T, C, G order) whose key
    has >= 70 % of the four successors' counts, >= 50 % of the SEED's count and a mean position within [-4, 2] of its predecessor's"""
    letters = "AUCG" if is_rna else "ATCG"
    mask = (1 << (2 * KEYLEN)) - 1
    adapter = int2seq(key, KEYLEN, is_rna)
    counts = counts.astype(np.int64)
    accf = acc.astype(np.float64)  # (double)positionAcc[...]
    done = {True: False, False: False}
    left = bool(left_first)
    while True:
        cur = key
        while len(adapter) < 64:
            succ = [((b << (2 * (KEYLEN - 1))) | (cur >> 2)) if left else (b | (mask & (cur << 2))) for b in range(4)]
            total = int(sum(int(counts[s]) for s in succ))
            grown = False
            for b, s in enumerate(succ):
                c = int(counts[s])
                if c == 0:
                    continue
                with np.errstate(divide="ignore", invalid="ignore"):
                    offset = float(np.float64(accf[s]) / np.float64(c) - np.float64(accf[cur]) / np.float64(counts[cur]))
                if c / total < 0.7 or c / int(counts[key]) < 0.5:
                    continue
                if offset > 2 or offset < -4:
                    continue
                cur, grown = s, True
                adapter = letters[b] + adapter if left else adapter + letters[b]
                break
            if not grown:
                done[left] = True
                break
            if len(adapter) == 64:
                done[True] = done[False] = True
                break
        left = not left
        if done[True] and done[False]:
            break
    return adapter
