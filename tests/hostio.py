"""Test helpers: FASTQ text <-> CSR batches and the expected output text of a run, composed in
Python from result records (independent of the C++ formatter it checks)."""
import numpy as np

from fastplong_amd import abi

PREFIX = {0: b"", 1: b"split-by-adapter-left-", 2: b"split-by-adapter-right-"}


def make_fastq(seq, qual, off, rng=None, crlf=False, strand_names=False):
    """-> (fastq bytes, names, strands)"""
    names, strands, parts = [], [], []
    nl = b"\r\n" if crlf else b"\n"
    for i in range(len(off) - 1):
        a, b = int(off[i]), int(off[i + 1])
        name = b"@read%d runid=abc ch=%d" % (i, i % 512)
        strand = (b"+" + name[1:]) if (strand_names and i % 3 == 0) else b"+"
        names.append(name)
        strands.append(strand)
        parts += [name, nl, seq[a:b].tobytes(), nl, strand, nl, qual[a:b].tobytes(), nl]
    return b"".join(parts), names, strands


def expected_outputs(seq, qual, off, names, strands, res, with_failed=True):
    """what src/seprocessor.cpp:265-281 writes: (out.fq bytes, failed.fq bytes)"""
    out, failed = [], []
    for i in range(len(off) - 1):
        r = res[i]
        if r["dropped"]:
            continue
        a = int(off[i])
        s, q = seq[a:int(off[i + 1])].tobytes(), qual[a:int(off[i + 1])].tobytes()
        for f in range(int(r["n_frag"])):
            fa, fl = int(r["frag_start"][f]), int(r["frag_len"][f])
            if r["code"][f] == abi.FPL_PASS_FILTER:
                name = names[i][:1] + PREFIX[int(r["kind"][f])] + names[i][1:]
                out += [name, b"\n", s[fa:fa + fl], b"\n", strands[i], b"\n", q[fa:fa + fl], b"\n"]
            elif with_failed and r["n_frag"] == 1:
                ra, rl = int(r["r1_start"]), int(r["r1_len"])
                failed += [names[i], b" ", abi.FAILED_TYPES[int(r["code"][f])].encode(), b"\n", s[ra:ra + rl], b"\n",
                           strands[i], b"\n", q[ra:ra + rl], b"\n"]
    return b"".join(out), b"".join(failed)


def expected_outputs_fragments(seq, qual, off, names, strands, res, frags, regs, with_failed=True):
    """the same for a --break / --mask run (src/seprocessor.cpp:234-281): output reads from the fragment list,
    names through breakByGap's and breakByRegions' insert(1, ..), bases with the N of maskRegionWithN"""
    out, failed = [], []
    for i in range(len(off) - 1):
        r = res[i]
        if r["dropped"]:
            continue
        a = int(off[i])
        s, q = seq[a:int(off[i + 1])].tobytes(), qual[a:int(off[i + 1])].tobytes()
        mine = frags[frags["read"] == i]

        def masked(start, length, fr, apply):
            sb = bytearray(s[start:start + length])
            if apply:
                for g in regs[fr["region_first"]:fr["region_first"] + fr["region_count"]]:
                    x = int(g["start"]) - start
                    sb[x:x + int(g["len"])] = b"N" * int(g["len"])
            return bytes(sb)

        for fr in mine:
            fa, fl = int(fr["start"]), int(fr["len"])
            if fr["code"] == abi.FPL_PASS_FILTER:
                name = names[i][:1] + (b"r%d-" % fr["break_no"] if fr["break_no"] else b"") + PREFIX[int(fr["kind"])] + names[i][1:]
                out += [name, b"\n", masked(fa, fl, fr, True), b"\n", strands[i], b"\n", q[fa:fa + fl], b"\n"]
            elif with_failed and len(mine) == 1:
                ra, rl = int(r["r1_start"]), int(r["r1_len"])
                in_place = fr["kind"] == 0 and fr["break_no"] == 0
                failed += [names[i], b" ", abi.FAILED_TYPES[int(fr["code"])].encode(), b"\n", masked(ra, rl, fr, in_place), b"\n",
                           strands[i], b"\n", q[ra:ra + rl], b"\n"]
    return b"".join(out), b"".join(failed)


def expected_split(texts, passed, out, workers, by_lines, number, size, digits=4):
    """--split / --split_by_lines: {file name: bytes}.  texts[i] / passed[i] = what read i contributes to the output /
    whether any of its output reads passed.  Packs of 16 reads go round-robin to `workers` workers
    (src/seprocessor.cpp:343-378); each worker appends to its current file and moves on by `workers` file numbers
    when the file has taken `size` reads (src/threadconfig.cpp:72-120).  A worker out of files keeps its last one
    (the reference would stop it and drop its queued packs -- a race; see fastplong_amd/host/cli.cpp)."""
    import os
    d, base = os.path.split(out)

    def name(k):
        num = str(k + 1)
        return os.path.join(d, num.rjust(digits, "0") + "." + base)
    files = {}
    working, current = list(range(workers)), [0] * workers
    for t in range(workers):
        files[name(t)] = b""
    n = len(texts)
    for p in range((n + 15) // 16):
        t = p % workers
        lo, hi = p * 16, min(n, p * 16 + 16)
        files[name(working[t])] += b"".join(texts[lo:hi])
        current[t] += sum(passed[lo:hi]) if by_lines else hi - lo
        if current[t] >= size and (by_lines or working[t] + workers < number):
            working[t] += workers
            files[name(working[t])] = b""
            current[t] = 0
    if not by_lines:
        for t in range(workers):
            while working[t] + workers < number:
                working[t] += workers
                files[name(working[t])] = b""
    return files
