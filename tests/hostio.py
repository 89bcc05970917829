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
