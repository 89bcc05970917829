#!/usr/bin/env python
"""Generate the committed golden fixtures (run in the build container, where /root/reference
exists).  For each case: a small seeded FASTQ, and the outputs the reference path produces for
it -- trimmed FASTQ and failed FASTQ composed from the oracle's result records (the oracle is
pinned on the reference's KATs and on real reference objects), fastplong.json written by the REAL
reference JsonReporter/Stats/FilterResult code (oracle/_ref) with the `command` line removed.

    python tests/golden/make_golden.py
"""
import gzip
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from fastplong_amd import abi, synth  # noqa: E402
from oracle import oracle  # noqa: E402
from tests import hostio, refjson  # noqa: E402

FASTA = {"ad03 third": "ACGTTGCAATGCCGTAGGCT", "ad01 first": "ttgaccagtaggcatcaggatcca", "ad02": "GATTACAGATTACA",
         "ad00 short": "ACGTA"}  # ad00 is < 6 bp and must be skipped; visited in header-sorted order, upper-cased

CASES = {
    # BASELINE.json configs[0]: quality-filter only (-A)
    "c1_qualfilter": dict(flags=["-A"], opt=dict(adapter_enabled=0), start="auto", end="auto", fasta=None, n=150, seed=41),
    # configs[2]: full pipeline
    "c3_full": dict(flags=["-s", synth.START_ADAPTER, "-e", synth.END_ADAPTER, "--cut_front", "--cut_tail", "-W", "5",
                           "-x", "-y"],
                    opt=dict(cut_front=1, cut_tail=1, cut_front_window=5, cut_tail_window=5, polyx=1, complexity_filter=1),
                    start=synth.START_ADAPTER, end=synth.END_ADAPTER, fasta=None, n=150, seed=42),
    # configs[4] in small: --adapter_fasta with -s (end adapter = reverse complement of start)
    "c5_fasta": dict(flags=["-s", synth.START_ADAPTER, "-a", "ADAPTERS.fa", "-d", "0.3", "--trimming_extension", "5",
                            "-l", "30", "-n", "5", "-m", "12"],
                     opt=dict(ed_max=0.3, trimming_extension=5, required_length=30, n_base_percent_limit=5, avg_qual_req=12),
                     start=synth.START_ADAPTER, end=synth.revcomp(synth.START_ADAPTER), fasta=FASTA, n=150, seed=43),
    # --break / --mask on top of the full pipeline (generous -n / -u so that masked reads pass)
    "c3_break_mask": dict(flags=["-s", synth.START_ADAPTER, "-e", synth.END_ADAPTER, "--cut_front", "--cut_tail", "-W", "5",
                                 "-x", "-y", "-b", "--break_window_size", "40", "--break_mean_quality", "12", "-N",
                                 "--mask_window_size", "15", "--mask_mean_quality", "14", "-n", "95", "-u", "90", "-Y", "5"],
                          opt=dict(cut_front=1, cut_tail=1, cut_front_window=5, cut_tail_window=5, polyx=1, complexity_filter=1,
                                   break_enabled=1, break_window=40, break_quality=12, mask_enabled=1, mask_window=15,
                                   mask_quality=14, n_base_percent_limit=95, unqualified_percent_limit=90,
                                   complexity_percent=5),
                          start=synth.START_ADAPTER, end=synth.END_ADAPTER, fasta=None, n=150, seed=44, lowq=True),
}


def fasta_adapters(d):
    """Options::loadFastaAdapters: header-sorted, upper-cased, >= 6 bp"""
    return [d[k].upper() for k in sorted(d) if len(d[k]) >= 6]


def main():
    oracle.build()
    ref = oracle.RefHarness()
    for name, c in CASES.items():
        fasta = fasta_adapters(c["fasta"]) if c["fasta"] else []
        half = c["n"] // 2
        a = synth.adversarial(half, seed=c["seed"], fasta=fasta)
        b = synth.ont_like(c["n"] - half, seed=c["seed"], median_len=900, max_len=2500, p_middle=0.15, p_polya=0.2)
        reads = []
        for (s, q, o) in (a, b):
            reads += [(s[int(o[i]):int(o[i + 1])], q[int(o[i]):int(o[i + 1])]) for i in range(len(o) - 1) if o[i + 1] > o[i]]
        seq, qual, off = synth.pack(reads)
        if c.get("lowq"):  # stretches far below the --break / --mask thresholds
            rng = np.random.default_rng(c["seed"])
            for i in range(len(off) - 1):
                lo, hi = int(off[i]), int(off[i + 1])
                pos = lo + int(rng.integers(0, max(1, (hi - lo) // 2)))
                while pos < hi:
                    run = int(rng.integers(10, 200))
                    if rng.random() < 0.4:
                        qual[pos:min(hi, pos + run)] = np.clip(np.round(rng.normal(6, 3, min(hi, pos + run) - pos)), 2, 40) + 33
                    pos += run + int(rng.integers(30, 600))
        seq[seq == ord("U")] = ord("R")  # a file with both U and T is rejected up front (src/evaluator.cpp:50-52)
        text, names, strands = hostio.make_fastq(seq, qual, off, strand_names=True)
        cfg = oracle.Config(abi.FplOptions.default(**c["opt"]), c["start"], c["end"], fasta)
        C = int(np.diff(off.astype(np.int64)).max())
        frags = regs = None
        if cfg.opt.break_enabled or cfg.opt.mask_enabled:
            res, counters, frags, regs = oracle.process_batch_ex(cfg, seq, qual, off, max_cycles=C)
            out, failed = hostio.expected_outputs_fragments(seq, qual, off, names, strands, res, frags, regs)
        else:
            res, counters = oracle.process_batch(cfg, seq, qual, off, max_cycles=C)
            out, failed = hostio.expected_outputs(seq, qual, off, names, strands, res)
        d = os.path.join(HERE, name)
        os.makedirs(d, exist_ok=True)
        with gzip.GzipFile(os.path.join(d, "in.fq.gz"), "wb", mtime=0) as f:
            f.write(text)
        with gzip.GzipFile(os.path.join(d, "expected.out.fq.gz"), "wb", mtime=0) as f:
            f.write(out)
        with gzip.GzipFile(os.path.join(d, "expected.failed.fq.gz"), "wb", mtime=0) as f:
            f.write(failed)
        if c["fasta"]:
            with open(os.path.join(d, "ADAPTERS.fa"), "w") as f:
                for k in c["fasta"]:  # file order differs from sorted order on purpose
                    f.write(">%s\n%s\n" % (k, c["fasta"][k]))
        tmp, tmph = os.path.join(d, "ref.json.tmp"), os.path.join(d, "ref.html.tmp")
        refjson.reference_json(ref, tmp, cfg, seq, qual, off, res, counters, C, threads=3, frags=frags, regs=regs, html=tmph)
        lines = [l for l in open(tmp, "rb").read().split(b"\n") if not l.startswith(b'\t"command":')]
        page = refjson.STAMP.sub(b"<time>", open(tmph, "rb").read())  # the real HtmlReporter's page, -w 3, empty command
        os.remove(tmp)
        os.remove(tmph)
        with gzip.GzipFile(os.path.join(d, "expected.html.gz"), "wb", mtime=0) as f:
            f.write(page)
        with gzip.GzipFile(os.path.join(d, "expected.json.gz"), "wb", mtime=0) as f:
            f.write(b"\n".join(lines))
        json.dump({"flags": c["flags"], "reads": len(off) - 1, "bases": int(off[-1]),
                   "fragments_passing": int(abi.CountersView(counters, C, cfg.n_adapters).post.reads)},
                  open(os.path.join(d, "case.json"), "w"), indent=1)
        print(name, len(off) - 1, "reads", int(off[-1]), "bases; out", len(out), "failed", len(failed))


if __name__ == "__main__":
    main()
