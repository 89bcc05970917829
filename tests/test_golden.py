"""Committed golden fixtures (tests/golden/*, produced by tests/golden/make_golden.py from the
oracle + the real reference report writer).  CPU: the oracle and the host formatter reproduce
them.  GPU (-m gpu): the `fastplong_amd` CLI -- FASTQ in, HIP path through the C-ABI, FASTQ +
fastplong.json + fastplong.html out -- reproduces them byte for byte (reports modulo the `command` line and the
HTML time stamps)."""
import gzip
import json
import os
import re
import subprocess

import numpy as np
import pytest

from fastplong_amd import abi, build, synth
from tests import hostio, refjson

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
CASES = sorted(d for d in os.listdir(GOLD) if os.path.isdir(os.path.join(GOLD, d)))

OPTS = {
    "c1_qualfilter": (dict(adapter_enabled=0), "auto", "auto"),
    "c3_full": (dict(cut_front=1, cut_tail=1, cut_front_window=5, cut_tail_window=5, polyx=1, complexity_filter=1),
                synth.START_ADAPTER, synth.END_ADAPTER),
    "c5_fasta": (dict(ed_max=0.3, trimming_extension=5, required_length=30, n_base_percent_limit=5, avg_qual_req=12),
                 synth.START_ADAPTER, synth.revcomp(synth.START_ADAPTER)),
    "c3_break_mask": (dict(cut_front=1, cut_tail=1, cut_front_window=5, cut_tail_window=5, polyx=1, complexity_filter=1,
                           break_enabled=1, break_window=40, break_quality=12, mask_enabled=1, mask_window=15,
                           mask_quality=14, n_base_percent_limit=95, unqualified_percent_limit=90, complexity_percent=5),
                      synth.START_ADAPTER, synth.END_ADAPTER),
}


def gz(path):
    with gzip.open(path, "rb") as f:
        return f.read()


def parse_fastq(text):
    lines = text.split(b"\n")
    names, strands, seqs, quals = [], [], [], []
    for i in range(0, len(lines) - 1, 4):
        names.append(lines[i])
        seqs.append(np.frombuffer(lines[i + 1], np.uint8))
        strands.append(lines[i + 2])
        quals.append(np.frombuffer(lines[i + 3], np.uint8))
    seq, qual, off = synth.pack(list(zip(seqs, quals)))
    return seq, qual, off, names, strands


def fasta_list(case):
    p = os.path.join(GOLD, case, "ADAPTERS.fa")
    if not os.path.exists(p):
        return []
    recs, name = {}, None
    for line in open(p):
        line = line.rstrip("\n")
        if line.startswith(">"):
            name = line[1:]
            recs[name] = ""
        else:
            recs[name] += line
    return [recs[k].upper() for k in sorted(recs) if len(recs[k]) >= 6]


@pytest.mark.parametrize("case", CASES)
def test_oracle_reproduces_golden(orc, case):
    okw, start, end = OPTS[case]
    seq, qual, off, names, strands = parse_fastq(gz(os.path.join(GOLD, case, "in.fq.gz")))
    cfg = orc.Config(abi.FplOptions.default(**okw), start, end, fasta_list(case))
    if cfg.opt.break_enabled or cfg.opt.mask_enabled:
        res, counters, frags, regs = orc.process_batch_ex(cfg, seq, qual, off)
        out, failed = hostio.expected_outputs_fragments(seq, qual, off, names, strands, res, frags, regs)
        assert (frags["break_no"] > 0).any() and ((frags["region_count"] > 0) & (frags["code"] == 0)).any() and len(failed)
    else:
        res, counters = orc.process_batch(cfg, seq, qual, off)
        out, failed = hostio.expected_outputs(seq, qual, off, names, strands, res)
    assert out == gz(os.path.join(GOLD, case, "expected.out.fq.gz"))
    assert failed == gz(os.path.join(GOLD, case, "expected.failed.fq.gz"))
    meta = json.load(open(os.path.join(GOLD, case, "case.json")))
    C = int(np.diff(off.astype(np.int64)).max())
    assert int(abi.CountersView(counters, C, cfg.n_adapters).post.reads) == meta["fragments_passing"]
    # the JSON fixture agrees with the oracle's counters on the headline numbers
    js = json.loads(gz(os.path.join(GOLD, case, "expected.json.gz")).replace(b"},\n}", b"}\n}"))
    v = abi.CountersView(counters, C, cfg.n_adapters)
    assert js["summary"]["before_filtering"]["total_reads"] == int(v.pre.reads) == meta["reads"]
    assert js["summary"]["after_filtering"]["total_reads"] == int(v.post.reads)
    assert js["filtering_result"]["passed_filter_reads"] == int(v.filter[0])


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
# 37 reads per batch: several batches, counters accumulate (sequential reader); "chunks": the chunk-parallel reader with
# chunks of 30 kB -- every cut falls inside some record -- and batches two deep in the copy / kernel stage
# "device_parse": the same chunks, but only LOADED by the host -- the device finds the records (fpl_process_text_async)
@pytest.mark.parametrize("batch_reads", ["0", "37", "chunks", "device_parse"])
def test_cli_reproduces_golden_on_gpu(tmp_path, case, batch_reads):
    build.build_all()
    meta = json.load(open(os.path.join(GOLD, case, "case.json")))
    inp = tmp_path / "in.fq"
    inp.write_bytes(gz(os.path.join(GOLD, case, "in.fq.gz")))
    flags = [f if f != "ADAPTERS.fa" else os.path.join(GOLD, case, "ADAPTERS.fa") for f in meta["flags"]]
    cmd = [build.CLI, "-i", str(inp), "-o", str(tmp_path / "out.fq"), "--failed_out", str(tmp_path / "failed.fq"),
           "-j", str(tmp_path / "out.json"), "-h", str(tmp_path / "out.html")] + flags
    env = dict(os.environ)
    if batch_reads in ("chunks", "device_parse"):
        # (the device parses wherever the input is cut into chunks: the default; --host_parse keeps the host's parsers)
        cmd += ["--reader_threads", "3", "-V"] + ([] if batch_reads == "device_parse" else ["--host_parse"])
        env["FPLH_CHUNK_BYTES"] = "30000"
    else:
        cmd += ["--batch_reads", batch_reads]
        if batch_reads != "0":
            cmd += ["--reads_to_process", str(meta["reads"])]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert (tmp_path / "out.fq").read_bytes() == gz(os.path.join(GOLD, case, "expected.out.fq.gz"))
    assert (tmp_path / "failed.fq").read_bytes() == gz(os.path.join(GOLD, case, "expected.failed.fq.gz"))
    got = [l for l in (tmp_path / "out.json").read_bytes().split(b"\n") if not l.startswith(b'\t"command":')]
    want = gz(os.path.join(GOLD, case, "expected.json.gz")).split(b"\n")
    assert got == want
    # the HTML report against the real HtmlReporter's page (default -w 3; time stamps and the command line masked)
    page = refjson.STAMP.sub(b"<time>", (tmp_path / "out.html").read_bytes())
    page = re.sub(rb"<div id='footer'> <p>.*?</p>", b"<div id='footer'> <p></p>", page, flags=re.S)
    assert page == gz(os.path.join(GOLD, case, "expected.html.gz"))
    assert b"reads passed filter: " in p.stderr and b"HTML report: " in p.stderr
    if batch_reads == "chunks":
        assert b"chunk parsers" in p.stderr  # (the chunk-parallel reader really ran)
    if batch_reads == "device_parse":
        if {"--break", "--mask", "-b", "-N"} & set(meta["flags"]):
            assert b"device parse:" not in p.stderr  # (--break / --mask: the host's parsers, silently -- the flag was not given)
        else:
            m = re.search(rb"device parse: (\d+) chunks parsed on the device, (\d+) handed back", p.stderr)
            assert m and int(m.group(1)) >= 3 and int(m.group(2)) == 0, p.stderr[-800:]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("how", ["stdout_pipe", "file_forced", "gz_input", "stdout_pipe_device_parse"])
def test_cli_gather_output_reproduces_golden_on_gpu(tmp_path, case, how):
    """plain --out without --failed_out that is not a regular file (here: --stdout into a pipe) is written as gather lists
    over the batches' own arrays (writev, nothing is formatted); FPLH_GATHER_FILES forces the same for a file.  Same bytes
    as the formatted output (the cases with --break / --mask keep the formatter: they must still agree)"""
    build.build_all()
    meta = json.load(open(os.path.join(GOLD, case, "case.json")))
    inp = tmp_path / "in.fq"
    inp.write_bytes(gz(os.path.join(GOLD, case, "in.fq.gz")))
    if how == "gz_input":  # the fixture as it is: ONE gzip member, inflated in one piece into memory and parsed in chunks there
        inp = os.path.join(GOLD, case, "in.fq.gz")
    flags = [f if f != "ADAPTERS.fa" else os.path.join(GOLD, case, "ADAPTERS.fa") for f in meta["flags"]]
    cmd = [build.CLI, "-i", str(inp), "-j", str(tmp_path / "out.json"), "-h", str(tmp_path / "out.html"), "--reader_threads", "3", "-V"] + flags
    env = dict(os.environ, FPLH_CHUNK_BYTES="30000")
    if how not in ("stdout_pipe_device_parse", "gz_input"):  # (device parse is the default -- the gather lists then point into the chunk's text; gz_input: the text inflated into memory, loaded chunk by chunk)
        cmd += ["--host_parse"]
    if how != "file_forced":
        cmd += ["--stdout"]
    else:
        cmd += ["-o", str(tmp_path / "out.fq")]
        env["FPLH_GATHER_FILES"] = "1"
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    got = p.stdout if how != "file_forced" else (tmp_path / "out.fq").read_bytes()
    assert got == gz(os.path.join(GOLD, case, "expected.out.fq.gz"))
    if how == "gz_input":
        assert b"input: gzip members inflated into memory" in p.stderr and b"chunk parsers" in p.stderr


@pytest.mark.gpu
def test_cli_counter_merge_through_rccl_on_gpu(tmp_path):
    """bin/fastplong_amd ends every run with fpl_allreduce_counters over its devices' contexts; FPL_RCCL_FORCE=1 makes the
    one-device run take the RCCL path too (a process WITHOUT torch: librccl comes from the loader's own search).  Reports
    and output must not change."""
    build.build_all()
    case = "c3_full"
    meta = json.load(open(os.path.join(GOLD, case, "case.json")))
    inp = tmp_path / "in.fq"
    inp.write_bytes(gz(os.path.join(GOLD, case, "in.fq.gz")))
    cmd = [build.CLI, "-i", str(inp), "-o", str(tmp_path / "out.fq"), "-j", str(tmp_path / "out.json"), "-h", str(tmp_path / "out.html"),
           "-V"] + list(meta["flags"])
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=dict(os.environ, FPL_RCCL_FORCE="1"))
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert b"counter merge: one all-reduce over 1 device(s), RCCL from " in p.stderr and b"librccl" in p.stderr
    assert re.search(rb"counter merge: [0-9.e-]+ s", p.stderr)  # (the communicator was made on its own thread beside the batches)
    assert (tmp_path / "out.fq").read_bytes() == gz(os.path.join(GOLD, case, "expected.out.fq.gz"))
    got = [l for l in (tmp_path / "out.json").read_bytes().split(b"\n") if not l.startswith(b'\t"command":')]
    assert got == gz(os.path.join(GOLD, case, "expected.json.gz")).split(b"\n")


@pytest.mark.gpu
def test_cli_gzip_outputs_on_gpu(tmp_path):
    """names ending in .gz are written as concatenated gzip members (one per formatted slice, deflated in parallel)"""
    case = "c3_full"
    build.build_all()
    meta = json.load(open(os.path.join(GOLD, case, "case.json")))
    cmd = [build.CLI, "-i", os.path.join(GOLD, case, "in.fq.gz"), "-o", str(tmp_path / "out.fq.gz"), "--failed_out",
           str(tmp_path / "failed.fq.gz"), "-j", str(tmp_path / "out.json"), "-h", str(tmp_path / "out.html"), "-z", "6",
           "--batch_reads", "50"] + meta["flags"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert gz(str(tmp_path / "out.fq.gz")) == gz(os.path.join(GOLD, case, "expected.out.fq.gz"))
    assert gz(str(tmp_path / "failed.fq.gz")) == gz(os.path.join(GOLD, case, "expected.failed.fq.gz"))
    assert (tmp_path / "out.fq.gz").read_bytes()[:2] == b"\x1f\x8b"


@pytest.mark.gpu
def test_cli_detects_adapters_when_left_at_auto(tmp_path):
    """-s / -e default to "auto": the host evaluator finds the adapters most reads carry and the run trims with them"""
    build.build_all()
    # (same seeded reads as tests/test_host_evaluator.py: the reference's top-key rule also looks at the bits of the
    #  k-mer COUNT, so whether a key qualifies depends on how many reads happen to carry it)
    rng = np.random.default_rng(3)
    sa = np.frombuffer(synth.START_ADAPTER.encode(), np.uint8)
    ea = np.frombuffer(synth.END_ADAPTER.encode(), np.uint8)
    reads = []
    for _ in range(400):
        body = synth._ACGT[rng.integers(0, 4, int(rng.integers(400, 900)))]
        s_ = np.concatenate([sa, body, ea, synth._ACGT[rng.integers(0, 4, 1)]]) if rng.random() < 0.8 else body
        reads.append((s_.astype(np.uint8), np.full(len(s_), 33 + 20, np.uint8)))
    seq, qual, off = synth.pack(reads)
    text, _, _ = hostio.make_fastq(seq, qual, off)
    inp = tmp_path / "in.fq"
    inp.write_bytes(text)
    p = subprocess.run([build.CLI, "-i", str(inp), "-o", str(tmp_path / "out.fq"), "-j", str(tmp_path / "out.json"), "-h",
                        str(tmp_path / "out.html")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    err = p.stderr.decode()
    assert err.count("Detected: ") == 2, err
    js = json.loads((tmp_path / "out.json").read_text().replace("},\n}", "}\n}"))
    assert js["adapter_cutting"]["adapter_trimmed_reads"] > 200
    # the 10-mer counting ran on the device (fpl_count_end_kmers); with the host's loops instead (FPLH_HOST_KMERS) the run must
    # detect the same two sequences and write the same bytes
    import os
    p2 = subprocess.run([build.CLI, "-i", str(inp), "-o", str(tmp_path / "out2.fq"), "-j", str(tmp_path / "out2.json"), "-h",
                         str(tmp_path / "out2.html")], env=dict(os.environ, FPLH_HOST_KMERS="1"),
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p2.returncode == 0, p2.stderr.decode()[-2000:]
    det = lambda t: [ln for ln in t.splitlines() if ln.startswith("Detected: ")]  # noqa: E731
    assert det(p2.stderr.decode()) == det(err) and len(det(err)) == 2
    assert (tmp_path / "out2.fq").read_bytes() == (tmp_path / "out.fq").read_bytes()
    strip = lambda t: "\n".join(ln for ln in t.splitlines() if '"command"' not in ln)  # noqa: E731
    assert strip((tmp_path / "out2.json").read_text()) == strip((tmp_path / "out.json").read_text())


def _per_read_outputs(seq, qual, off, names, strands, res):
    texts, passed = [], []
    for i in range(len(off) - 1):
        out, _ = hostio.expected_outputs(seq, qual, off[i:i + 2], names[i:i + 1], strands[i:i + 1], res[i:i + 1], with_failed=False)
        texts.append(out)
        passed.append(bool((res[i]["code"][:int(res[i]["n_frag"])] == abi.FPL_PASS_FILTER).any()))
    return texts, passed


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["number", "number_gz_one_batch", "lines"])
def test_cli_split_outputs_on_gpu(orc, tmp_path, mode):
    """--split N / --split_by_lines L: the numbered files hold what the reference's per-worker writers would put
    there (which file a read lands in follows from its input index, -w and the evaluated read count)"""
    build.build_all()
    okw, start, end = OPTS["c3_full"]
    seq, qual, off = synth.ont_like(1500, seed=21, median_len=700, max_len=2500, p_middle=0.1, p_polya=0.2)
    text, names, strands = hostio.make_fastq(seq, qual, off)
    inp = tmp_path / "in.fq"
    inp.write_bytes(text)
    cfg = orc.Config(abi.FplOptions.default(**okw), start, end)
    res, _ = orc.process_batch(cfg, seq, qual, off)
    texts, passed = _per_read_outputs(seq, qual, off, names, strands, res)
    flags = json.load(open(os.path.join(GOLD, "c3_full", "case.json")))["flags"]
    gzipped = mode == "number_gz_one_batch"
    out = str(tmp_path / ("out.fq.gz" if gzipped else "out.fq"))
    if mode == "lines":
        extra, want = ["--split_by_lines", "1000", "-w", "3", "--split_prefix_digits", "0"], \
            hostio.expected_split(texts, passed, out, 3, True, 0, 250, digits=0)
    elif gzipped:  # 7 files, -w 16 is capped at the file count; 1500 reads are below the evaluator's limits: exact count
        extra, want = ["--split", "7", "-w", "16"], hostio.expected_split(texts, passed, out, 7, False, 7, 1500 // 7)
    else:  # two workers walk through files 1,3,5,7 and 2,4,6
        extra, want = ["--split", "7", "-w", "2"], hostio.expected_split(texts, passed, out, 2, False, 7, 1500 // 7)
    cmd = [build.CLI, "-i", str(inp), "-o", out, "--failed_out", str(tmp_path / "failed.fq"), "-j", str(tmp_path / "o.json"),
           "-h", str(tmp_path / "o.html"), "--batch_reads", "0" if gzipped else "101"] + flags + extra
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    got = {str(f): (gz(str(f)) if gzipped else f.read_bytes()) for f in tmp_path.iterdir() if "out.fq" in f.name}
    assert sorted(got) == sorted(want) and len(want) >= 6
    for k in want:
        assert got[k] == want[k], k
    assert sum(len(v) > 0 for v in want.values()) >= 6
    assert not (tmp_path / "failed.fq").exists()  # the reference opens --failed_out only without --split*


@pytest.mark.gpu
def test_cli_split_rejects_what_the_reference_rejects(tmp_path):
    build.build_all()
    inp = os.path.join(GOLD, "c3_full", "in.fq.gz")
    for extra, msg in ([["--split", "1"], "should be 2 ~ 999"], [["--split_by_lines", "1001"], "multiple of 4"],
                       [["--split_by_lines", "400"], "should be >= 1000"], [["--split", "3", "--split_by_lines", "2000"], "either"],
                       [["--split", "3", "--split_prefix_digits", "11"], "should be 0 ~ 10"],
                       [["--dont_overwrite", "-j", inp], "already exists and you have set to not rewrite"]):
        p = subprocess.run([build.CLI, "-i", inp, "-o", str(tmp_path / "o.fq"), "-j", str(tmp_path / "o.json"), "-h",
                            str(tmp_path / "o.html")] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert p.returncode != 0 and msg in p.stderr.decode(), (extra, p.stderr.decode()[-300:])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["device", "host"])
def test_cli_reads_longer_than_a_chunk_on_gpu(tmp_path, mode):
    """tests/test_cli_multi_device_stub.py::long_read_case on the real library: reads of 200 kb / 95 kb / 61 kb across 30 kB chunks,
    the chunk loader's record boundaries searched through a whole read, chunks that hold nothing"""
    from tests.test_cli_multi_device_stub import long_read_case

    build.build_all()
    long_read_case(tmp_path, dict(os.environ), 1, mode)
