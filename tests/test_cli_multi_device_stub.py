"""The HOST side of a multi-device run on a box without GPUs: bin/fastplong_amd --gpus N against tests/stub/libfastplong_amd.so
(LD_LIBRARY_PATH), a stand-in for the C-ABI library whose "devices" compute with the oracle.  What is under test is the CLI's
scheduling -- batches dealt round-robin in input order, two in flight per device thread, the formatter stage, the in-order
writer, the counter merge behind fpl_allreduce_counters -- against the committed golden
files, which the single-GPU CLI reproduces on hardware (tests/test_golden.py)."""
import ctypes as C
import gzip
import json
import os
import re
import subprocess

import pytest

from fastplong_amd import abi, build, engine
from tests import refjson
from tests.stub import build as stub_build

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
ROOT = os.path.dirname(HERE)
CASES = sorted(d for d in os.listdir(GOLD) if os.path.isdir(os.path.join(GOLD, d)))


def gz(path):
    with gzip.open(path, "rb") as f:
        return f.read()


@pytest.fixture(scope="module")
def stub_env():
    build.build_host()
    lib = stub_build.build()
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.dirname(lib) + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    return env


def test_cli_makes_the_communicators_ahead_of_the_merge(tmp_path, stub_env):
    """--gpus N: fpl_comm_init over the N contexts (device order 0 .. N-1) is called once, on a thread of its own while the
    batches run, and the one fpl_allreduce_counters over the same contexts comes after it; a one-device run calls neither
    fpl_comm_init nor needs a communicator"""
    clog = tmp_path / "comm.log"
    p, out, _ = run_cli(tmp_path, CASES[0], dict(stub_env, FPL_STUB_COMM_LOG=str(clog)), 3)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    # (... and the kept communicators are handed back right after the merge, before any context goes)
    assert open(clog).read().splitlines() == ["comm_init 3 0 1 2", "allreduce 3 0 1 2", "comm_init 0"]
    assert b"for fpl_comm_init after the last batch" in p.stderr
    clog.unlink()
    # FPL_NO_COMM_PREINIT=1: no thread beside the batches, the merge makes its own communicators
    p, out, _ = run_cli(tmp_path, CASES[0], dict(stub_env, FPL_STUB_COMM_LOG=str(clog), FPL_NO_COMM_PREINIT="1"), 3)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert open(clog).read().splitlines() == ["allreduce 3 0 1 2"]
    clog.unlink()
    p, out, _ = run_cli(tmp_path, CASES[0], dict(stub_env, FPL_STUB_COMM_LOG=str(clog)), 1, devices=1)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert open(clog).read().splitlines() == ["allreduce 1 0"]


def run_cli(tmp_path, case, env, gpus, extra=(), chunk=30000, devices=3):
    meta = json.load(open(os.path.join(GOLD, case, "case.json")))
    inp = tmp_path / "in.fq"
    inp.write_bytes(gz(os.path.join(GOLD, case, "in.fq.gz")))
    flags = [f if f != "ADAPTERS.fa" else os.path.join(GOLD, case, "ADAPTERS.fa") for f in meta["flags"]]
    out = tmp_path / ("g%d" % gpus)
    out.mkdir(exist_ok=True)
    log = out / "stub.log"
    cmd = [build.CLI, "-i", str(inp), "-o", str(out / "out.fq"), "--failed_out", str(out / "failed.fq"), "-j", str(out / "out.json"),
           "-h", str(out / "out.html"), "--gpus", str(gpus), "--reader_threads", "3", "-V"] + flags + list(extra)
    if "--device_parse" not in extra:
        cmd.append("--host_parse")  # (these tests read the stub's log of CSR submissions: the host's parsers; the device-parse tests below take the default)
    e = dict(env, FPL_STUB_DEVICES=str(devices), FPL_STUB_LOG=str(log), FPLH_CHUNK_BYTES=str(chunk))
    if gpus == 3:
        e["FPLH_PARALLEL_WRITE"] = "1"  # (the positional writer: the same bytes)
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=e)
    return p, out, log


@pytest.mark.parametrize("case", CASES)
def test_cli_three_stub_devices_reproduce_golden(tmp_path, stub_env, case):
    p, out, log = run_cli(tmp_path, case, stub_env, 3)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert (out / "out.fq").read_bytes() == gz(os.path.join(GOLD, case, "expected.out.fq.gz"))
    assert (out / "failed.fq").read_bytes() == gz(os.path.join(GOLD, case, "expected.failed.fq.gz"))
    got = [l for l in (out / "out.json").read_bytes().split(b"\n") if not l.startswith(b'\t"command":')]
    assert got == gz(os.path.join(GOLD, case, "expected.json.gz")).split(b"\n")  # the merged counters of three contexts
    page = refjson.STAMP.sub(b"<time>", (out / "out.html").read_bytes())
    page = re.sub(rb"<div id='footer'> <p>.*?</p>", b"<div id='footer'> <p></p>", page, flags=re.S)
    assert page == gz(os.path.join(GOLD, case, "expected.html.gz"))
    # scheduling: batch k of the input went to device k mod 3, and the batches are the input cut in order
    lines = [l.split() for l in open(log).read().splitlines()]
    assert len(lines) >= 6, "the test input must make several batches per device"
    per_dev = {d: [l for l in lines if int(l[0]) == d] for d in range(3)}
    assert all(abs(len(per_dev[d]) - len(lines) / 3) <= 1 for d in range(3))
    meta = json.load(open(os.path.join(GOLD, case, "case.json")))
    assert sum(int(l[1]) for l in lines) == meta["reads"]
    # the first bases of every batch, taken in round-robin order over the devices' own submission order, walk the input
    # front to back: batch k is the k-th cut of the file
    text = gz(os.path.join(GOLD, case, "in.fq.gz")).split(b"\n")
    seqs = [text[i + 1] for i in range(0, len(text) - 1, 4)]
    k, idx = 0, [0, 0, 0]
    for b in range(len(lines)):
        d = b % 3
        dev, n, head = per_dev[d][idx[d]]
        idx[d] += 1
        cat = b"".join(seqs[k:k + int(n)])[:16]
        assert bytes.fromhex(head.decode() if isinstance(head, bytes) else head) == cat, "batch %d is not reads %d.." % (b, k)
        k += int(n)


def test_cli_device_count_is_partition_invariant_and_checked(tmp_path, stub_env):
    case = "c3_full"
    p1, out1, _ = run_cli(tmp_path, case, stub_env, 1)
    p2, out2, _ = run_cli(tmp_path, case, stub_env, 2, chunk=17000)
    assert p1.returncode == 0 and p2.returncode == 0, (p1.stderr[-500:], p2.stderr[-500:])
    for f in ("out.fq", "failed.fq"):
        assert (out1 / f).read_bytes() == (out2 / f).read_bytes()
    strip = lambda b: [l for l in b.split(b"\n") if not l.startswith(b'\t"command":')]  # noqa: E731
    assert strip((out1 / "out.json").read_bytes()) == strip((out2 / "out.json").read_bytes())
    # more devices than there are: the reference-style error, not a crash
    p4, _, _ = run_cli(tmp_path, case, stub_env, 4)
    assert p4.returncode != 0 and b"needs 4 HIP device(s)" in p4.stderr
    # gzip output goes through the same in-order writer (members concatenated at their offsets)
    p3, out3, _ = run_cli(tmp_path, case, stub_env, 3, extra=["-z", "3"])
    assert p3.returncode == 0


def test_cli_eight_stub_devices(tmp_path, stub_env):
    """the node's full width: eight device threads, sixteen batches in flight, one writer; default reader threads"""
    case = "c3_full"
    p, out, log = run_cli(tmp_path, case, stub_env, 8, chunk=12000, devices=8)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert (out / "out.fq").read_bytes() == gz(os.path.join(GOLD, case, "expected.out.fq.gz"))
    assert (out / "failed.fq").read_bytes() == gz(os.path.join(GOLD, case, "expected.failed.fq.gz"))
    got = [l for l in (out / "out.json").read_bytes().split(b"\n") if not l.startswith(b'\t"command":')]
    assert got == gz(os.path.join(GOLD, case, "expected.json.gz")).split(b"\n")
    assert sorted(set(int(l.split()[0]) for l in open(log))) == list(range(8))


def test_cli_gz_output_with_three_devices(tmp_path, stub_env):
    case = "c3_full"
    meta = json.load(open(os.path.join(GOLD, case, "case.json")))
    inp = tmp_path / "in.fq"
    inp.write_bytes(gz(os.path.join(GOLD, case, "in.fq.gz")))
    cmd = [build.CLI, "-i", str(inp), "-o", str(tmp_path / "out.fq.gz"), "--failed_out", str(tmp_path / "failed.fq.gz"),
           "-j", str(tmp_path / "o.json"), "-h", str(tmp_path / "o.html"), "--gpus", "3", "--reader_threads", "2"] + meta["flags"]
    e = dict(stub_env, FPL_STUB_DEVICES="3", FPLH_CHUNK_BYTES="25000")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=e)
    assert p.returncode == 0, p.stderr.decode()[-1500:]
    assert gz(tmp_path / "out.fq.gz") == gz(os.path.join(GOLD, case, "expected.out.fq.gz"))
    assert gz(tmp_path / "failed.fq.gz") == gz(os.path.join(GOLD, case, "expected.failed.fq.gz"))


def test_stub_agrees_with_the_library_on_defaults():
    """the stand-in restates fpl_options_default: it must be the real library's (which loads without a GPU)"""
    real = engine.load_library()
    stub = C.CDLL(stub_build.build())
    a, b = abi.FplOptions(), abi.FplOptions()
    real.fpl_options_default(C.byref(a))
    stub.fpl_options_default(C.byref(b))
    assert bytes(a) == bytes(b)
    assert stub.fpl_abi_version() == abi.FPL_ABI_VERSION


@pytest.mark.parametrize("mode", ["number_w16", "number_w2", "lines_w3", "number_gz", "one_thread"])
def test_cli_split_outputs_per_worker_writer_threads(tmp_path, stub_env, orc, mode):
    """--split / --split_by_lines with one writer thread per worker (SplitOutput::start_threads: the in-order thread only plans
    which reads go to which worker, the workers' threads gather and write): every numbered file holds the bytes the reference's
    per-worker writers put there (tests/hostio.expected_split, the replay pinned against the real ThreadConfig / Writer), whatever
    the batch cuts -- 30 kB chunks, so packs of 16 reads straddle batches -- and on two devices."""
    import numpy as np
    from fastplong_amd import synth
    from tests import hostio
    from tests.test_golden import OPTS, _per_read_outputs

    okw, start, end = OPTS["c3_full"]
    seq, qual, off = synth.ont_like(1500, seed=21, median_len=700, max_len=2500, p_middle=0.1, p_polya=0.2)
    text, names, strands = hostio.make_fastq(seq, qual, off)
    inp = tmp_path / "in.fq"
    inp.write_bytes(text)
    cfg = orc.Config(abi.FplOptions.default(**okw), start, end)
    res, _ = orc.process_batch(cfg, seq, qual, off)
    texts, passed = _per_read_outputs(seq, qual, off, names, strands, res)
    flags = json.load(open(os.path.join(GOLD, "c3_full", "case.json")))["flags"]
    gzipped = mode == "number_gz"
    out = str(tmp_path / ("out.fq.gz" if gzipped else "out.fq"))
    env = dict(stub_env, FPL_STUB_DEVICES="2", FPLH_CHUNK_BYTES="30000")
    if mode == "lines_w3":
        extra, want = ["--split_by_lines", "1000", "-w", "3", "--split_prefix_digits", "0"], \
            hostio.expected_split(texts, passed, out, 3, True, 0, 250, digits=0)
    elif mode == "number_w2" or mode == "one_thread":
        extra, want = ["--split", "7", "-w", "2"], hostio.expected_split(texts, passed, out, 2, False, 7, 1500 // 7)
        if mode == "one_thread":
            env["FPLH_SPLIT_ONE_THREAD"] = "1"
    else:  # 7 files, -w 16 is capped at the file count
        extra, want = ["--split", "7", "-w", "16"], hostio.expected_split(texts, passed, out, 7, False, 7, 1500 // 7)
    cmd = [build.CLI, "-i", str(inp), "-o", out, "-j", str(tmp_path / "o.json"), "-h", str(tmp_path / "o.html"), "--gpus", "2",
           "--reader_threads", "3"] + flags + extra
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    got = {str(f): (gz(str(f)) if gzipped else f.read_bytes()) for f in tmp_path.iterdir() if "out.fq" in f.name}
    assert sorted(got) == sorted(want) and len(want) >= 6
    for k in want:
        assert got[k] == want[k], k


def test_cli_against_the_null_device_is_the_host_side_alone(tmp_path):
    """tools/nulldev (measurement infrastructure behind bench.py's e2e host_ceiling runs): eight devices that take no time and call
    every read one passing fragment -- the CLI then writes its input back out, batch by batch over eight device threads, and the
    reports carry the totals; the library exports every symbol the header declares (the CLI links against that set)"""
    import ctypes as C
    import re
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from nulldev import build as null_build

    lib = null_build.build()
    build.build_host()
    L = C.CDLL(lib)
    for name in sorted(set(re.findall(r"\b(fpl_[a-z_0-9]+)\s*\(", open(os.path.join(ROOT, "include", "fastplong_amd.h")).read()))):
        if name not in ("fpl_process_batch_device", "fpl_counters_device_ptr", "fpl_enable_timing", "fpl_get_kernel_times", "fpl_wait_"):
            assert hasattr(L, name), name  # (the device-pointer calls and the kernel timers have no meaning without a device)
    assert L.fpl_abi_version() == abi.FPL_ABI_VERSION
    text = gz(os.path.join(GOLD, CASES[0], "in.fq.gz"))
    inp = tmp_path / "in.fq"
    inp.write_bytes(text)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.dirname(lib) + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""),
               FPL_NULL_DEVICES="8", FPLH_CHUNK_BYTES="30000")
    p = subprocess.run([build.CLI, "-i", str(inp), "-o", str(tmp_path / "out.fq"), "-s", "ACGTACGTACGTACGTACGT", "-e", "TTGCATTGCATTGCATTGCA",
                        "-j", str(tmp_path / "o.json"), "-h", str(tmp_path / "o.html"), "--gpus", "8", "--reader_threads", "3", "-V"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, env=env)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = text.split(b"\n")
    want = b"".join(lines[i] + b"\n" + lines[i + 1] + b"\n" + lines[i + 2] + b"\n" + lines[i + 3] + b"\n" for i in range(0, len(lines) - 1, 4))
    assert (tmp_path / "out.fq").read_bytes() == want
    j = json.load(open(tmp_path / "o.json"))
    assert j["summary"]["before_filtering"]["total_reads"] == (len(lines) - 1) // 4 == j["filtering_result"]["passed_filter_reads"]
    # nine devices are one too many for eight
    p = subprocess.run([build.CLI, "-i", str(inp), "-o", "/dev/null", "-s", "ACGTACGTACGTACGTACGT", "-e", "TTGCATTGCATTGCATTGCA", "--gpus", "9",
                        "-j", str(tmp_path / "o.json"), "-h", str(tmp_path / "o.html")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, env=env)
    assert p.returncode != 0


@pytest.mark.parametrize("case", CASES)
def test_cli_device_parse_reproduces_golden(tmp_path, stub_env, case):
    """--device_parse: the chunk parsers only load the file's bytes, the DEVICE finds the records (fpl_process_text_async; here
    the stub's CPU stand-in for the device) and the output is formatted out of the text -- the same bytes as the host's reader
    gives, over two devices and chunks of 30 kB (every chunk boundary falls somewhere else in a record)"""
    p, out, log = run_cli(tmp_path, case, stub_env, 2, extra=["--device_parse"])
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert (out / "out.fq").read_bytes() == gz(os.path.join(GOLD, case, "expected.out.fq.gz"))
    assert (out / "failed.fq").read_bytes() == gz(os.path.join(GOLD, case, "expected.failed.fq.gz"))
    got = [l for l in (out / "out.json").read_bytes().split(b"\n") if not l.startswith(b'\t"command":')]
    assert got == gz(os.path.join(GOLD, case, "expected.json.gz")).split(b"\n")
    meta = json.load(open(os.path.join(GOLD, case, "case.json")))
    if {"--break", "--mask", "-b", "-N"} & set(meta["flags"]):
        assert b"--device_parse does not apply" in p.stderr  # (fragment lists come back through the CSR entry points)
    else:
        m = re.search(rb"device parse: (\d+) chunks parsed on the device, (\d+) handed back", p.stderr)
        assert m and int(m.group(1)) >= 6 and int(m.group(2)) == 0, p.stderr[-800:]
        # every submission was text (the stub logs one more line per text batch: its own CSR call behind the parse)
        kinds = [l.split()[1] == "text" for l in open(log).read().splitlines()]
        assert sum(kinds) == int(m.group(1)) and len(kinds) == 2 * sum(kinds)


def _records(text):
    ls = text.split(b"\n")
    return [ls[i:i + 4] for i in range(0, len(ls) - 1, 4)]


@pytest.mark.parametrize("what", ["blank lines", "crlf and no final line break", "junk line in front of a header"])
def test_cli_device_parse_hands_irregular_chunks_to_the_host_reader(tmp_path, stub_env, what):
    """text the reference's reader treats specially (src/fastqreader.cpp:219-347: blank lines and lines without '@' are skipped
    in front of a header, "\\r\\n" and a missing last line break are fine) is refused by the device chunk by chunk and parsed by
    the host's reader: the run's output is what it is without --device_parse"""
    case = "c3_full"
    recs = _records(gz(os.path.join(GOLD, case, "in.fq.gz")))
    parts = []
    for i, r in enumerate(recs):
        nl = b"\r\n" if (what.startswith("crlf") and i % 3) else b"\n"
        if what == "blank lines" and i % 40 == 7:
            parts.append(b"\n")
        if what.startswith("junk") and i % 55 == 9:
            parts.append(b"this line is no header\n")
        parts.append(nl.join(r) + nl)
    text = b"".join(parts)
    if what.startswith("crlf"):
        text = text[:-1] if text.endswith(b"\n") and not text.endswith(b"\r\n") else text[:-2]
    meta = json.load(open(os.path.join(GOLD, case, "case.json")))
    outs = {}
    for mode in ("host", "device"):
        d = tmp_path / mode
        d.mkdir()
        inp = d / "in.fq"
        inp.write_bytes(text)
        cmd = [build.CLI, "-i", str(inp), "-o", str(d / "out.fq"), "--failed_out", str(d / "failed.fq"), "-j", str(d / "out.json"),
               "-h", str(d / "out.html"), "--gpus", "2", "--reader_threads", "3", "-V"] + meta["flags"] + (["--device_parse"] if mode == "device" else ["--host_parse"])
        e = dict(stub_env, FPL_STUB_DEVICES="2", FPLH_CHUNK_BYTES="30000")
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=e)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        outs[mode] = (p, d)
    for f in ("out.fq", "failed.fq"):
        assert (outs["host"][1] / f).read_bytes() == (outs["device"][1] / f).read_bytes(), f
    strip = lambda b: [l for l in b.split(b"\n") if not l.startswith(b'\t"command":')]  # noqa: E731
    assert strip((outs["host"][1] / "out.json").read_bytes()) == strip((outs["device"][1] / "out.json").read_bytes())
    m = re.search(rb"device parse: (\d+) chunks parsed on the device, (\d+) handed back", outs["device"][0].stderr)
    assert m and int(m.group(2)) >= 1 and int(m.group(1)) >= 1, outs["device"][0].stderr[-800:]
    if what == "blank lines":  # the reference's reader (and this host's) reads every record of such a file
        assert len(_records((outs["device"][1] / "out.fq").read_bytes())) > 0.5 * len(recs)


@pytest.mark.parametrize("gpus", [1, 3])
@pytest.mark.parametrize("where", ["first chunk", "middle", "last record"])
def test_cli_device_parse_stops_at_a_malformed_record_like_the_reference(tmp_path, stub_env, gpus, where):
    """a record the reference stops reading at (qualities and bases of different lengths, src/fastqreader.cpp:326-341): the
    input ENDS there -- message on stderr, exit code 0, reports over the reads in front of it.  With --device_parse the chunk's
    verdict comes from its device while chunks behind it are under way on other devices: every chunk's verdict is published
    (fpl_peek_text) and a chunk runs only when all chunks in front of it were good, so nothing behind the record is counted --
    output and JSON are those of the host's reader, on one device and on three"""
    case = "c3_full"
    recs = _records(gz(os.path.join(GOLD, case, "in.fq.gz")))
    bad = {"first chunk": 3, "middle": len(recs) // 2, "last record": len(recs) - 1}[where]
    recs[bad][3] = recs[bad][3][:-2]
    text = b"".join(b"\n".join(r) + b"\n" for r in recs)
    meta = json.load(open(os.path.join(GOLD, case, "case.json")))
    outs = {}
    for mode in ("host", "device"):
        d = tmp_path / mode
        d.mkdir()
        inp = d / "in.fq"
        inp.write_bytes(text)
        log = d / "stub.log"
        cmd = [build.CLI, "-i", str(inp), "-o", str(d / "out.fq"), "--failed_out", str(d / "failed.fq"), "-j", str(d / "out.json"),
               "-h", str(d / "out.html"), "--gpus", str(gpus), "--reader_threads", "3", "-V"] + meta["flags"] + (["--device_parse"] if mode == "device" else ["--host_parse"])
        e = dict(stub_env, FPL_STUB_DEVICES=str(gpus), FPLH_CHUNK_BYTES="30000", FPL_STUB_LOG=str(log))
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=e)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        assert b"sequence and quality have different length" in p.stderr
        outs[mode] = (p, d)
    for f in ("out.fq", "failed.fq"):
        assert (outs["host"][1] / f).read_bytes() == (outs["device"][1] / f).read_bytes(), f
    strip = lambda b: [l for l in b.split(b"\n") if not l.startswith(b'\t"command":')]  # noqa: E731
    assert strip((outs["host"][1] / "out.json").read_bytes()) == strip((outs["device"][1] / "out.json").read_bytes())
    js = json.loads((outs["device"][1] / "out.json").read_bytes().replace(b"},\n}", b"}\n}"))
    assert js["summary"]["before_filtering"]["total_reads"] == bad  # the reads in front of the record, no more
    if where == "first chunk" and gpus == 3:  # chunks behind the record were under way on the other devices: dropped, not run
        assert b"cancel" in open(outs["device"][1] / "stub.log", "rb").read()


@pytest.mark.parametrize("mode", ["device", "host"])
def test_cli_reads_longer_than_a_chunk(tmp_path, stub_env, mode):
    long_read_case(tmp_path, dict(stub_env, FPL_STUB_DEVICES="2"), 2, mode)


def long_read_case(tmp_path, env, gpus, mode):
    """reads that run across many 30 kB chunks (200 kb and 95 kb between short ones; quality lines that start with '@' and '+'):
    chunks in which no record starts hold nothing, the chunk a long read starts in runs on to its end -- with the device parsing
    (text-backed batches) and with the host's parsers, the output is the oracle's for the records a plain line scan finds"""
    import numpy as np
    from fastplong_amd import synth
    from oracle import oracle
    from tests import hostio

    rng = np.random.default_rng(11)
    lens = [400, 200_000, 300, 350, 95_000, 29_990, 30_010, 500, 61_000, 120, 450]
    reads = []
    for i, L in enumerate(lens):
        s = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)].copy()
        q = rng.integers(33 + 10, 33 + 40, L).astype(np.uint8)
        q[0] = ord("@") if i % 2 else ord("+")
        reads.append((s, q))
    seq, qual, off = synth.pack(reads)
    text, names, strands = hostio.make_fastq(seq, qual, off)
    inp = tmp_path / "in.fq"
    inp.write_bytes(text)
    cfg = oracle.Config(abi.FplOptions.default(cut_front=1, cut_tail=1), synth.START_ADAPTER, synth.END_ADAPTER)
    res, _ = oracle.process_batch(cfg, seq, qual, off)
    want_out, want_failed = hostio.expected_outputs(seq, qual, off, names, strands, res)
    cmd = [build.CLI, "-i", str(inp), "-o", str(tmp_path / "out.fq"), "--failed_out", str(tmp_path / "failed.fq"), "-j", str(tmp_path / "o.json"),
           "-h", str(tmp_path / "o.html"), "-s", synth.START_ADAPTER, "-e", synth.END_ADAPTER, "--cut_front", "--cut_tail", "--gpus", str(gpus),
           "--reader_threads", "3", "-V"] + (["--host_parse"] if mode == "host" else [])
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=dict(env, FPLH_CHUNK_BYTES="30000"))
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert (tmp_path / "out.fq").read_bytes() == want_out
    assert (tmp_path / "failed.fq").read_bytes() == want_failed
    js = json.loads((tmp_path / "o.json").read_bytes().replace(b"},\n}", b"}\n}"))
    assert js["summary"]["before_filtering"]["total_reads"] == len(lens)
    assert js["summary"]["before_filtering"]["total_bases"] == sum(lens)
    if mode == "device":
        m = re.search(rb"device parse: (\d+) chunks parsed on the device, (\d+) handed back", p.stderr)
        assert m and int(m.group(2)) == 0 and int(m.group(1)) >= 5, p.stderr[-600:]
