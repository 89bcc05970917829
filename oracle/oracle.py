"""TEST INFRASTRUCTURE ONLY -- ctypes front end of oracle/liboracle.so (the C restatement of
the reference hot path) and a line-protocol client for oracle/_ref/ref_harness (real reference
objects).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from fastplong_amd import abi

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
REF_HARNESS = os.path.join(HERE, "_ref", "ref_harness")


class OrcRead(C.Structure):
    _fields_ = [("seq", C.c_char_p), ("qual", C.c_char_p), ("start", C.c_int), ("len", C.c_int)]


class OrcConfig(C.Structure):
    _fields_ = [
        ("opt", abi.FplOptions),
        ("start_adapter", C.c_char_p), ("start_len", C.c_int),
        ("end_adapter", C.c_char_p), ("end_len", C.c_int),
        ("fasta", C.POINTER(abi.FplAdapter)), ("n_fasta", C.c_int),
    ]


def build(force=False):
    """Compile the C restatement (and, when /root/reference is present, the real-reference
    harness).  Building the checker is not using it."""
    if force:
        subprocess.check_call(["make", "-s", "-C", HERE, "clean"])
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])  # make tracks the header dependencies
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.orc_edit_distance.restype = C.c_uint
        L.orc_edit_distance.argtypes = [C.c_char_p, C.c_uint, C.c_char_p, C.c_uint]
        L.orc_search_adapter.restype = C.c_int
        L.orc_search_adapter.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_double,
                                         C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_trim_and_cut.restype = C.c_int
        L.orc_trim_and_cut.argtypes = [C.POINTER(OrcRead), C.POINTER(abi.FplOptions), C.POINTER(C.c_int)]
        L.orc_trim_polyx.restype = C.c_int
        L.orc_trim_polyx.argtypes = [C.POINTER(OrcRead), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        for f in (L.orc_trim_start, L.orc_trim_end):
            f.restype = C.c_int
            f.argtypes = [C.POINTER(OrcRead), C.c_char_p, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_int)]
        L.orc_find_middle.restype = C.c_int
        L.orc_find_middle.argtypes = [C.POINTER(OrcRead), C.c_char_p, C.c_int, C.c_char_p, C.c_int,
                                      C.c_double, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_pass_filter.restype = C.c_int
        L.orc_pass_filter.argtypes = [C.POINTER(OrcRead), C.POINTER(abi.FplOptions)]
        L.orc_stat_read.restype = None
        L.orc_stat_read.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(OrcRead), C.POINTER(C.c_uint8)]
        L.orc_process_batch.restype = None
        L.orc_process_batch.argtypes = [C.POINTER(OrcConfig), C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_process_batch_ex.restype = None
        L.orc_process_batch_ex.argtypes = [C.POINTER(OrcConfig), C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_detect_low_quality_regions.restype = C.c_int
        L.orc_detect_low_quality_regions.argtypes = [C.POINTER(OrcRead), C.c_int, C.c_int, C.POINTER(C.c_int),
                                                     C.POINTER(C.c_int), C.c_int]
        L.orc_fraglist_free.restype = None
        L.orc_fraglist_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _b(s):
    return s.encode("latin-1") if isinstance(s, str) else bytes(s)


def _read(seq, qual=None):
    seq = _b(seq)
    qual = _b(qual) if qual is not None else b"I" * len(seq)
    r = OrcRead(seq, qual, 0, len(seq))
    r._keep = (seq, qual)
    return r


def edit_distance(a, b):
    a, b = _b(a), _b(b)
    return lib().orc_edit_distance(a, len(a), b, len(b))


def search_adapter(seq, adapter, ed_max=0.3, start=0, length=-1, as_left=False, as_right=False):
    seq, adapter = _b(seq), _b(adapter)
    return lib().orc_search_adapter(seq, len(seq), adapter, len(adapter), ed_max, start, length,
                                    int(as_left), int(as_right))


def _window(r):
    s, q = r._keep
    return s[r.start:r.start + r.len].decode("latin-1"), q[r.start:r.start + r.len].decode("latin-1")


def trim_and_cut(seq, qual, opt):
    """-> None if dropped else (frontTrimmed, seq, qual)"""
    r = _read(seq, qual)
    ft = C.c_int(0)
    if lib().orc_trim_and_cut(C.byref(r), C.byref(opt), C.byref(ft)) != 0:
        return None
    s, q = _window(r)
    return ft.value, s, q


def trim_polyx(seq, qual=None, min_len=10):
    """-> (seq, called, poly, trimmed_len)"""
    r = _read(seq, qual)
    poly, tl = C.c_int(-1), C.c_int(0)
    called = lib().orc_trim_polyx(C.byref(r), min_len, C.byref(poly), C.byref(tl))
    return _window(r)[0], called, poly.value, tl.value


def trim_start(seq, adapter, ed_max=0.3, ext=10):
    """-> (seq_after, returned_trimmed, key_len)"""
    r = _read(seq)
    adapter = _b(adapter)
    kl = C.c_int(0)
    t = lib().orc_trim_start(C.byref(r), adapter, len(adapter), ed_max, ext, C.byref(kl))
    return _window(r)[0], t, kl.value


def trim_end(seq, adapter, ed_max=0.3, ext=10):
    r = _read(seq)
    adapter = _b(adapter)
    kl = C.c_int(0)
    t = lib().orc_trim_end(C.byref(r), adapter, len(adapter), ed_max, ext, C.byref(kl))
    return _window(r)[0], t, kl.value


def find_middle(seq, start_ad, end_ad, ed_max=0.3, ext=10):
    r = _read(seq)
    sa, ea = _b(start_ad), _b(end_ad)
    st, ln = C.c_int(-1), C.c_int(0)
    f = lib().orc_find_middle(C.byref(r), sa, len(sa), ea, len(ea), ed_max, ext, C.byref(st), C.byref(ln))
    return bool(f), st.value, ln.value


def pass_filter(seq, qual, opt):
    r = _read(seq, qual)
    return lib().orc_pass_filter(C.byref(r), C.byref(opt))


class Config:
    """The inputs of processSingleEnd other than the reads."""

    def __init__(self, opt=None, start_adapter="", end_adapter="", fasta=()):
        self.opt = opt if opt is not None else abi.FplOptions.default()
        self.start = _b(start_adapter)
        self.end = _b(end_adapter)
        self.fasta = [_b(a) for a in fasta]

    @property
    def n_adapters(self):
        return 2 + len(self.fasta)

    def adapter_list(self):
        return [self.start, self.end] + self.fasta

    def _c(self):
        cfg = OrcConfig()
        cfg.opt = self.opt
        cfg.start_adapter, cfg.start_len = self.start, len(self.start)
        cfg.end_adapter, cfg.end_len = self.end, len(self.end)
        arr = (abi.FplAdapter * max(1, len(self.fasta)))()
        for i, a in enumerate(self.fasta):
            arr[i].seq, arr[i].len = a, len(a)
        cfg.fasta, cfg.n_fasta = arr, len(self.fasta)
        cfg._keep = arr
        return cfg


def process_batch(cfg, seq, qual, off, max_cycles=None, counters=None):
    """Run the restated processSingleEnd over a CSR batch.
    seq, qual: uint8 arrays; off: uint64 [n+1].  -> (results structured array, counters int64)"""
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    qual = np.ascontiguousarray(qual, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    n = len(off) - 1
    lens = np.diff(off.astype(np.int64)) if n > 0 else np.zeros(0, np.int64)
    need = int(lens.max()) if n > 0 else 0
    if max_cycles is None:
        max_cycles = max(need, 1)
    assert max_cycles >= need
    if counters is None:
        counters = np.zeros(abi.counters_len(max_cycles, cfg.n_adapters), dtype=np.int64)
    res = np.zeros(n, dtype=abi.RESULT_DTYPE)
    c = cfg._c()
    # keep the buffers addressable even when empty
    seq_p = seq.ctypes.data if seq.size else None
    qual_p = qual.ctypes.data if qual.size else None
    lib().orc_process_batch(C.byref(c), seq_p, qual_p, off.ctypes.data, n, counters.ctypes.data,
                            max_cycles, res.ctypes.data if n else None)
    return res, counters


class OrcFragList(C.Structure):
    _fields_ = [("frag", C.c_void_p), ("n_frag", C.c_uint32), ("cap_frag", C.c_uint32), ("reg", C.c_void_p),
                ("n_reg", C.c_uint32), ("cap_reg", C.c_uint32)]


def process_batch_ex(cfg, seq, qual, off, max_cycles=None):
    """process_batch for option sets with --break / --mask:
    -> (results, counters, fragments [FRAGMENT_DTYPE, in output order], regions [REGION_DTYPE])"""
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    qual = np.ascontiguousarray(qual, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    n = len(off) - 1
    need = int(np.diff(off.astype(np.int64)).max()) if n > 0 else 0
    if max_cycles is None:
        max_cycles = max(need, 1)
    counters = np.zeros(abi.counters_len(max_cycles, cfg.n_adapters), dtype=np.int64)
    res = np.zeros(n, dtype=abi.RESULT_DTYPE)
    c = cfg._c()
    fl = OrcFragList()
    L = lib()
    L.orc_process_batch_ex(C.byref(c), seq.ctypes.data if seq.size else None, qual.ctypes.data if qual.size else None,
                           off.ctypes.data, n, counters.ctypes.data, max_cycles, res.ctypes.data if n else None,
                           C.byref(fl))
    frags = np.frombuffer(C.string_at(fl.frag, fl.n_frag * 32), dtype=abi.FRAGMENT_DTYPE).copy() if fl.n_frag else \
        np.zeros(0, dtype=abi.FRAGMENT_DTYPE)
    regs = np.frombuffer(C.string_at(fl.reg, fl.n_reg * 8), dtype=abi.REGION_DTYPE).copy() if fl.n_reg else \
        np.zeros(0, dtype=abi.REGION_DTYPE)
    L.orc_fraglist_free(C.byref(fl))
    return res, counters, frags, regs


def detect_low_quality_regions(qual, window, quality):
    """-> list of (first, last) like Filter::detectLowQualityRegions"""
    q = _b(qual)
    r = _read(b"A" * len(q), q)
    cap = len(q) // 2 + 2
    a, b = (C.c_int * cap)(), (C.c_int * cap)()
    n = lib().orc_detect_low_quality_regions(C.byref(r), window, quality, a, b, cap)
    return [(a[i], b[i]) for i in range(n)]


# ---- client of the real-reference harness ---------------------------------------------------
def have_ref():
    return os.path.exists(REF_HARNESS)


class RefHarness:
    """Feeds command lines to oracle/_ref/ref_harness and returns its raw stdout."""

    def __init__(self):
        if not have_ref():
            raise FileNotFoundError(REF_HARNESS)

    @staticmethod
    def s(x):
        x = x.decode("latin-1") if isinstance(x, (bytes, bytearray)) else x
        assert " " not in x and "\n" not in x
        return "=" + x

    def run(self, lines):
        p = subprocess.run([REF_HARNESS], input=("\n".join(lines) + "\n").encode("latin-1"),
                           stdout=subprocess.PIPE, check=True)
        return p.stdout.decode("latin-1")
