"""ctypes mirror of include/fastplong_amd.h (struct layouts, constants, counter-buffer layout).

Pure declarations: importing this module needs neither a GPU nor the built library.
"""
import ctypes as C

FPL_ABI_VERSION = 7
FPL_MAX_IN_FLIGHT = 3
FPL_MAX_ADAPTER_LEN = 255
FPL_END_WINDOW = 200
FPL_PATTERN_LEN = 16

FPL_PASS_FILTER = 0
FPL_FAIL_N_BASE = 12
FPL_FAIL_LENGTH = 16
FPL_FAIL_TOO_LONG = 17
FPL_FAIL_QUALITY = 20
FPL_FAIL_COMPLEXITY = 24
FPL_FILTER_RESULT_TYPES = 32

FPL_OK = 0
FPL_ERR_ARG = -1
FPL_ERR_NO_DEVICE = -2
FPL_ERR_HIP = -3
FPL_ERR_ADAPTER = -4
FPL_ERR_CAPACITY = -5
FPL_ERR_STATE = -6

FPL_TEXT_OK = 0
FPL_TEXT_IRREGULAR = 1
FPL_TEXT_TOO_MANY = 2
FPL_TEXT_CANCELLED = 3


class FplTextResult(C.Structure):
    """struct fpl_text_result (fpl_wait_text)"""

    _fields_ = [("n_reads", C.c_uint32), ("status", C.c_uint32), ("n_bases", C.c_uint64), ("bad_record", C.c_uint64),
                ("max_read_len", C.c_uint32), ("n_lines", C.c_uint32)]


# FAILED_TYPES, reference src/common.h:55-64
FAILED_TYPES = [""] * FPL_FILTER_RESULT_TYPES
FAILED_TYPES[0] = "passed"
FAILED_TYPES[4] = "failed_polyx_filter"
FAILED_TYPES[8] = "failed_bad_overlap"
FAILED_TYPES[12] = "failed_too_many_n_bases"
FAILED_TYPES[16] = "failed_too_short"
FAILED_TYPES[17] = "failed_too_long"
FAILED_TYPES[20] = "failed_quality_filter"
FAILED_TYPES[24] = "failed_low_complexity"


class FplOptions(C.Structure):
    """struct fpl_options; defaults = reference CLI defaults (src/main.cpp:27-103)."""

    _fields_ = [
        ("trim_front", C.c_int32),
        ("trim_tail", C.c_int32),
        ("cut_front", C.c_int32),
        ("cut_tail", C.c_int32),
        ("cut_front_window", C.c_int32),
        ("cut_front_quality", C.c_int32),
        ("cut_tail_window", C.c_int32),
        ("cut_tail_quality", C.c_int32),
        ("polyx", C.c_int32),
        ("polyx_min_len", C.c_int32),
        ("adapter_enabled", C.c_int32),
        ("ed_max", C.c_double),
        ("trimming_extension", C.c_int32),
        ("qual_filter", C.c_int32),
        ("qualified_qual", C.c_int32),
        ("unqualified_percent_limit", C.c_int32),
        ("n_base_limit", C.c_int32),
        ("n_base_percent_limit", C.c_int32),
        ("avg_qual_req", C.c_int32),
        ("length_filter", C.c_int32),
        ("required_length", C.c_int32),
        ("max_length", C.c_int32),
        ("complexity_filter", C.c_int32),
        ("complexity_percent", C.c_int32),
        ("break_enabled", C.c_int32),
        ("break_window", C.c_int32),
        ("break_quality", C.c_int32),
        ("mask_enabled", C.c_int32),
        ("mask_window", C.c_int32),
        ("mask_quality", C.c_int32),
    ]

    @classmethod
    def default(cls, **kw):
        o = cls(
            trim_front=0, trim_tail=0, cut_front=0, cut_tail=0,
            cut_front_window=4, cut_front_quality=20, cut_tail_window=4, cut_tail_quality=20,
            polyx=0, polyx_min_len=10, adapter_enabled=1, ed_max=0.25, trimming_extension=10,
            qual_filter=1, qualified_qual=ord("0"), unqualified_percent_limit=40,
            n_base_limit=1000000, n_base_percent_limit=10, avg_qual_req=0,
            length_filter=1, required_length=20, max_length=0,
            complexity_filter=0, complexity_percent=30,
            break_enabled=0, break_window=100, break_quality=10, mask_enabled=0, mask_window=50, mask_quality=10,
        )
        for k, v in kw.items():
            if not hasattr(o, k):
                raise AttributeError(k)
            setattr(o, k, v)
        return o

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class FplAdapter(C.Structure):
    _fields_ = [("seq", C.c_char_p), ("len", C.c_int32)]


class FplAdapterPick(C.Structure):
    """struct fpl_adapter_pick: the adapter auto-detection's verdict for one read end (fpl_pick_adapter)."""

    _fields_ = [("key", C.c_int32), ("count", C.c_uint32), ("total_key", C.c_uint32), ("len", C.c_int32), ("total", C.c_uint64),
                ("seq", C.c_char * 72)]


class FplReadResult(C.Structure):
    """struct fpl_read_result, 36 bytes."""

    _fields_ = [
        ("r1_start", C.c_uint32),
        ("r1_len", C.c_uint32),
        ("frag_start", C.c_uint32 * 2),
        ("frag_len", C.c_uint32 * 2),
        ("n_frag", C.c_uint8),
        ("dropped", C.c_uint8),
        ("code", C.c_uint8 * 2),
        ("kind", C.c_uint8 * 2),
        ("median_q_pre", C.c_uint8),
        ("median_q_post", C.c_uint8 * 2),
        ("reserved", C.c_uint8 * 3),
    ]


assert C.sizeof(FplReadResult) == 36

# numpy view of the same record
RESULT_DTYPE = [
    ("r1_start", "<u4"), ("r1_len", "<u4"), ("frag_start", "<u4", (2,)), ("frag_len", "<u4", (2,)),
    ("n_frag", "u1"), ("dropped", "u1"), ("code", "u1", (2,)), ("kind", "u1", (2,)),
    ("median_q_pre", "u1"), ("median_q_post", "u1", (2,)), ("reserved", "u1", (3,)),
]

# struct fpl_fragment (32 bytes) / fpl_region (8 bytes): the --break / --mask outcome list
FRAGMENT_DTYPE = [
    ("read", "<u4"), ("seq_no", "<u4"), ("start", "<u4"), ("len", "<u4"), ("region_first", "<u4"),
    ("region_count", "<u4"), ("break_no", "<u2"), ("code", "u1"), ("kind", "u1"), ("median_q", "u1"),
    ("reserved", "u1", (3,)),
]
REGION_DTYPE = [("start", "<u4"), ("len", "<u4")]

# ---- flat int64 counter layout (see the header) -------------------------------------------
FPL_CYC_STRIDE = 32
FPL_STATS_TAIL = 128 * 3 + 1024 + 2
FPL_FR_LEN = 42
FPL_FR_FILTER = 0
FPL_FR_ADAPTER_READS = 32
FPL_FR_ADAPTER_BASES = 33
FPL_FR_POLYX_READS = 34
FPL_FR_POLYX_BASES = 38
FPL_KEY_STRIDE = FPL_MAX_ADAPTER_LEN + 1


def stats_len(c):
    return c * FPL_CYC_STRIDE + FPL_STATS_TAIL


def keyhist_len(nad):
    return nad * 2 * FPL_KEY_STRIDE


def counters_len(c, nad):
    return 2 * stats_len(c) + FPL_FR_LEN + keyhist_len(nad)


def off_pre(c):
    return 0


def off_post(c):
    return stats_len(c)


def off_fr(c):
    return 2 * stats_len(c)


def off_keyhist(c):
    return 2 * stats_len(c) + FPL_FR_LEN


class StatsView:
    """Named numpy views into one Stats block of the counter buffer."""

    def __init__(self, block, c):
        self.C = c
        self.cyc = block[: c * 32].reshape(c, 4, 8)  # [cycle][kind][cls]
        t = block[c * 32:]
        self.base_qual_hist = t[0:128]
        self.median_hist = t[128:256]
        self.median_bases = t[256:384]
        self.kmer = t[384:384 + 1024]
        self.reads = t[384 + 1024]
        self.length_sum = t[384 + 1025]


class CountersView:
    def __init__(self, buf, c, nad):
        assert buf.shape[0] == counters_len(c, nad), (buf.shape, c, nad)
        self.C, self.nad = c, nad
        self.pre = StatsView(buf[off_pre(c): off_post(c)], c)
        self.post = StatsView(buf[off_post(c): off_fr(c)], c)
        fr = buf[off_fr(c): off_keyhist(c)]
        self.filter = fr[0:32]
        self.adapter_reads = fr[32]
        self.adapter_bases = fr[33]
        self.polyx_reads = fr[34:38]
        self.polyx_bases = fr[38:42]
        self.key_hist = buf[off_keyhist(c):].reshape(nad, 2, FPL_KEY_STRIDE)


def regrid_counters(buf, c_old, c_new, nad):
    """Re-lay a counter buffer out for a different cycle capacity (cycle-major => append)."""
    import numpy as np

    out = np.zeros(counters_len(c_new, nad), dtype=np.int64)
    c = min(c_old, c_new)
    for k in range(2):
        o_old, o_new = k * stats_len(c_old), k * stats_len(c_new)
        if c_new < c_old and buf[o_old + c * 32: o_old + c_old * 32].any():
            raise ValueError("shrinking would drop non-zero cycles")
        out[o_new: o_new + c * 32] = buf[o_old: o_old + c * 32]
        out[o_new + c_new * 32: o_new + stats_len(c_new)] = buf[o_old + c_old * 32: o_old + stats_len(c_old)]
    out[off_fr(c_new):] = buf[off_fr(c_old):]
    return out
