"""TEST INFRASTRUCTURE ONLY -- builds and drives tests/emu/libfpl_emu.so: the product kernels
compiled for the host on a lock-step thread emulator (hip_emu.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

from fastplong_amd import abi

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "libfpl_emu.so")
SRCS = [os.path.join(HERE, "emu_driver.cpp"), os.path.join(HERE, "hip_emu.h")] + [
    os.path.join(ROOT, "fastplong_amd", "csrc", f) for f in ("kernels.h", "pipeline.h", "dev_prims.h", "dev_types.h")
] + [os.path.join(ROOT, "include", "fastplong_amd.h")]


def build():
    if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in SRCS):
        # (into a file of this process's own, then renamed: several pytest-xdist workers may find the library stale at once, and
        # none of them must load another one's half-written file)
        tmp = "%s.tmp.%d" % (LIB, os.getpid())
        subprocess.check_call(["g++", "-std=c++20", "-O1", "-g", "-fPIC", "-shared", "-pthread", "-I" + HERE,
                               "-o", tmp, SRCS[0]])
        os.replace(tmp, LIB)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        L.emu_process_batch.restype = C.c_int
        L.emu_process_batch.argtypes = [C.POINTER(abi.FplOptions), C.c_char_p, C.c_int, C.c_char_p, C.c_int,
                                        C.POINTER(abi.FplAdapter), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.emu_lev_bp64.restype = C.c_int
        L.emu_lev_bp64.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.emu_lev_wave.restype = C.c_int
        L.emu_lev_wave.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int]
        for f in (L.emu_lev_bp32_start, L.emu_lev_bp32_end):
            f.restype = C.c_int
            f.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        _lib = L
    return _lib


def process_batch(cfg, seq, qual, off, max_cycles, n_cu=2, with_fragments=False):
    """cfg: oracle.Config-like object with .opt, .start, .end, .fasta (bytes)."""
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    qual = np.ascontiguousarray(qual, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    n = len(off) - 1
    # at least one addressable byte so the pointers are never NULL
    if seq.size == 0:
        seq = np.zeros(1, np.uint8)
        qual = np.zeros(1, np.uint8)
    nad = 2 + len(cfg.fasta)
    counters = np.zeros(abi.counters_len(max_cycles, nad), dtype=np.int64)
    res = np.zeros(max(n, 1), dtype=abi.RESULT_DTYPE)
    arr = (abi.FplAdapter * max(1, len(cfg.fasta)))()
    for i, a in enumerate(cfg.fasta):
        arr[i].seq, arr[i].len = a, len(a)
    rc = lib().emu_process_batch(C.byref(cfg.opt), cfg.start, len(cfg.start), cfg.end, len(cfg.end), arr,
                                 len(cfg.fasta), seq.ctypes.data, qual.ctypes.data, off.ctypes.data, n,
                                 counters.ctypes.data, max_cycles, res.ctypes.data, n_cu)
    if rc != 0:
        raise RuntimeError("emu_process_batch rc=%d" % rc)
    if with_fragments:
        L = lib()
        L.emu_fragment_count.restype = C.c_uint32
        L.emu_region_count.restype = C.c_uint32
        nf, nr = L.emu_fragment_count(), L.emu_region_count()
        frags = np.zeros(max(nf, 1), dtype=abi.FRAGMENT_DTYPE)
        regs = np.zeros(max(nr, 1), dtype=abi.REGION_DTYPE)
        L.emu_get_fragments(C.c_void_p(frags.ctypes.data), C.c_void_p(regs.ctypes.data))
        return res[:n], counters, frags[:nf], regs[:nr]
    return res[:n], counters
