"""Python front end of libfastplong_amd.so (the HIP library behind include/fastplong_amd.h).

There is no fallback of any kind: if the library has not been built, or no MI355X is visible,
construction raises.  torch is used only as plumbing (device memory, streams, collectives)."""
import ctypes as C
import os

import numpy as np

from . import abi

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libfastplong_amd.so")

# A context drives five streams; the HIP runtime's default is four hardware queues per device, and two streams on one queue run in
# submission order.  The HOST asks for more, before its first HIP call (the library leaves the environment alone).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

EXPORTS = [
    "fpl_abi_version", "fpl_strerror", "fpl_last_error", "fpl_options_default", "fpl_create", "fpl_destroy",
    "fpl_process_batch_device", "fpl_process_batch", "fpl_max_cycles", "fpl_n_adapters", "fpl_counters_len",
    "fpl_reserve_cycles", "fpl_counters_device_ptr", "fpl_get_counters", "fpl_reset_counters", "fpl_synchronize",
    "fpl_enable_timing", "fpl_get_kernel_times", "fpl_fragment_counts", "fpl_get_fragments",
    "fpl_process_batch_async", "fpl_wait", "fpl_in_flight", "fpl_host_alloc", "fpl_host_free", "fpl_allreduce_counters",
    "fpl_count_end_kmers", "fpl_pick_adapter", "fpl_rccl_library", "fpl_comm_init", "fpl_get_batch_forms", "fpl_assume_inputs_ready",
    "fpl_process_text_async", "fpl_wait_text", "fpl_peek_text", "fpl_start_text", "fpl_cancel_text",
]


class FplError(RuntimeError):
    pass


_lib = None


def load_library(path=None):
    """dlopen the in-tree HIP library and declare its prototypes; raises if it is missing.
    path: another build of the same library (tools/ab_bench.py compares kernel variants side by side); not cached."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # torch first: PyTorch-ROCm carries its own libamdhip64; loading ours afterwards makes the
    # dynamic loader resolve to that same runtime, so device pointers and streams are shared.
    # (Loading the system HIP runtime before torch leaves two runtimes in one process.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib_path = path or LIB_PATH
    if not os.path.exists(lib_path):
        raise FplError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950); there is no CPU fallback" % lib_path)
    L = C.CDLL(lib_path)
    L.fpl_abi_version.restype = C.c_int
    L.fpl_strerror.restype = C.c_char_p
    L.fpl_strerror.argtypes = [C.c_int]
    L.fpl_last_error.restype = C.c_char_p
    L.fpl_last_error.argtypes = [C.c_void_p]
    L.fpl_options_default.restype = None
    L.fpl_options_default.argtypes = [C.POINTER(abi.FplOptions)]
    L.fpl_create.restype = C.c_int
    L.fpl_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(abi.FplOptions), C.c_char_p, C.c_int32, C.c_char_p,
                             C.c_int32, C.POINTER(abi.FplAdapter), C.c_int32, C.c_int32, C.c_uint32]
    L.fpl_destroy.restype = None
    L.fpl_destroy.argtypes = [C.c_void_p]
    L.fpl_process_batch_device.restype = C.c_int
    L.fpl_process_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64,
                                           C.c_uint32, C.c_void_p, C.c_void_p]
    L.fpl_process_batch.restype = C.c_int
    L.fpl_process_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.fpl_max_cycles.restype = C.c_uint32
    L.fpl_max_cycles.argtypes = [C.c_void_p]
    L.fpl_n_adapters.restype = C.c_int32
    L.fpl_n_adapters.argtypes = [C.c_void_p]
    L.fpl_counters_len.restype = C.c_size_t
    L.fpl_counters_len.argtypes = [C.c_void_p]
    L.fpl_reserve_cycles.restype = C.c_int
    L.fpl_reserve_cycles.argtypes = [C.c_void_p, C.c_uint32]
    L.fpl_counters_device_ptr.restype = C.c_void_p
    L.fpl_counters_device_ptr.argtypes = [C.c_void_p]
    L.fpl_get_counters.restype = C.c_int
    L.fpl_get_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.fpl_reset_counters.restype = C.c_int
    L.fpl_reset_counters.argtypes = [C.c_void_p]
    L.fpl_synchronize.restype = C.c_int
    L.fpl_synchronize.argtypes = [C.c_void_p]
    L.fpl_enable_timing.restype = C.c_int
    L.fpl_enable_timing.argtypes = [C.c_void_p, C.c_int]
    L.fpl_get_kernel_times.restype = C.c_int
    L.fpl_get_kernel_times.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_char_p), C.POINTER(C.c_int),
                                       C.POINTER(C.c_int)]
    L.fpl_fragment_counts.restype = C.c_int
    L.fpl_fragment_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.fpl_get_fragments.restype = C.c_int
    L.fpl_get_fragments.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    L.fpl_process_batch_async.restype = C.c_int
    L.fpl_process_batch_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.fpl_wait.restype = C.c_int
    L.fpl_wait.argtypes = [C.c_void_p]
    L.fpl_in_flight.restype = C.c_int
    L.fpl_in_flight.argtypes = [C.c_void_p]
    L.fpl_host_alloc.restype = C.c_void_p
    L.fpl_host_alloc.argtypes = [C.c_size_t]
    L.fpl_host_free.restype = None
    L.fpl_host_free.argtypes = [C.c_void_p]
    L.fpl_allreduce_counters.restype = C.c_int
    L.fpl_allreduce_counters.argtypes = [C.POINTER(C.c_void_p), C.c_int32]
    L.fpl_comm_init.restype = C.c_int
    L.fpl_comm_init.argtypes = [C.POINTER(C.c_void_p), C.c_int32]
    L.fpl_assume_inputs_ready.restype = C.c_int
    L.fpl_assume_inputs_ready.argtypes = [C.c_void_p, C.c_int]
    L.fpl_get_batch_forms.restype = C.c_int
    L.fpl_get_batch_forms.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.fpl_rccl_library.restype = C.c_char_p
    L.fpl_rccl_library.argtypes = []
    L.fpl_count_end_kmers.restype = C.c_int
    L.fpl_count_end_kmers.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                      C.POINTER(C.c_uint64)]
    L.fpl_pick_adapter.restype = C.c_int
    L.fpl_pick_adapter.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_int32, C.c_int32,
                                   C.POINTER(abi.FplAdapterPick)]
    L.fpl_process_text_async.restype = C.c_int
    L.fpl_process_text_async.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.fpl_peek_text.restype = C.c_int
    L.fpl_peek_text.argtypes = [C.c_void_p, C.POINTER(abi.FplTextResult)]
    L.fpl_start_text.restype = C.c_int
    L.fpl_start_text.argtypes = [C.c_void_p]
    L.fpl_cancel_text.restype = C.c_int
    L.fpl_cancel_text.argtypes = [C.c_void_p]
    L.fpl_wait_text.restype = C.c_int
    L.fpl_wait_text.argtypes = [C.c_void_p, C.POINTER(abi.FplTextResult), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    if L.fpl_abi_version() != abi.FPL_ABI_VERSION:
        raise FplError("ABI version mismatch")
    if path is None:
        _lib = L
    return L


def _b(s):
    return s.encode("latin-1") if isinstance(s, str) else bytes(s)


class Engine:
    """One fpl_ctx on one device."""

    def __init__(self, opt=None, start_adapter="", end_adapter="", fasta=(), device=0, max_cycles=1024, lib=None):
        self.L = lib if lib is not None else load_library()
        self.opt = opt if opt is not None else abi.FplOptions.default()
        self.start, self.end = _b(start_adapter), _b(end_adapter)
        self.fasta = [_b(a) for a in fasta]
        arr = (abi.FplAdapter * max(1, len(self.fasta)))()
        for i, a in enumerate(self.fasta):
            arr[i].seq, arr[i].len = a, len(a)
        h = C.c_void_p()
        rc = self.L.fpl_create(C.byref(h), C.byref(self.opt), self.start, len(self.start), self.end, len(self.end),
                               arr, len(self.fasta), device, max_cycles)
        if rc != 0:
            raise FplError("fpl_create: %s" % self.L.fpl_strerror(rc).decode())
        self.h = h
        self.device = device

    def _check(self, rc, what):
        if rc != 0:
            raise FplError("%s: %s (%s)" % (what, self.L.fpl_strerror(rc).decode(),
                                            (self.L.fpl_last_error(self.h) or b"").decode()))

    def close(self):
        if getattr(self, "h", None):
            self.L.fpl_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n_adapters(self):
        return 2 + len(self.fasta)

    @property
    def max_cycles(self):
        return int(self.L.fpl_max_cycles(self.h))

    def reserve_cycles(self, c):
        self._check(self.L.fpl_reserve_cycles(self.h, int(c)), "fpl_reserve_cycles")

    def process_host(self, seq, qual, off):
        """fpl_process_batch: host numpy buffers in, structured result array out."""
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        qual = np.ascontiguousarray(qual, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = len(off) - 1
        res = np.zeros(max(n, 1), dtype=abi.RESULT_DTYPE)
        if seq.size == 0:
            seq = np.zeros(1, np.uint8)
            qual = np.zeros(1, np.uint8)
        self._check(self.L.fpl_process_batch(self.h, seq.ctypes.data, qual.ctypes.data, off.ctypes.data, n,
                                             res.ctypes.data), "fpl_process_batch")
        return res[:n]

    def submit_host(self, seq, qual, off, res):
        """fpl_process_batch_async: the arrays (uint8, uint8, uint64 offsets; ideally views of pinned_array()) and
        the result array must stay alive until wait() has returned for this batch"""
        n = len(off) - 1
        self._check(self.L.fpl_process_batch_async(self.h, seq.ctypes.data, qual.ctypes.data, off.ctypes.data, n,
                                                   res.ctypes.data), "fpl_process_batch_async")

    def wait(self):
        self._check(self.L.fpl_wait(self.h), "fpl_wait")

    def submit_text(self, text):
        """fpl_process_text_async: a chunk of FASTQ text (a pinned uint8 array: pinned_array) that starts at a record and ends
        behind one; the parse runs on the device"""
        self._keep_text = getattr(self, "_keep_text", []) + [text]
        self._keep_text = self._keep_text[-(abi.FPL_MAX_IN_FLIGHT + 1):]
        self._check(self.L.fpl_process_text_async(self.h, text.ctypes.data, len(text)), "fpl_process_text_async")

    def peek_text(self):
        """fpl_peek_text: the parse's verdict for the oldest text batch (nothing of it is counted yet)"""
        out = abi.FplTextResult()
        self._check(self.L.fpl_peek_text(self.h, C.byref(out)), "fpl_peek_text")
        return {k: getattr(out, k) for k, _ in abi.FplTextResult._fields_}

    def start_text(self):
        """fpl_start_text: the per-read kernels of the next pending text batch"""
        self._check(self.L.fpl_start_text(self.h), "fpl_start_text")

    def cancel_text(self):
        self._check(self.L.fpl_cancel_text(self.h), "fpl_cancel_text")

    def wait_text(self):
        """fpl_wait_text -> (fpl_text_result as a dict, records [n] as a numpy copy, line starts [n, 4] as a numpy copy)"""
        out = abi.FplTextResult()
        rp, lp = C.c_void_p(), C.c_void_p()
        self._check(self.L.fpl_wait_text(self.h, C.byref(out), C.byref(rp), C.byref(lp)), "fpl_wait_text")
        info = {k: getattr(out, k) for k, _ in abi.FplTextResult._fields_}
        n = out.n_reads
        if out.status != 0 or n == 0:
            return info, np.zeros(0, dtype=abi.RESULT_DTYPE), np.zeros((0, 4), np.uint32)
        res = np.ctypeslib.as_array(C.cast(rp, C.POINTER(C.c_uint8)), shape=(n * 36,)).view(abi.RESULT_DTYPE).copy()
        lines = np.ctypeslib.as_array(C.cast(lp, C.POINTER(C.c_uint32)), shape=(n, 4)).copy()
        return info, res, lines

    def in_flight(self):
        return int(self.L.fpl_in_flight(self.h))

    def pinned_array(self, n, dtype=np.uint8):
        """numpy view of n items of page-locked host memory (fpl_host_alloc); freed when the array is collected"""
        nbytes = max(1, int(n) * np.dtype(dtype).itemsize)
        ptr = self.L.fpl_host_alloc(nbytes)
        if not ptr:
            raise FplError("fpl_host_alloc(%d) failed" % nbytes)
        L = self.L

        class _Owner:
            def __del__(self_inner):
                L.fpl_host_free(ptr)

        buf = (C.c_uint8 * nbytes).from_address(ptr)
        arr = np.frombuffer(buf, dtype=dtype, count=int(n))
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append((_Owner(), buf))
        return arr

    def fragments(self):
        """--break / --mask outcome of the LAST batch: (fpl_fragment records sorted by (read, seq_no), fpl_region list)"""
        nf, nr = C.c_uint32(0), C.c_uint32(0)
        self._check(self.L.fpl_fragment_counts(self.h, C.byref(nf), C.byref(nr)), "fpl_fragment_counts")
        frags = np.zeros(max(nf.value, 1), dtype=abi.FRAGMENT_DTYPE)
        regs = np.zeros(max(nr.value, 1), dtype=abi.REGION_DTYPE)
        self._check(self.L.fpl_get_fragments(self.h, frags.ctypes.data, nf.value, regs.ctypes.data, nr.value),
                    "fpl_get_fragments")
        return frags[:nf.value], regs[:nr.value]

    def process_device(self, seq_t, qual_t, off_t, max_read_len, results_t=None, stream=None):
        """fpl_process_batch_device on torch CUDA tensors (uint8, uint8, int64 offsets).
        Asynchronous on `stream` (default: torch's current stream)."""
        import torch

        n = off_t.numel() - 1
        if results_t is None:
            results_t = torch.empty(max(n, 1) * C.sizeof(abi.FplReadResult), dtype=torch.uint8, device=seq_t.device)
        if stream is None:
            stream = torch.cuda.current_stream(seq_t.device).cuda_stream
        self._check(self.L.fpl_process_batch_device(self.h, seq_t.data_ptr(), qual_t.data_ptr(), off_t.data_ptr(), n,
                                                    seq_t.numel(), int(max_read_len), results_t.data_ptr(),
                                                    C.c_void_p(stream)), "fpl_process_batch_device")
        return results_t

    @staticmethod
    def results_to_numpy(results_t, n):
        return results_t.cpu().numpy().view(abi.RESULT_DTYPE)[:n]

    def counters(self):
        n = int(self.L.fpl_counters_len(self.h))
        buf = np.zeros(n, dtype=np.int64)
        self._check(self.L.fpl_get_counters(self.h, buf.ctypes.data, n), "fpl_get_counters")
        return buf

    def counters_tensor(self):
        """Zero-copy torch view (int64) of the device counter buffer, e.g. for
        torch.distributed.all_reduce over RCCL.  Invalid after a capacity grow."""
        import torch

        n = int(self.L.fpl_counters_len(self.h))
        ptr = int(self.L.fpl_counters_device_ptr(self.h))

        class _Holder:
            pass

        hld = _Holder()
        hld.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (ptr, False), "version": 2}
        hld._keep = self
        return torch.as_tensor(hld, device="cuda:%d" % self.device)

    def reset_counters(self):
        self._check(self.L.fpl_reset_counters(self.h), "fpl_reset_counters")

    def synchronize(self):
        self._check(self.L.fpl_synchronize(self.h), "fpl_synchronize")

    def enable_timing(self, on=True):
        self._check(self.L.fpl_enable_timing(self.h, int(on)), "fpl_enable_timing")

    def assume_inputs_ready(self, yes=True):
        """the batches handed to process_device are complete on the device when the call is made (resident tensors): the end trims
        of a batch may then start beside the previous batch's kernels (fpl_assume_inputs_ready)"""
        self._check(self.L.fpl_assume_inputs_ready(self.h, int(bool(yes))), "fpl_assume_inputs_ready")

    def batch_forms(self):
        """-> dict: batches, reads, through k_trim_ends_batched, through k_stats_sorted, largest batch, batches whose end trims ran ahead
        (fpl_get_batch_forms)"""
        out = (C.c_uint64 * 6)()
        self._check(self.L.fpl_get_batch_forms(self.h, out), "fpl_get_batch_forms")
        return dict(batches=int(out[0]), reads=int(out[1]), trim_batched=int(out[2]), stats_sorted=int(out[3]), largest=int(out[4]), trims_ahead=int(out[5]))

    def kernel_times(self):
        """-> ({kernel name: ms summed over the window}, n_batches)"""
        ms = (C.c_float * 16)()
        names = (C.c_char_p * 16)()
        n, nb = C.c_int(0), C.c_int(0)
        self._check(self.L.fpl_get_kernel_times(self.h, ms, names, C.byref(n), C.byref(nb)), "fpl_get_kernel_times")
        return {names[i].decode(): float(ms[i]) for i in range(n.value)}, nb.value
