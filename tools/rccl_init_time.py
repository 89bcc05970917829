"""usage (on a GPU box): python tools/rccl_init_time.py [notorch]
Measurement aid: what the first use of RCCL costs a process, and what a communicator costs after that -- fpl_comm_init (one rank,
FPL_RCCL_FORCE=1) three times, the kept communicators given back in between.  `notorch`: the library loaded the way
bin/fastplong_amd has it (no PyTorch in the process, so no librccl mapped yet)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["FPL_RCCL_FORCE"] = "1"
from fastplong_amd import abi  # noqa: E402
from fastplong_amd import engine as E  # noqa: E402

L = C.CDLL(E.LIB_PATH) if len(sys.argv) > 1 else E.load_library()
L.fpl_create.restype = C.c_int
L.fpl_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(abi.FplOptions), C.c_char_p, C.c_int32, C.c_char_p, C.c_int32,
                         C.POINTER(abi.FplAdapter), C.c_int32, C.c_int32, C.c_uint32]
L.fpl_comm_init.restype = C.c_int
L.fpl_comm_init.argtypes = [C.POINTER(C.c_void_p), C.c_int32]
opt = abi.FplOptions.default()
h = C.c_void_p()
t = time.time()
rc = L.fpl_create(C.byref(h), C.byref(opt), b"ACGTACGTACGTACGTAC", 18, b"TTGACCAGTAGGCATCAG", 18, (abi.FplAdapter * 1)(), 0, 0, 1024)
print("fpl_create rc=%d %.3f s" % (rc, time.time() - t))
arr = (C.c_void_p * 1)(h)
for i in range(3):
    t = time.time()
    rc = L.fpl_comm_init(arr, 1)
    print("fpl_comm_init #%d rc=%d %.3f s" % (i, rc, time.time() - t))
    L.fpl_comm_init(None, 0)
