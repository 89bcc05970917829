"""measurement aid: what a context costs at process start -- the HIP runtime's own start (hipInit, the first call that touches the device)
against fpl_create -- in a process without torch:  python tools/create_probe.py"""
import ctypes as C, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.perf_counter()
hip = C.CDLL("libamdhip64.so")
t1 = time.perf_counter()
hip.hipInit(0)
t2 = time.perf_counter()
hip.hipSetDevice(0)
p = C.c_void_p()
hip.hipMalloc(C.byref(p), 4096)
t3 = time.perf_counter()
from fastplong_amd import abi, engine  # noqa: E402
L = engine.load_library()
t4 = time.perf_counter()
engs = []
ts = []
for i in range(3):
    a = time.perf_counter()
    engs.append(engine.Engine(abi.FplOptions.default(cut_front=1, cut_tail=1, polyx=1), "AAGGATTCATTCCCACGGTAACAC", "GTGTTACCGTGGGAATGAATCCTT", device=0,
                              max_cycles=65536, lib=L))
    ts.append(time.perf_counter() - a)
print("dlopen libamdhip64 %.3f s, hipInit %.3f s, first device call (hipSetDevice + hipMalloc) %.3f s, dlopen libfastplong_amd %.3f s" % (
    t1 - t0, t2 - t1, t3 - t2, t4 - t3))
print("fpl_create: first %.3f s, second %.3f s, third %.3f s" % tuple(ts))
