"""MEASUREMENT INFRASTRUCTURE ONLY: builds tools/nulldev/libfastplong_amd.so, a NULL device behind the C-ABI (see fpl_null.cpp).
bench.py's e2e.host_ceiling leg puts this directory on LD_LIBRARY_PATH of bin/fastplong_amd; nothing else ever loads it."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "libfastplong_amd.so")
SRCS = [os.path.join(HERE, "fpl_null.cpp"), os.path.join(ROOT, "include", "fastplong_amd.h")]


def build():
    if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in SRCS):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", LIB, SRCS[0]])
    return LIB


if __name__ == "__main__":
    print(build())
