"""TEST INFRASTRUCTURE ONLY: builds tests/stub/libfastplong_amd.so, a stand-in for the C-ABI library whose "devices" compute
with the oracle (see fpl_stub.cpp).  Loaded by the CLI only when a test puts this directory on LD_LIBRARY_PATH."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "libfastplong_amd.so")
SRCS = [os.path.join(HERE, "fpl_stub.cpp"), os.path.join(ROOT, "oracle", "fpl_oracle.c"), os.path.join(ROOT, "oracle", "fpl_oracle.h"),
        os.path.join(ROOT, "include", "fastplong_amd.h")]


def build():
    if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in SRCS):
        obj = os.path.join(HERE, "fpl_oracle.o")
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-c", "-o", obj, SRCS[1]])
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", LIB, SRCS[0], obj, "-lm"])
    return LIB


if __name__ == "__main__":
    print(build())
