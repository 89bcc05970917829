"""Build the native pieces in-tree: the HIP library (hipcc, gfx950) and the C++ host tools."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
LIB = os.path.join(HERE, "libfastplong_amd.so")
HOST_LIB = os.path.join(HERE, "libfastplong_host.so")
CLI = os.path.join(ROOT, "bin", "fastplong_amd")


def _newer(target, srcs):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in srcs)


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def hip_sources():
    return [os.path.join(CSRC, f) for f in ("fpl_hip.hip", "kernels.h", "pipeline.h", "dev_prims.h", "dev_types.h", "adapter_pick.h")] + [
        os.path.join(ROOT, "include", "fastplong_amd.h")]


def build_hip(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> fastplong_amd/libfastplong_amd.so (cross-compiles without a GPU)"""
    srcs = hip_sources()
    if force or _newer(LIB, srcs):
        # the atomic optimizer rewrites every single-lane LDS accumulator update into a wave reduction: pure
        # overhead for the per-read bookkeeping of k_scan / k_trim_ends
        tmp = "%s.tmp.%d" % (LIB, os.getpid())  # (into a file of this process's own, then renamed: nobody ever maps a half-written library)
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm",
               "-amdgpu-atomic-optimizer-strategy=None", "-o", tmp, srcs[0]]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(tmp, LIB)
    return LIB


def host_sources():
    if not os.path.isdir(HOST):
        return []
    return sorted(os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith((".cpp", ".h")))


def build_host(force=False, verbose=False):
    """g++ -> fastplong_amd/libfastplong_host.so (FASTQ I/O, report writers) and bin/fastplong_amd (CLI)"""
    srcs = host_sources()
    cpps = [s for s in srcs if s.endswith(".cpp")]
    if not cpps:
        return None
    lib_cpps = [s for s in cpps if not s.endswith("cli.cpp")]
    hdr = os.path.join(ROOT, "include", "fastplong_amd.h")
    if force or _newer(HOST_LIB, srcs + [hdr]):
        tmp = "%s.tmp.%d" % (HOST_LIB, os.getpid())
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I" + os.path.join(ROOT, "include"),
               "-o", tmp] + lib_cpps + ["-ldl", "-lz"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(tmp, HOST_LIB)
    cli_src = os.path.join(HOST, "cli.cpp")
    if os.path.exists(cli_src) and (force or _newer(CLI, srcs + [hdr, LIB])):
        build_hip(force, verbose)
        os.makedirs(os.path.dirname(CLI), exist_ok=True)
        rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
        # the CLI only knows the C-ABI: HIP and RCCL stay behind libfastplong_amd.so (RCCL is dlopen'ed on first use)
        tmp = "%s.tmp.%d" % (CLI, os.getpid())  # (a test run with several workers may find the binary stale in all of them at once)
        cmd = ["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "include"), "-o", tmp, cli_src, "-L" + HERE, "-lfastplong_host",
               "-lfastplong_amd", "-Wl,-rpath,$ORIGIN/../fastplong_amd", "-Wl,-rpath," + os.path.join(rocm, "lib"),
               "-Wl,-rpath-link," + os.path.join(rocm, "lib"), "-ldl", "-lz"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(tmp, CLI)
    return HOST_LIB


def build_all(force=False, verbose=False):
    build_hip(force, verbose)
    build_host(force, verbose)


if __name__ == "__main__":
    build_all(force=True, verbose=True)
