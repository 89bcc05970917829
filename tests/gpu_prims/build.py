"""TEST INFRASTRUCTURE ONLY -- hipcc --offload-arch=gfx950 for tests/gpu_prims/prims_test.hip (cross-compiles without a GPU; the
.so travels to the GPU box with the snapshot)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "libprims_test.so")
SRCS = [os.path.join(HERE, "prims_test.hip")] + [os.path.join(ROOT, "fastplong_amd", "csrc", f) for f in
                                                  ("kernels.h", "dev_prims.h", "dev_types.h", "adapter_pick.h")]


def build(force=False):
    if force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in SRCS):
        hipcc = "/opt/rocm/bin/hipcc"
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm",
                               "-amdgpu-atomic-optimizer-strategy=None", "-I" + os.path.join(ROOT, "include"), "-o", LIB, SRCS[0]])
    return LIB


if __name__ == "__main__":
    print(build(force=True))
