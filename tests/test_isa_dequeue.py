"""The work-counter loops of the persistent kernels in the gfx950 code of the built library: every path to a loop's exit
test passes its dequeue atomic, and the value travels through a readlane to a scalar compare (tools/dequeue_isa.py).  A build
that breaks this never ends on the GPU; this test sees it on the CPU, from the disassembly."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_work_counter_dequeues_in_the_built_library():
    import dequeue_isa
    from fastplong_amd import build

    lib = build.build_hip()
    sites = {dequeue_isa.kernel_name(k): v for k, v in dequeue_isa.check(lib).items()}
    # the kernels that take chunks of reads / groups of 64 reads off a device counter, wave by wave
    for k in ("k_scan<4,1>", "k_scan<4,0>", "k_trim_ends_batched<4>"):  # (k_stats_sorted's block-wide dequeue goes through LDS)
        assert k in sites, (k, sorted(sites))
    for k, v in sites.items():
        for addr, cmp_addr, ok, why in v:
            assert ok, (k, hex(addr), why)
