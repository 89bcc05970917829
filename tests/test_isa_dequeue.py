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
    for k in ("k_scan<4,1>", "k_scan<4,0>", "k_trim_ends_batched<4,4,0>", "k_trim_ends_batched<4,8,0>", "k_trim_ends_batched<4,8,1>"):  # (k_stats_sorted's block-wide dequeue goes through LDS)
        assert k in sites, (k, sorted(sites))
    for k, v in sites.items():
        for addr, cmp_addr, ok, why in v:
            assert ok, (k, hex(addr), why)


def test_dot_product_results_are_given_their_wait_states():
    """gfx950 needs wait states between a v_dot4 and another instruction that touches its destination; this hipcc inserts none
    (found on the GPU in round 4: a row set-up that read its packs one instruction behind the dots counted wrong 5-mers).  The
    kernels put s_nop behind such dots (dev_prims.h::dot_settle); tools/dot_hazard_isa.py walks the code behind every dot of the
    built library."""
    import dot_hazard_isa
    from fastplong_amd import build

    lib = build.build_hip()
    res = {dot_hazard_isa.kernel_name(k): v for k, v in dot_hazard_isa.check(lib).items()}
    for k in ("k_scan<4,1>", "k_stats_sorted<16>"):
        assert k in res and res[k][0] > 0, (k, sorted(res))
    for k, (n, bad) in res.items():
        assert not bad, (k, [(hex(a), hex(b), ws) for a, b, ws in bad[:4]])


def test_dot_hazard_checker_sees_a_result_read_too_early():
    """the checker on hand-made instruction lists: a read one wait state behind the dot is flagged, a same-opcode dot taking
    the result as accumulator is not, an s_nop counts its wait states, a branch target is followed"""
    import dot_hazard_isa as d

    dot = (0, "v_dot4_u32_u8", "v5, v1, s2, 0", None)
    use = lambda a: (a, "v_lshl_or_b32", "v6, v5, 8, v7", None)  # noqa: E731
    other = lambda a: (a, "v_and_b32_e32", "v9, 1, v8", None)  # noqa: E731
    assert d.check_kernel([dot, other(4), use(8)]) == [(0, 8, 1)]
    assert d.check_kernel([dot, other(4), other(8), other(12), use(16)]) == []
    assert d.check_kernel([dot, (4, "s_nop", "2", None), use(8)]) == []
    assert d.check_kernel([dot, (4, "s_nop", "1", None), use(8)]) == [(0, 8, 2)]
    assert d.check_kernel([dot, (4, "v_dot4_u32_u8", "v5, v3, s2, v5", None), other(8), other(12), other(16), use(20)]) == []
    assert d.check_kernel([dot, (4, "v_dot4_u32_u8", "v5, v5, s2, v5", None), other(8), other(12), other(16), use(20)]) == [(0, 4, 0)]
    branchy = [dot, (4, "s_cbranch_scc1", "12", 12), other(8), (12, "v_mov_b32_e32", "v5, 0", None)]
    assert sorted(d.check_kernel(branchy)) == [(0, 12, 1), (0, 12, 2)]  # (over the branch and over the fall-through)
    assert d.check_kernel([dot, other(4), other(8), other(12), (16, "v_mov_b32_e32", "v5, 0", None)]) == [(0, 16, 3)]  # (overwriting needs 4)
