#!/usr/bin/env python3
"""Static check of the v_dot4 results in the gfx950 code of libfastplong_amd.so.

Measured on the MI355X (round 4, k_stats_sorted's row set-up): a VALU instruction that reads the result of
v_dot4_u32_u8 one instruction behind it gets a stale register -- the hardware needs wait states between a dot product and
a different instruction that touches its destination (LLVM's hazard recogniser knows the rule for gfx90a:
DotWriteDifferentVALURead 3, DotWriteDifferentVALUWrite 4) and this hipcc does not insert them for gfx950.  The kernels
therefore put their own `s_nop` behind a group of dot products (dev_prims.h::dot_settle); this script proves it for the
built library: for every v_dot4 it walks the code that can follow (fall-through and branch targets) and fails when any
instruction other than a same-opcode dot taking it as accumulator reads the destination register within 3 wait states or
overwrites it within 4 (k_scan's planes are read 3 wait states behind their last dot in places and have been bit-exact on the
GPU over every soak; one wait state is what failed).

usage: python tools/dot_hazard_isa.py [lib.so]      (exit code 1 on a violation)
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dequeue_isa import ROOT, disassemble, kernel_name  # noqa: E402

WAIT_READ, WAIT_WRITE = 3, 4  # wait states behind a dot product before its destination is read / overwritten
WAIT = WAIT_WRITE


def regs_of(args):
    """-> set of VGPR numbers an operand string names"""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", args):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def check_kernel(code):
    """-> [(addr of the dot, addr of the instruction that comes too early, wait states between them)]"""
    index = {a: i for i, (a, _o, _a, _t) in enumerate(code)}
    bad = []
    for i, (addr, op, args, _t) in enumerate(code):
        if not op.startswith("v_dot"):
            continue
        dst = int(re.match(r"v(\d+)", args.split(",")[0].strip()).group(1))
        work = [(i + 1, 0)]
        seen = set()
        while work:
            j, ws = work.pop()
            while j < len(code) and ws < WAIT and (j, ws) not in seen:
                seen.add((j, ws))
                a2, o2, g2, t2 = code[j]
                ops = [x.strip() for x in g2.split(",")]
                if dst in regs_of(g2):
                    # a dot of the same opcode that takes it as accumulator: forwarded, no wait (the chain goes on with that dot,
                    # which is checked on its own)
                    acc = o2 == op and len(ops) >= 4 and regs_of(ops[3]) == {dst} and dst not in regs_of(",".join(ops[1:3]))
                    only_written = o2.startswith("v_") and dst in regs_of(ops[0]) and dst not in regs_of(",".join(ops[1:]))
                    if not acc and ws < (WAIT_WRITE if only_written else WAIT_READ):
                        bad.append((addr, a2, ws))
                    break
                if o2 == "s_endpgm":
                    break
                if o2.startswith(("s_cbranch", "s_branch")) and t2 in index:
                    work.append((index[t2], ws + 1))
                    if o2 == "s_branch":
                        break
                ws += (int(g2.split()[0]) + 1) if o2 == "s_nop" else 1
                j += 1
    return bad


def check(lib):
    res = {}
    for name, code in disassemble(lib).items():
        n = sum(1 for c in code if c[1].startswith("v_dot"))
        if n:
            res[name] = (n, check_kernel(code))
    return res


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "fastplong_amd", "libfastplong_amd.so")
    rc = 0
    for name, (n, bad) in sorted(check(lib).items()):
        print("%-4s %-44s %3d dot products, %d results touched too early" % ("ok" if not bad else "BAD", kernel_name(name)[:44], n, len(bad)))
        for a, b, ws in bad[:6]:
            print("       dot at 0x%x, destination named at 0x%x after %d wait states" % (a, b, ws))
        rc |= 1 if bad else 0
    sys.exit(rc)
