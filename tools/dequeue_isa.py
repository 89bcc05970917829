#!/usr/bin/env python3
"""Static check of the work-counter dequeues in the gfx950 code of libfastplong_amd.so.

The persistent kernels take their work off device counters:

    for (;;) { u32 it = 0; if (lane == 0) it = atomicAdd(ctr, k); it = readlane(it, 0); if (it >= n) break; ... }

The loop only ends if every path to the exit test passes the atomic: a build in which the control-flow structurizer
puts a back edge BEHIND the dequeue (the exit test then sees the lane-0 value of the previous trip for ever) never
terminates on the GPU -- what round 3 saw once in a form of k_redo.  This script finds every
dequeue in the disassembly

    s_and_saveexec ...; s_cbranch_execz join; global_atomic_add vD, ... sc0; join: s_or exec; s_waitcnt vmcnt(0);
    v_readlane_b32 sX, vD, 0 | v_readfirstlane_b32 sX, vD;  s_cmp_*_u32 sX, ...

and checks (1) that the value reaches the compare through a readlane (wave-uniform, scalar branch) and (2) that no
branch from outside that sequence lands inside it.

usage: python tools/dequeue_isa.py [lib.so]      (exit code 1 when a dequeue fails the check)
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def disassemble(lib):
    """-> {kernel symbol: [(addr, op, operands, branch target addr or None)]}"""
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fatbin"), os.path.join(d, "co.o")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib],
                              stderr=subprocess.DEVNULL)
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co])
        text = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", co]).decode("latin-1")
    funcs = {}
    cur, base = None, 0
    head = re.compile(r"^([0-9a-f]{16}) <(\S+)>:")
    ins = re.compile(r"^\t(\S+)\s*(.*?)\s*// ([0-9A-F]+): [0-9A-F ]+(?:<\S+\+0x([0-9a-f]+)>)?\s*$")
    for line in text.split("\n"):
        m = head.match(line)
        if m:
            base = int(m.group(1), 16)
            cur = funcs.setdefault(m.group(2), [])
            continue
        m = ins.match(line)
        if m and cur is not None:
            tgt = base + int(m.group(4), 16) if (m.group(4) and m.group(1).startswith(("s_cbranch", "s_branch"))) else None
            cur.append((int(m.group(3), 16), m.group(1), m.group(2), tgt))
    return funcs


def dequeues(code):
    """-> [(atomic addr, compare addr or None, ok, why)] for the returning counter atomics whose value feeds a scalar compare"""
    out = []
    for i, (addr, op, args, _) in enumerate(code):
        if not (op.startswith("global_atomic_add") and "sc0" in args and "x2" not in op):
            continue
        vd = args.split(",")[0].strip()
        # the lane-0 value on the scalar unit
        sx, j_rl = None, None
        for j in range(i + 1, min(i + 14, len(code))):
            o, a = code[j][1], code[j][2]
            if o in ("v_readlane_b32", "v_readfirstlane_b32") and a.split(",")[1].strip() == vd:
                sx, j_rl = a.split(",")[0].strip(), j
                break
            if o.startswith(("s_cbranch", "s_branch", "s_endpgm")) and code[j][3] is not None and code[j][3] < addr:
                break
        if sx is None:
            continue  # (a list reservation whose value stays in vector registers, or no dequeue at all)
        j_cmp = None
        for j in range(j_rl + 1, min(j_rl + 8, len(code))):
            o, a = code[j][1], code[j][2]
            if o.startswith("s_cmp_") and sx in [x.strip() for x in a.split(",")]:
                j_cmp = j
                break
            if o.startswith(("s_", "v_")) and a.split(",")[0].strip() == sx:
                break  # overwritten
        if j_cmp is None:
            continue  # (a place reservation: base = readlane(atomicAdd(..)) used as an address, no exit test)
        # the guard `if (lane == 0)` / `if (threadIdx.x == 0)` in front of the atomic belongs to the sequence: an s_cbranch_execz a
        # few instructions ahead of it that lands behind it, in front of the readlane (the join)
        a_lo, a_hi = addr, code[j_cmp][0]
        bad = []
        for k, (s_addr, o, _a, t) in enumerate(code):
            if t is None or not (a_lo < t <= a_hi) or a_lo <= s_addr <= a_hi:
                continue
            if o == "s_cbranch_execz" and i - 12 <= k < i and t <= code[j_rl][0]:
                continue  # (the guard: it skips the atomic -- and, in a block-wide dequeue, the LDS store behind it -- for the other lanes)
            bad.append((s_addr, t))
        if bad:
            out.append((addr, a_hi, False, "a branch at 0x%x lands inside the dequeue (0x%x): a path to the exit test that skips the atomic" % bad[0]))
        else:
            out.append((addr, a_hi, True, "atomic -> %s -> %s; no outside branch lands between them" % (code[j_rl][1], code[j_cmp][1])))
    return out


def kernel_name(sym):
    """_ZN3fpl6k_scanILi4ELb1EEEv... -> k_scan<4,1>"""
    m = re.match(r"_ZN3fpl(\d+)", sym)
    if not m:
        return sym[:40]
    n = int(m.group(1))
    at = m.end()
    name, rest = sym[at:at + n], sym[at + n:]
    t = re.match(r"I((?:L[ib]\d+E)+)E", rest)
    if t:
        name += "<" + ",".join(re.findall(r"L[ib](\d+)E", t.group(1))) + ">"
    return name


def check(lib):
    res = {}
    for name, code in disassemble(lib).items():
        d = dequeues(code)
        if d:
            res[name] = d
    return res


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "fastplong_amd", "libfastplong_amd.so")
    rc = 0
    for name, sites in sorted(check(lib).items()):
        short = kernel_name(name)
        for addr, cmp_addr, ok, why in sites:
            print("%-4s %-40s atomic 0x%x  test 0x%x  %s" % ("ok" if ok else "BAD", short[:40], addr, cmp_addr, why))
            rc |= 0 if ok else 1
    sys.exit(rc)
