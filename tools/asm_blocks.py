#!/usr/bin/env python3
"""Static instruction mix per basic block of one kernel in a hipcc --save-temps .s file.
usage: asm_blocks.py file.s kernel_substring [min_valu]"""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
minv = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(tuple([":"])) is False and ":" in l)
end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel" in lines[i] or lines[i].startswith("\t.section\t.rodata"))
blocks = []
cur = ["entry", 0, 0, 0, 0, 0, start]
for i in range(start + 1, end):
    l = lines[i].strip()
    if not l or l.startswith(";"):
        continue
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur)
        cur = [m.group(1), 0, 0, 0, 0, 0, i]
        continue
    op = l.split()[0]
    if op.startswith("v_"):
        cur[1] += 1
    elif op.startswith("s_"):
        cur[2] += 1
    elif op.startswith("ds_"):
        cur[3] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        cur[4] += 1
        if op.startswith("scratch_"):
            cur[5] += 1
blocks.append(cur)
tot = [sum(b[k] for b in blocks) for k in range(1, 6)]
print("total valu %d salu %d lds %d vmem %d (scratch %d) blocks %d" % (*tot, len(blocks)))
for b in blocks:
    if b[1] >= minv:
        print("%-14s line %6d valu %5d salu %5d lds %4d vmem %3d scratch %3d" % (b[0], b[6] + 1, b[1], b[2], b[3], b[4], b[5]))
