"""Instruction mix of a gfx950 kernel between its s_barrier instructions (spill traffic, matrix / vector / scalar counts).
usage: asm_regions.py file.s <substring of the kernel symbol>"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
buf, on = [], False
for l in lines:
    if re.match(r"^_Z\w+:", l) and pat in l:
        on = True
    if on:
        buf.append(l)
        if l.startswith(".Lfunc_end"):
            break
cuts = [0] + [i for i, l in enumerate(buf) if "s_barrier" in l] + [len(buf)]
for a, b in zip(cuts[:-1], cuts[1:]):
    c = collections.Counter()
    for l in buf[a:b]:
        t = l.strip().split(" ")[0]
        if t.startswith(("v_", "s_", "ds_", "buffer_", "global_", "scratch_", "flat_")):
            key = ("mfma" if t.startswith("v_mfma") else "readlane" if "readlane" in t else "writelane" if "writelane" in t else
                   "scratch" if t.startswith("scratch") else "valu" if t.startswith("v_") else "salu" if t.startswith("s_") else
                   "lds" if t.startswith("ds_") else "vmem")
            c[key] += 1
    print(f"lines {a:6d}..{b:6d}: " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
