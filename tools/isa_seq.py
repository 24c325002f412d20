"""Condensed instruction sequence of one kernel from a hipcc -S dump:  python tools/isa_seq.py file.s <mangled-substring> [from_line to_line]"""
import re, sys
src, pat = sys.argv[1], sys.argv[2]
lines = open(src).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and pat in l and l.rstrip().endswith(":") or (pat in l and re.match(r"^_ZN.*: ", l)))
end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel" in lines[i])
keep = re.compile(r"buffer_load|global_load|v_mfma|s_waitcnt|s_barrier|ds_write|ds_read|s_cbranch|^\.LBB|buffer_store|global_store|s_setprio|s_endpgm|scratch_")
prev, cnt, first = None, 0, None
def flush():
    if prev is not None:
        print(f"{first:6d}: {prev}" + (f"  x{cnt}" if cnt > 1 else ""))
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 10**9
for i in range(start, end):
    l = lines[i].strip()
    if not keep.search(l):
        continue
    rel = i - start
    if rel < lo or rel > hi:
        continue
    toks = l.split()
    key = toks[0] if not toks[0].startswith("s_waitcnt") else " ".join(toks[:2])
    if key.startswith(".LBB"):
        key = toks[0]
    if key == prev:
        cnt += 1
    else:
        flush()
        prev, cnt, first = key, 1, rel
flush()
