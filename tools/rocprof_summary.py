"""Turn a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel summary table committed under profiles/.

    python tools/rocprof_summary.py <results.db> <out.md> [steps] [warm]

`steps` = train steps the profiled command ran in all, `warm` (default 1) = how many of them, from the start, are warm-up and
are DROPPED: every launch that started before the (warm + 1)-th `stem_fwd_kernel` launch (the first kernel of a train step) is
left out (the first step's launches read 3-10 % long -- cold instruction caches, first-touch page faults, the clocks ramping --
and it carries the one-off operand-measuring launches of the split-fp16 call sites), so averages and the per-step total are
over steady-state steps only.
"""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
warm = int(sys.argv[4]) if len(sys.argv) > 4 else (1 if steps > 1 else 0)
cur = sqlite3.connect(db).cursor()
try:
    raw = list(cur.execute("select name, start, duration from kernels order by start"))
except sqlite3.OperationalError:
    raw = [(n, i, d) for i, (n, d) in enumerate(cur.execute("select name, duration from kernels"))]
# step boundaries: `stem_fwd_kernel` is launched exactly once per train step and is the step's first conv -- everything that
# started before its (warm + 1)-th launch is warm-up (the first step also carries the one-off operand-measuring launches of the
# split-fp16 call sites, so dropping a fixed share of each kernel's launches would miscount)
marks = [st for name, st, _ in raw if "stem_fwd_kernel" in name]
cut = marks[warm] if (warm and len(marks) > warm) else None
by = {}
for name, st, dur in raw:
    if cut is not None and st < cut:
        continue
    by.setdefault(name, []).append(dur)
kept_steps = steps - warm if cut is not None or not warm else steps
rows = []
for name, durs in by.items():
    rows.append((name, len(durs), sum(durs), sum(durs) / len(durs), min(durs), max(durs)))
rows.sort(key=lambda r: -r[2])
total = sum(r[2] for r in rows)
with open(out, "w") as f:
    f.write(f"# rocprofv3 --kernel-trace summary ({db.split('/')[-1]}); durations in microseconds, {kept_steps} steady-state "
            f"steps ({steps} profiled, the first {warm} dropped as warm-up)\n\n")
    f.write("| kernel | calls | total us | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|\n")
    for name, n, tot, avg, mn, mx in rows:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")
        short = short.split("(")[0]
        f.write(f"| `{short}` | {n} | {tot/1e3:.1f} | {avg/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*tot/total:.2f} |\n")
    f.write(f"\nTotal kernel time {total/1e6:.2f} ms over {kept_steps} steps = {total/1e6/max(kept_steps, 1):.2f} ms/step\n")
print(open(out).read()[:3000])
