"""Per-kernel HBM table: measured traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) joined with the
kernel-trace durations of the one-stream run (every launch alone).  Kernels are grouped by (name, grid size), i.e. by
launch shape.  HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE counts half, see
profiles/r01_pmc_traffic.md for the calibration).

    python tools/roofline_table.py <trace.db> <fetch.db> <write.db> <out.md> [steps_in_trace]
"""
import sqlite3
import sys

trace, fetch, write, out = sys.argv[1:5]
HBM_PEAK = 8.0e12


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def pmc(path, counter):
    cur = sqlite3.connect(path).cursor()
    acc = {}
    for name, grid, val in cur.execute("select kernel_name, grid_size, sum(value) from counters_collection where "
                                       "counter_name=? group by dispatch_id", (counter,)):
        acc.setdefault((short(name), grid), []).append(val)
    return {k: sum(v) / len(v) for k, v in acc.items()}


f, w = pmc(fetch, "FETCH_SIZE"), pmc(write, "WRITE_SIZE")
cur = sqlite3.connect(trace).cursor()
rows = {}
# the first profiled step is warm-up (3-10 % long, and it carries the one-off measuring launches of the split-fp16 call sites):
# everything that started before the second `stem_fwd_kernel` launch (first kernel of a train step) is dropped
STEPS = int(sys.argv[5]) if len(sys.argv) > 5 else 1
allk = list(cur.execute("select name, grid_x, grid_y, grid_z, duration, start from kernels order by start"))
marks = [st for name, _, _, _, _, st in allk if "stem_fwd_kernel" in name]
cut = marks[1] if (STEPS > 1 and len(marks) > 1) else None
for name, gx, gy, gz, dur, st in allk:
    if cut is not None and st < cut:
        continue
    rows.setdefault((short(name), gx * gy * gz), []).append(dur)
total = sum(sum(v) for v in rows.values())
lines = []
for key, durs in rows.items():
    if key not in f or key not in w:
        continue
    avg = sum(durs) / len(durs)
    by = (2 * f[key] + w[key]) * 1024.0
    lines.append((sum(durs), key, len(durs), avg, by))
lines.sort(reverse=True)
with open(out, "w") as fh:
    fh.write("# Measured HBM traffic and achieved bandwidth per launch shape (config 2, one-stream run: launches alone; warm-up step dropped)\n\n")
    fh.write("bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 from the PMC passes; time from the kernel trace; peak 8 TB/s.\n"
             "MFMA-bound kernels (conv_*) are listed for their traffic only -- their roof is the matrix peak (fp16 for the split-fp16 kernels conv_h2*, fp32 for the others).\n\n")
    fh.write("| kernel | workitems | launches | avg us | HBM MB / launch | TB/s | % of 8 TB/s | % of kernel time |\n"
             "|---|---:|---:|---:|---:|---:|---:|---:|\n")
    for tot, (name, grid), n, avg, by in lines:
        if tot / total < 0.0015:
            continue
        bw = by / (avg * 1e-9)
        fh.write(f"| `{name}` | {grid} | {n} | {avg/1e3:.1f} | {by/1e6:.0f} | {bw/1e12:.2f} | {100*bw/HBM_PEAK:.0f} | "
                 f"{100*tot/total:.2f} |\n")
print(open(out).read()[:2500])
