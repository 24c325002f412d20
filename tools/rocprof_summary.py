"""Turn a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel summary table committed under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_r01/r01_results.db profiles/r01_kernel_stats.md [steps]
"""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                        "group by name order by sum(duration) desc"))
total = sum(r[2] for r in rows)
with open(out, "w") as f:
    f.write(f"# rocprofv3 --kernel-trace summary ({db.split('/')[-1]}); durations in microseconds, "
            f"{steps} profiled steps (warm-up included)\n\n")
    f.write("| kernel | calls | total us | avg us | min us | max us | % |\n|---|---:|---:|---:|---:|---:|---:|\n")
    for name, n, tot, avg, mn, mx in rows:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")
        short = short.split("(")[0]
        f.write(f"| `{short}` | {n} | {tot/1e3:.1f} | {avg/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*tot/total:.2f} |\n")
    f.write(f"\nTotal kernel time {total/1e6:.2f} ms over {steps} steps = {total/1e6/steps:.2f} ms/step\n")
print(open(out).read()[:3000])
