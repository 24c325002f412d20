#!/bin/bash
# Arbitrary PMC group on the conv micro-benchmark.  usage: tools/pmc_any.sh "<shape filter>" <out dir under gpurun_out> COUNTER...
# env passes through (NEF_LIB, F4, ONLY_WHAT).  One --pmc group per call: counters of one pass must fit the hardware.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
SH="$1"; O=gpurun_out/$2; shift 2
rm -rf $O && mkdir -p $O
ITERS=3 timeout 600 rocprofv3 --pmc "$@" --kernel-trace -d $O/p -o t -- python tools/bench_conv.py "$SH" > $O/run.log 2>&1
python - <<PY
import sqlite3, glob
dbs = glob.glob("$O/p/**/*results.db", recursive=True)
if not dbs:
    print(open("$O/run.log").read()[-1500:]); raise SystemExit
cur = sqlite3.connect(dbs[0]).cursor()
rows = cur.execute("select kernel_name, counter_name, avg(v), count(*) from (select dispatch_id, kernel_name, counter_name, sum(value) v from counters_collection group by dispatch_id, counter_name) group by kernel_name, counter_name").fetchall()
dur = dict(cur.execute("select name, avg(duration) from kernels group by name").fetchall())
out = {}
for k, c, v, n in rows:
    out.setdefault(k, {})[c] = v
with open("$O/pmc.md", "w") as f:
    for k, d in out.items():
        if "conv" not in k or "reduce" in k: continue
        short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        f.write(f"{short}  avg_dur_us={dur.get(k,0)/1e3:.1f}\n")
        for c, v in sorted(d.items()):
            f.write(f"    {c:36s} {v:.5g}\n")
print(open("$O/pmc.md").read())
PY
rm -rf $O/p
