#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-small}
rm -rf $O && mkdir -p $O
ITERS=3 timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $O/p -o t -- python tools/bench_small.py > $O/run.log 2>&1
python - <<PY
import sqlite3, glob
db = glob.glob("$O/p/**/*results.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select kernel_name, counter_name, avg(v) from (select dispatch_id, kernel_name, counter_name, sum(value) v from counters_collection group by dispatch_id, counter_name) group by kernel_name, counter_name").fetchall()
dur = dict(cur.execute("select name, avg(duration) from kernels group by name").fetchall())
out = {}
for k, c, v in rows:
    out.setdefault(k, {})[c] = v
with open("$O/sq.md", "w") as f:
    for k, d in out.items():
        short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        f.write(f"{short}  avg_dur_us={dur.get(k,0)/1e3:.1f}\n")
        for c, v in sorted(d.items()):
            f.write(f"    {c:28s} {v:.4g}\n")
print(open("$O/sq.md").read())
PY
rm -rf $O/p
