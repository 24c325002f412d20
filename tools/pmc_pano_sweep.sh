#!/bin/bash
# SQ / GRBM counters of the fp16 panorama decoder kernels inside the real configs[3] sweep (tools/bench_sweep.py, PANO=fp16)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-pano_sweep_sq}
rm -rf $O && mkdir -p $O
PANO=fp16 timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $O/p -o t -- python tools/bench_sweep.py > $O/run.log 2>&1
python - <<PY
import sqlite3, glob
db = glob.glob("$O/p/**/*results.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select kernel_name, counter_name, avg(v), count(*) from (select dispatch_id, kernel_name, counter_name, sum(value) v from counters_collection group by dispatch_id, counter_name) group by kernel_name, counter_name").fetchall()
dur = dict(cur.execute("select name, avg(duration) from kernels group by name").fetchall())
out = {}
for k, c, v, n in rows:
    out.setdefault(k, {})[c] = v
with open("$O/sq.md", "w") as f:
    for k, d in out.items():
        if "hconv" not in k: continue
        us = dur.get(k, 0) / 1e3
        f.write(f"{k.split('(')[0]}  avg_dur_us={us:.1f}\n")
        for c, v in sorted(d.items()):
            f.write(f"    {c:28s} {v:.4g}\n")
        if us and "GRBM_GUI_ACTIVE" in d:
            f.write(f"    effective clock (GRBM_GUI_ACTIVE / 8 XCDs / time)  {d['GRBM_GUI_ACTIVE'] / 8 / us / 1e3:.2f} GHz\n")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
            f.write(f"    matrix pipes busy (MFMA_BUSY / (1024 SIMDs x GRBM_GUI_ACTIVE / 8))  {d['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * d['GRBM_GUI_ACTIVE'] / 8):.3f}\n")
print(open("$O/sq.md").read())
PY
rm -rf $O/p
