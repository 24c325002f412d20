#!/bin/bash
# SQ / GRBM counters of the split-fp16 conv kernels WITH their decoder prologues / epilogues (tools/h2p_check.py cases, default
# kernel form, + the weight gradient of the same case): conv_h2_kernel<3, PRO, TM> and conv_h2w2_kernel<3, PRO, ..>.
# usage: tools/pmc_sq_dec.sh <out dir under gpurun_out> ["case filter"]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-sqdec}
rm -rf $O && mkdir -p $O
CNT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
FORMS=0 WG=1 CHECK=0 ITERS=3 WARM=3 timeout 600 rocprofv3 --pmc $CNT --kernel-trace -d $O/p -o t -- python tools/h2p_check.py "${2:-dec}" > $O/run.log 2>&1
python - <<PY
import sqlite3, glob, json
dbs = glob.glob("$O/p/**/*results.db", recursive=True)
if not dbs:
    print(open("$O/run.log").read()[-2000:]); raise SystemExit
cur = sqlite3.connect(dbs[0]).cursor()
rows = cur.execute("select kernel_name, grid_size, counter_name, avg(v), count(*) from (select dispatch_id, kernel_name, grid_size, counter_name, sum(value) v from counters_collection group by dispatch_id, counter_name) group by kernel_name, grid_size, counter_name").fetchall()
dur = {(k, g): d for k, g, d in cur.execute("select name, grid_size, avg(duration) from kernels group by name, grid_size").fetchall()} if False else {}
out = {}
for k, g, c, v, n in rows:
    out.setdefault((k, g), {})[c] = v
durs = {}
try:
    for k, g, d in cur.execute("select kernel_name, grid_size, avg(end - start) from (select distinct dispatch_id, kernel_name, grid_size, start, end from counters_collection) group by kernel_name, grid_size").fetchall():
        durs[(k, g)] = d
except Exception as e:
    pass
js = {}
with open("$O/sq.md", "w") as f:
    f.write("# SQ / GRBM counters of the split-fp16 conv kernels behind their decoder prologues (tools/h2p_check.py, FORMS=0 WG=1; rocprofv3 --pmc, per-dispatch sums averaged over the launches of a kernel and grid)\n\n")
    for (k, g), d in sorted(out.items()):
        if "conv_h2" not in k: continue
        short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        f.write(f"{short}  grid={g}\n")
        for c, v in sorted(d.items()):
            f.write(f"    {c:28s} {v:.4g}\n")
        if d.get("GRBM_GUI_ACTIVE", 0) > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
            occ = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8 * 1024)
            f.write(f"    matrix-pipe occupancy          {occ:.3f}   (busy cycles / (cycles per XCD x 1024 SIMDs))\n")
            js[f"{short}@{g}"] = dict(pipe_busy=round(occ, 3), gui_active_per_xcd=d["GRBM_GUI_ACTIVE"] / 8,
                                      wait_any_frac=round(d.get("SQ_WAIT_ANY", 0) / max(d.get("SQ_WAVE_CYCLES", 1), 1), 3),
                                      wait_inst_frac=round(d.get("SQ_WAIT_INST_ANY", 0) / max(d.get("SQ_WAVE_CYCLES", 1), 1), 3))
json.dump(js, open("$O/sq.json", "w"), indent=1)
print(open("$O/sq.md").read())
PY
rm -rf $O/p
