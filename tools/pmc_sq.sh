#!/bin/bash
# SQ / GRBM counters of the conv micro-benchmark (one shape): where the waves' cycles go.   usage: tools/pmc_sq.sh "enc k7" out
# Two passes: NEF_H2=0 F4=1 (the fp32 kernels: direct forward, F(4,.) forward / backward-data form, both fp32 weight-gradient forms)
# and the default path ONLY_WHAT=wino,bwd_h2 (the split-fp16 kernels conv_h2_kernel / conv_h2w_kernel bench.py's roofline prices).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${2:-sq}
rm -rf $O && mkdir -p $O
CNT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
WARM=3 NEF_H2=0 ITERS=3 F4=1 timeout 600 rocprofv3 --pmc $CNT --kernel-trace -d $O/p1 -o t -- python tools/bench_conv.py "$1" > $O/run1.log 2>&1
WARM=3 ITERS=3 ONLY_WHAT=wino,bwd_h2 timeout 600 rocprofv3 --pmc $CNT --kernel-trace -d $O/p2 -o t -- python tools/bench_conv.py "$1" > $O/run2.log 2>&1
python - <<PY
import sqlite3, glob
out = {}
dur = {}
for d in ("$O/p1", "$O/p2"):
    dbs = glob.glob(d + "/**/*results.db", recursive=True)
    if not dbs:
        continue
    cur = sqlite3.connect(dbs[0]).cursor()
    rows = cur.execute("select kernel_name, counter_name, avg(v), count(*) from (select dispatch_id, kernel_name, counter_name, sum(value) v from counters_collection group by dispatch_id, counter_name) group by kernel_name, counter_name").fetchall()
    dur.update(dict(cur.execute("select name, avg(duration) from kernels group by name").fetchall()))
    for k, c, v, n in rows:
        out.setdefault(k, {})[c] = v
def short(k):
    return k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
with open("$O/sq.md", "w") as f:
    f.write("# SQ / GRBM counters, tools/bench_conv.py \"$1\" (rocprofv3 --pmc, per-dispatch sums averaged over the launches of a kernel)\n\n")
    direct = None
    for k, d in out.items():
        if "conv_fwd_kernel" in k and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
            direct = d["SQ_VALU_MFMA_BUSY_CYCLES"]
    for k, d in out.items():
        if "conv" not in k: continue
        f.write(f"{short(k)}  avg_dur_us={dur.get(k,0)/1e3:.1f}\n")
        for c, v in sorted(d.items()):
            f.write(f"    {c:28s} {v:.4g}\n")
        if direct and "SQ_VALU_MFMA_BUSY_CYCLES" in d and d["SQ_VALU_MFMA_BUSY_CYCLES"] > 0:
            f.write(f"    MFMA_BUSY / direct forward's   {d['SQ_VALU_MFMA_BUSY_CYCLES'] / direct:.4f}   (split-fp16: 3 x 32 / (8 x 64) = 0.1875; 13/28 = 0.4643 F(4,4)+F(4,3); 1/2 F(4,3))\n")
        if "GRBM_GUI_ACTIVE" in d and "SQ_VALU_MFMA_BUSY_CYCLES" in d and d["GRBM_GUI_ACTIVE"] > 0:
            f.write(f"    matrix-pipe occupancy          {d['SQ_VALU_MFMA_BUSY_CYCLES'] / (d['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}   (busy cycles / (cycles per XCD x 1024 SIMDs))\n")
print(open("$O/sq.md").read())
PY
rm -rf $O/p1 $O/p2
