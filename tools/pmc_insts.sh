#!/bin/bash
# Dynamic instruction mix (SQ_INSTS_*) of the split-fp16 conv kernels at the bench shapes: tools/pmc_insts.sh <out> "<bench_conv filter>"
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-insts}
rm -rf $O && mkdir -p $O
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_MFMA"; do
  tag=$(echo $pass | cut -d' ' -f1)
  ITERS=3 WARM=3 ONLY_WHAT=wino,bwd_h2 timeout 600 rocprofv3 --pmc $pass --kernel-trace -d $O/p_$tag -o t -- python tools/bench_conv.py "${2:-k3 128->128 g3}" > $O/run_$tag.log 2>&1
  tail -n 3 $O/run_$tag.log | cut -c1-200
done
python - <<PY
import sqlite3, glob
out = {}
for db in glob.glob("$O/p_*/**/*results.db", recursive=True):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, counter_name, avg(v) from (select dispatch_id, kernel_name, counter_name, sum(value) v from counters_collection group by dispatch_id, counter_name) group by kernel_name, counter_name").fetchall()
    for k, c, v in rows:
        out.setdefault(k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0], {})[c] = v
with open("$O/insts.md", "w") as f:
    for k, d in sorted(out.items()):
        if "conv_h2" not in k and "hconv" not in k: continue
        w = d.get("SQ_WAVES", 0) or 1
        f.write(k + "\n")
        for c, v in sorted(d.items()):
            f.write(f"    {c:20s} {v:.4g}   per wave {v / w:.1f}\n")
        if d.get("SQ_INSTS_MFMA"):
            tot = sum(v for c, v in d.items() if c.startswith("SQ_INSTS"))
            f.write(f"    instructions per matrix instruction: {tot / d['SQ_INSTS_MFMA']:.2f} (vector {d.get('SQ_INSTS_VALU', 0) / d['SQ_INSTS_MFMA']:.2f} incl. the matrix instructions themselves)\n")
print(open("$O/insts.md").read())
PY
rm -rf $O/p_*
