#!/bin/bash
# One GPU call: serialized + two-stream rocprofv3 kernel stats (warm-up step dropped from the averages), the two PMC traffic
# passes of bench.py (calibrated on chscale_bwd_kernel -- the two PMC passes run with NEF_FOLD_CHSCALE_BWD=0 so that this pure
# streaming pass of known traffic is in the trace; it is folded into a conv epilogue in the default path --; the script fails on a missing / off calibration) and the SQ counters of
# the K=7 conv family: the fp32 kernels (NEF_H2=0) and the split-fp16 kernels of the default path, reduced to the
# markdown summaries under gpurun_out/ (copy the ones to keep into profiles/).   usage: tools/profile_round.sh r02
R=${1:-r02}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$R
# bench.py --steps 3 --warmup 1 --no-kernel-events runs exactly 4 train steps (no breakdown steps, no inference configs):
# the summarisers divide by THIS number (round 2 passed 6 while the run did 7 steps: its per-step totals were 7/6 too high)
STEPS=4
rm -rf $O && mkdir -p $O
NEF_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d $O/one -o t -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-secondary --no-graph > $O/one.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/two -o t -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-secondary --no-graph > $O/two.log 2>&1
NEF_DIAG=1 NEF_FOLD_CHSCALE_BWD=0 NEF_SIDE_STREAM=0 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o t -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --no-secondary --no-graph > $O/fetch.log 2>&1
NEF_DIAG=1 NEF_FOLD_CHSCALE_BWD=0 NEF_SIDE_STREAM=0 timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o t -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --no-secondary --no-graph > $O/write.log 2>&1
python tools/rocprof_summary.py $(find $O/one -name "*results.db" | head -1) $O/${R}_kernel_stats_serialized.md $STEPS 1 > /dev/null
python tools/rocprof_summary.py $(find $O/two -name "*results.db" | head -1) $O/${R}_kernel_stats.md $STEPS 1 > /dev/null
python tools/pmc_table.py $(find $O/fetch -name "*results.db" | head -1) $(find $O/write -name "*results.db" | head -1) $O/${R}_pmc_traffic.md $O/traffic.json > /dev/null 2> $O/pmc.err
python tools/roofline_table.py $(find $O/one -name "*results.db" | head -1) $(find $O/fetch -name "*results.db" | head -1) $(find $O/write -name "*results.db" | head -1) $O/${R}_hbm_kernels.md $STEPS > /dev/null 2>> $O/pmc.err
bash tools/pmc_sq.sh "enc k7" $R/sq > /dev/null 2>&1
cp $O/sq/sq.md $O/${R}_sq_counters_k7.md 2>/dev/null
cat $O/pmc.err
rm -rf $O/one $O/two $O/fetch $O/write
ls -la $O
