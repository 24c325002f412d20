#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6gen; rm -rf $O; mkdir -p $O
PANO=fp16 N=3 timeout 600 rocprofv3 --kernel-trace --stats -d $O/p -o t -- python tools/bench_gen.py > $O/run.log 2>&1
python tools/rocprof_summary.py $(find $O/p -name "*results.db" | head -1) $O/r06_gen_fp16_kernel_stats.md 1 0 | head -22 | cut -c1-170
rm -rf $O/p; tail -n 1 $O/run.log
