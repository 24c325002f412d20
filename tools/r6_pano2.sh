#!/bin/bash
# fp16 panorama decoder after the round-6 second pass (32 x 128 wave tiles, packed epilogues, padding rows, ...): tests, sweep, gen_ecg share, kernel trace
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6pano2; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_pano_gpu.py -q -x 2>&1 | tail -n 3 | cut -c1-200
for rep in 1 2 3; do PANO=fp16 timeout 300 python tools/bench_sweep.py 2>/dev/null | tail -n 1 | cut -c1-120; done
for rep in 1 2; do PANO=fp16 timeout 300 python tools/bench_gen.py 2>/dev/null | tail -n 1 | cut -c1-160; done
PANO=fp16 timeout 600 rocprofv3 --kernel-trace --stats -d $O/p -o t -- python tools/bench_sweep.py > $O/run.log 2>&1
python tools/rocprof_summary.py $(find $O/p -name "*results.db" | head -1) $O/r06_pano_fp16_kernel_stats.md 1 0 | head -12 | cut -c1-170
rm -rf $O/p
