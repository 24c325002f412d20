#!/bin/bash
# rocprofv3 kernel stats of the reference-native training shape (batch 32, 3 leads, L = 512; codes/train_net.py:27-28), eager, one stream.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-small32}
rm -rf $O && mkdir -p $O
NEF_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d $O/one -o t -- python bench.py --batch 32 --len 512 --steps 9 --warmup 1 --no-cpu-baseline --no-kernel-events --no-secondary --no-graph > $O/one.log 2>&1
python tools/rocprof_summary.py $(find $O/one -name "*results.db" | head -1) $O/kernel_stats.md 10 1 > /dev/null
rm -rf $O/one
head -60 $O/kernel_stats.md; tail -3 $O/kernel_stats.md
