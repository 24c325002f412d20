#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -x -k "stem or batchnorm or bn or decoder or sgd_steps or train_golden" 2>&1 | tail -n 4 | cut -c1-200
O=gpurun_out/r6s; rm -rf $O; mkdir -p $O
NEF_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d $O/one -o t -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-secondary --no-graph > $O/one.log 2>&1
python tools/rocprof_summary.py $(find $O/one -name "*results.db" | head -1) $O/ks.md 4 1 > /dev/null; rm -rf $O/one
grep "bn_slots\|stem_bwd_weight_reduce\|Total" $O/ks.md | cut -c1-120
