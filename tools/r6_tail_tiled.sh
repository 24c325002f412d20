#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_pano_gpu.py -q -x 2>&1 | tail -n 6 | cut -c1-300
timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -k "heavy_tail" 2>&1 | tail -n 4 | cut -c1-300
for rep in 1 2; do
for v in 1 0; do
  NEF_DIAG=1 NEF_PANO_FUSE_TAIL=$v PANO=fp16 timeout 300 python tools/bench_gen.py 2>/dev/null | tail -n 1 | cut -c1-80 | sed "s#^#FUSE_TAIL=$v #"
done; done
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ms', d['ms_per_step'], 'tail sites', d['h2_tail_sites'], 'worst (count, energy)', d['h2_tail_worst'])"
