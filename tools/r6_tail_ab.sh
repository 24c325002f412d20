#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_pano_gpu.py -q -x 2>&1 | tail -n 8 | cut -c1-300
for rep in 1 2; do
for v in 1 0; do
  NEF_DIAG=1 NEF_PANO_FUSE_TAIL=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['secondary']; print('FUSE_TAIL=$v', 'configs[3]', s['configs[3]']['ms'], s['configs[3]']['hbm_frac'], s['configs[3]']['mfma_frac'], 'configs[4]', s['configs[4] share']['ms'])"
done; done
