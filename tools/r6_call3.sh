#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -n 6 $O/pytest.log | cut -c1-300
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-kernel-events --no-secondary --steps 30 --warmup 5 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'])"; done
timeout 300 python bench.py --batch 32 --len 512 --no-cpu-baseline --no-kernel-events --no-secondary --steps 200 --warmup 20 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('small graph', d['ms_per_step'])"
