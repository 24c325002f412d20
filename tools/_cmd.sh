cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 600 python bench.py --leads 8 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-900
