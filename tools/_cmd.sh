cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "nefnet2 or golden or oracle_live or graphed" 2>&1 | tail -12
