python -m pytest tests/test_ops_gpu.py -x -q -k "slot_sums" 2>&1 | tail -5
for c in 0 1; do echo "== bench NEF_FUSE_STATS=$c"; NEF_FUSE_STATS=$c python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])"; done
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
