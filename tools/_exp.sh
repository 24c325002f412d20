python -m pytest tests/test_ops_gpu.py -x -q -k "bwd_weight_winograd" 2>&1 | tail -15
for c in 0 1; do echo "== NEF_BW_WINO4=$c"; NEF_BW_WINO4=$c ITERS=10 python tools/bench_conv.py 2>&1 | grep "bwd_ww"; done
for c in 0 1; do echo "== bench NEF_BW_WINO4=$c"; NEF_BW_WINO4=$c python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])"; done
