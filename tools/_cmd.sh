cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -x -q 2>&1 | tail -4
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
mkdir -p gpurun_out/prof_bn
NEF_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bn -o bn -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bn.log 2>&1
python tools/rocprof_summary.py gpurun_out/prof_bn/bn_results.db /tmp/p.md > /dev/null 2>&1; grep -E "bn_|upsample|affine|gate_kernel|mix_|chan_sum" /tmp/p.md
