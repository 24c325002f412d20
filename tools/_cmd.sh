cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_pano_gpu.py -q -x 2>&1 | tail -1
PANO=fp16 PAIR_BUDGET=16384 timeout 300 python tools/bench_sweep.py 2>&1 | tail -1
PANO=fp16 PAIR_BUDGET=4096 timeout 300 python tools/bench_sweep.py 2>&1 | tail -1
PANO=fp16 PAIR_BUDGET=65536 timeout 300 python tools/bench_sweep.py 2>&1 | tail -1
PANO=fp16 timeout 300 python tools/bench_gen.py 2>&1 | tail -1
mkdir -p gpurun_out/prof_pano
PANO=fp16 PAIR_BUDGET=16384 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_pano -o pano -- python tools/bench_sweep.py > gpurun_out/prof_pano.log 2>&1
