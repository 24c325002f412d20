#!/bin/bash
# same-box A/B of NEF_H2_XORDER (A fragments issued in front of the next stage's activation rows): per-launch and whole step
cd $GRAFT_REPO_ROOT
V=electrocardio_panorama_amd/csrc/variants/libxorder0.so
for rep in 1 2; do
for lib in default $V; do
  if [ $lib = default ]; then unset NEF_LIB; else export NEF_LIB=$GRAFT_REPO_ROOT/$lib; fi
  echo "== lib=$lib rep=$rep"
  ITERS=20 ONLY_WHAT=wino python tools/bench_conv.py 2>/dev/null | grep -v "roi"
done; done
for rep in 1 2 3; do
for lib in default $V; do
  if [ $lib = default ]; then unset NEF_LIB; else export NEF_LIB=$GRAFT_REPO_ROOT/$lib; fi
  python bench.py --no-cpu-baseline --no-kernel-events --no-secondary --steps 30 --warmup 5 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step lib=$lib', d['ms_per_step'])"
done; done
