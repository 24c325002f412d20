#!/bin/bash
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/electrocardio_panorama_amd/csrc/variants/$1
timeout 300 python -m pytest tests/test_pano_gpu.py -q -x -k "tail" 2>&1 | tail -n 3 | cut -c1-200
for rep in 1 2 3; do
for lib in default $V; do
  if [ $lib = default ]; then unset NEF_LIB; else export NEF_LIB=$lib; fi
  PANO=fp16 timeout 300 python tools/bench_sweep.py 2>/dev/null | tail -n 1 | cut -c1-60 | sed "s#^#lib=$(basename $lib) #"
done; done
