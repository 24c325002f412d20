#!/bin/bash
# Timing-only variants of csrc/conv_h2p.hip (results wrong): build with
#   python -c "from electrocardio_panorama_amd.csrc import build as b; [b.build_variant('h2p_t%d' % t, ['NEF_H2P_T=%d' % t], sources=('conv_h2p.hip',)) for t in (4, 5, 6, 13, 61, 68, 70)]"
# then `VARS="full t4 t13 ..." tools/h2p_variants.sh [name filter]` prints, per variant, the time of the producer / consumer form per shape
# (NEF_H2P_T bits: see the top of conv_h2p.hip).
for v in $VARS; do
  if [ "$v" = full ]; then lib=""; else lib="electrocardio_panorama_amd/csrc/variants/libh2p_$v.so"; fi
  echo "=== variant $v"
  NEF_LIB=$lib CHECK=0 ITERS=10 WARM=10 timeout 200 python tools/h2p_check.py "$1" 2>&1 | grep -v amdgpu.ids
done
