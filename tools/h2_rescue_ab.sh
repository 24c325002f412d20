#!/bin/bash
# per-launch A/B of the range-rescue build switches of conv_h2.hip (csrc/variants/lib*.so from build.build_variant), same box,
# two alternating rounds: pre-rescue | current | RESCUE=0 | RESCUE=0 AMAX=0 | EPI_ALWAYS=1
V=electrocardio_panorama_amd/csrc/variants
mkdir -p gpurun_out
for i in 1 2; do
  for n in prerescue CURRENT norescue noamax epialways; do
    if [ $n = CURRENT ]; then unset NEF_LIB; else export NEF_LIB=$V/lib$n.so; fi
    FORMS=0 CHECK=0 python tools/h2p_check.py 2>&1 | grep "form 0" | awk '{print $(NF-1)}' > gpurun_out/rab_${n}_$i.col
  done
done
unset NEF_LIB
FORMS=0 CHECK=0 python tools/h2p_check.py 2>&1 | grep "form 0" | cut -c1-28 > gpurun_out/rab_names.col
echo "launch                       prerescue current norescue noamax epialways (x2)"
paste gpurun_out/rab_names.col gpurun_out/rab_{prerescue,CURRENT,norescue,noamax,epialways}_1.col gpurun_out/rab_{prerescue,CURRENT,norescue,noamax,epialways}_2.col
