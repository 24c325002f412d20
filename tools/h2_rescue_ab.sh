#!/bin/bash
# per-launch A/B of build switches of conv_h2.hip (csrc/variants/lib<name>.so from build.build_variant), same box, two
# alternating rounds.  usage: tools/h2_rescue_ab.sh name1 name2 ...   (CURRENT = the shipped library)
V=electrocardio_panorama_amd/csrc/variants
NAMES="${@:-prerescue CURRENT norescue noamax epialways}"
mkdir -p gpurun_out
for i in 1 2; do
  for n in $NAMES; do
    if [ $n = CURRENT ]; then unset NEF_LIB; else export NEF_LIB=$V/lib$n.so; fi
    FORMS=0 CHECK=0 python tools/h2p_check.py 2>&1 | grep "form 0" | awk '{print $(NF-1)}' > gpurun_out/rab_${n}_$i.col
  done
done
unset NEF_LIB
FORMS=0 CHECK=0 python tools/h2p_check.py 2>&1 | grep "form 0" | cut -c1-28 > gpurun_out/rab_names.col
echo "launch                       $NAMES | $NAMES"
F=""; for i in 1 2; do for n in $NAMES; do F="$F gpurun_out/rab_${n}_$i.col"; done; done
paste gpurun_out/rab_names.col $F
