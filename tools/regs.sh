#!/bin/bash
# VGPR / SGPR / occupancy / scratch per kernel:  tools/regs.sh <file.hip> <grep pattern> [-DFLAG ...]
F=$1; PAT=$2; shift 2
cd /root/repo/electrocardio_panorama_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I ../../include -I . "$@" -c $F -o /tmp/regs_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 | \
 awk '/Function Name/ {name=$0} /VGPRs:/ {v=$NF} /AGPRs/ {a=$NF} /ScratchSize/ {s=$NF} /Occupancy/ {o=$NF} /LDS Size/ {print name " VGPR=" v " AGPR=" a " scratch=" s " occ=" o " lds=" $NF}' | sed 's/.*Function Name: //' | grep -E "$PAT" | while read l; do n=$(echo $l | cut -d' ' -f1); echo "$(echo $n | /opt/rocm/lib/llvm/bin/llvm-cxxfilt | cut -c1-90) $(echo $l | cut -d' ' -f2-)"; done
rm -f /tmp/regs_$$.o
