#!/bin/bash
# The device eigen-solver's QL sweep, ablated (scripts/kbench/kbench_eig.hip built with -DGPCA_EIG_ABL=<bits>; results are wrong on purpose,
# only the tql2 stamp is read): what a rotation's parts cost one lone wave.
for abl in 0 1 2 4 8 3 7 15; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DGPCA_EIG_STAMP=1 -DGPCA_EIG_ABL=$abl -o /tmp/kbench_eig_$abl scripts/kbench/kbench_eig.hip 2>/dev/null || exit 1
  echo "== GPCA_EIG_ABL=$abl"; timeout -k 5 60 /tmp/kbench_eig_$abl | grep "n =  30 (L =  32, 16\|n = 128" | sed 's/| w0.*//'
done
