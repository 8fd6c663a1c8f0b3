#!/bin/bash
# The device eigen-solver's Jacobi step, ablated (scripts/kbench/kbench_eig.hip built with -DGPCA_EIG_ABL=<bits>; results are wrong on
# purpose, every variant runs exactly 8 sweeps): what the parts of a step cost.
for abl in 16 17 18 20 23; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DGPCA_EIG_STAMP=1 -DGPCA_EIG_ABL=$abl -o /tmp/kbench_eig_$abl scripts/kbench/kbench_eig.hip 2>/dev/null || exit 1
  echo "== GPCA_EIG_ABL=$abl (16 = the full step, 8 sweeps)"; timeout -k 5 60 /tmp/kbench_eig_$abl | grep "n =  30 (L =  32, 16\|n =  64\|n = 128" | sed 's/| w0.*//'
done
