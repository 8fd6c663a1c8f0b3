#!/bin/bash
# ablation of the packed K1 kernel (scripts/kbench/kbench_gq2.hip)
set -e
mkdir -p gpurun_out
: > gpurun_out/gq2_ablate.log
for a in ${ABL:-0 1 2 3 16 19}; do
  hipcc --offload-arch=gfx950 -O3 -DGPCA_ABLATE=$a -o /tmp/kb_$a scripts/kbench/kbench_gq2.hip 2>/dev/null
  timeout -k 10 60 /tmp/kb_$a ${WAVES:-1024} ${MODE:-random} | tee -a gpurun_out/gq2_ablate.log
done
