#!/bin/bash
# Ablation of the packed K1 kernel (scripts/kbench/kbench_gq2.hip: the product kernel compiled with GPCA_ABLATE bits).
#   ABL="16 17 18 20 24" MODE=real|random WAVES=1024 bash scripts/gpu_gq2_ablate.sh <tag>
tag=${1:-x}
mkdir -p gpurun_out
log=gpurun_out/gq2_ablate_${MODE:-random}_$tag.log
: > $log
for a in ${ABL:-0 1 2 3 16 19}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DGPCA_ABLATE=$a -o /tmp/kb_$a scripts/kbench/kbench_gq2.hip 2> gpurun_out/gq2_ablate_build.err || { echo "build failed for $a"; tail -3 gpurun_out/gq2_ablate_build.err; exit 1; }
  timeout -k 10 60 /tmp/kb_$a ${WAVES:-1024} ${MODE:-random} >> $log 2>&1 || { echo "run failed for $a"; tail -3 $log; exit 1; }
  if grep -q "Memory access fault" $log; then echo "GPU FAULT in ablate $a"; exit 1; fi
done
cat $log
