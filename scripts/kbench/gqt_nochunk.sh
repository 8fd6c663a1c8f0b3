#!/bin/bash
# the drip kernel (gqt_drip.inc) with and without its epilogue chunks: what do the three-unit rest stages cost by themselves?
set -e
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DGPCA_STAMP=1 -o /tmp/kb_a scripts/kbench/kbench_gqd.hip 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DGPCA_STAMP=1 -DGQT_NOCHUNK=1 -o /tmp/kb_b scripts/kbench/kbench_gqd.hip 2>/dev/null
for bin in kb_a kb_b kb_a kb_b; do
  echo "== $bin (kb_b: chunks skipped)"
  /tmp/$bin ab 8000000 1000 3 0:0:1:0 0:0:1:100 | grep "mean"
  /tmp/$bin ab 3993600 2504 3 0:2:1:0 0:2:1:100 | grep "mean"
done
