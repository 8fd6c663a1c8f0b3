#!/bin/bash
# phase sweep of k_gq_d on the GPU box: workgroup b starts its sweep at stage ((b % 8) * A + (b / 8) * B) mod stages.  usage: sweep_gqd_phase.sh M N "A:B ..."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DGPCA_STAMP=1 -o /tmp/kbench_gqd scripts/kbench/kbench_gqd.hip 2>/dev/null || exit 1
for ab in $3; do
  a=${ab%%:*}; b=${ab##*:}
  /tmp/kbench_gqd $1 $2 1 0 1 $a $b | head -2 | tr '\n' ' ' | sed 's/per workgroup; first start -> last start [0-9.]* us,//'; echo
done
