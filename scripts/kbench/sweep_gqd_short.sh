#!/bin/bash
# int8 K1: k_gq_d (chained rounds) against the two round-5 experiments, interleaved in one process per shape, outputs compared bit by bit
# (kbench_gqd ab; setting = A:B:chain:skew -- skew 1..9 = k_gq_s, skewed tile rounds (gqs_skew.inc); skew 100 = k_gq_t, the epilogue
# dripped between the other tiles' units (gqt_drip.inc)).  usage: scripts/gpu.sh <tag> sh=scripts/kbench/sweep_gqd_short.sh
set -e
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DGPCA_STAMP=1 -o /tmp/kbench_gqd_ab scripts/kbench/kbench_gqd.hip 2>/dev/null
/tmp/kbench_gqd_ab ab 9600 1000 2 0:0:1:0 0:0:1:100
/tmp/kbench_gqd_ab ab 100064 2504 2 0:2:1:0 0:2:1:100
/tmp/kbench_gqd_ab ab 8000000 1000 4 0:0:1:0 0:0:1:1 0:0:1:100
/tmp/kbench_gqd_ab ab 3993600 2504 4 0:2:1:0 0:2:1:1 0:2:1:100
/tmp/kbench_gqd_ab ab 2000000 5000 4 0:5:1:0 0:5:1:100
/tmp/kbench_gqd_ab ab 1000064 10000 4 5:1:1:0 5:1:1:1 5:1:1:100
/tmp/kbench_gqd_ab ab 99968 100000 4 0:0:1:0 0:0:1:100
