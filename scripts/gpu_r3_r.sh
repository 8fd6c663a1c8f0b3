#!/bin/bash
# GPU-box helper (round 3): K2 row-slice count (GPCA_GTT_WAVES) A/B on the headline shape, both orders, and 3072
tag=$1
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > gpurun_out/ab_gttwaves_$tag.log
for pair in "GPCA_GTT_WAVES=4096 GPCA_GTT_WAVES=2048" "GPCA_GTT_WAVES=2048 GPCA_GTT_WAVES=4096" "GPCA_GTT_WAVES=3072 GPCA_GTT_WAVES=2048" "GPCA_GTT_WAVES=2048 GPCA_GTT_WAVES=6144"; do
  timeout -k 10 300 python scripts/ab_env.py $pair 8 >> gpurun_out/ab_gttwaves_$tag.log 2>&1 || { tail -5 gpurun_out/ab_gttwaves_$tag.log; exit 1; }
done
cat gpurun_out/ab_gttwaves_$tag.log
