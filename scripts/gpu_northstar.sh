#!/bin/bash
# GPU-box helper: north_star's literal configuration -- MFMA-fp32 GEMMs, 10M SNPs x 100k samples, k = 20, ONE MI355X
# (the matrix is 250 GB as 2-bit codes).  usage: gpu_northstar.sh <tag>
tag=$1
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python bench.py --precision f32 --storage 2bit --snps 10000000 --samples 100000 --steps 1 --warmup 0 --no-cpu-baseline --no-second-path \
  > gpurun_out/northstar_f32_2bit_$tag.json 2> gpurun_out/northstar_f32_2bit_$tag.err || tail -5 gpurun_out/northstar_f32_2bit_$tag.err
cat gpurun_out/northstar_f32_2bit_$tag.json
