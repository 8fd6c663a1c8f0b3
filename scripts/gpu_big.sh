#!/bin/bash
# Scale checks on one MI355X (bounded by timeouts; each prints bench.py's JSON line):
#   1. BASELINE configs[3] per-GPU shard: 1.25M SNPs x 100k samples, int8 residency (125 GB)
#   2. north_star target shape on ONE GPU: 10M SNPs x 100k samples, 2-bit residency (251 GB)
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --snps 1250000 --samples 100000 --steps 2 --warmup 1 --no-cpu-baseline --no-second-path > gpurun_out/big_c4shard.json 2> gpurun_out/big_c4shard.err
echo "c4 shard rc=$?"; tail -2 gpurun_out/big_c4shard.err; cut -c1-600 gpurun_out/big_c4shard.json
timeout -k 10 500 python bench.py --snps 10000000 --samples 100000 --storage 2bit --steps 2 --warmup 1 --no-cpu-baseline --no-second-path > gpurun_out/big_10Mx100k.json 2> gpurun_out/big_10Mx100k.err
echo "10M x 100k rc=$?"; tail -2 gpurun_out/big_10Mx100k.err; cut -c1-600 gpurun_out/big_10Mx100k.json
