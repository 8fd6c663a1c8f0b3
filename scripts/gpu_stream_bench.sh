#!/bin/bash
# GPU-box helper: out-of-core (streamed-panel) bench lines.  usage: gpu_stream_bench.sh <tag>
tag=$1
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
# BASELINE.json configs[4]: the per-GPU shard of 50M x 500k, k = 40, panels from the device generator
timeout -k 10 400 python bench.py --streamed --snps 6250000 --samples 500000 -k 40 --storage 2bit --steps 1 --warmup 0 > gpurun_out/stream_c5_2bit_$tag.json 2> gpurun_out/stream_c5_2bit_$tag.err || tail -5 gpurun_out/stream_c5_2bit_$tag.err
echo "--- c5 2bit"; cat gpurun_out/stream_c5_2bit_$tag.json
timeout -k 10 400 python bench.py --streamed --snps 6250000 --samples 500000 -k 40 --storage int8 --steps 1 --warmup 0 > gpurun_out/stream_c5_int8_$tag.json 2> gpurun_out/stream_c5_int8_$tag.err || tail -5 gpurun_out/stream_c5_int8_$tag.err
echo "--- c5 int8"; cat gpurun_out/stream_c5_int8_$tag.json
# configs[1] shape streamed next to resident
timeout -k 10 300 python bench.py --no-cpu-baseline --no-second-path --streamed-extra > gpurun_out/stream_c2_$tag.json 2> gpurun_out/stream_c2_$tag.err || tail -5 gpurun_out/stream_c2_$tag.err
echo "--- c2"; cat gpurun_out/stream_c2_$tag.json
