#!/bin/bash
# GPU-box helper (round 2): Infinity-Cache pairing probe, config-3 timing, streamed config-5 lines.  usage: gpu_misc_r2.sh <tag>
tag=$1
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/kbench_stream scripts/kbench/kbench_stream.hip && timeout -k 10 240 /tmp/kbench_stream > gpurun_out/stream_$tag.log 2>&1
tail -12 gpurun_out/stream_$tag.log
timeout -k 10 300 python scripts/bench_config3.py > gpurun_out/config3_$tag.json 2> gpurun_out/config3_$tag.err || tail -5 gpurun_out/config3_$tag.err
cat gpurun_out/config3_$tag.json
timeout -k 10 400 python bench.py --streamed --snps 6250000 --samples 500000 -k 40 --storage 2bit --steps 1 --warmup 0 > gpurun_out/stream_c5_2bit_$tag.json 2> gpurun_out/stream_c5_2bit_$tag.err || tail -5 gpurun_out/stream_c5_2bit_$tag.err
cat gpurun_out/stream_c5_2bit_$tag.json
timeout -k 10 400 python bench.py --streamed --snps 6250000 --samples 500000 -k 40 --storage int8 --ring 2 --steps 1 --warmup 0 > gpurun_out/stream_c5_int8_$tag.json 2> gpurun_out/stream_c5_int8_$tag.err || tail -5 gpurun_out/stream_c5_int8_$tag.err
cat gpurun_out/stream_c5_int8_$tag.json
