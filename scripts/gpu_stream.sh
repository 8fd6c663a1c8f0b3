#!/bin/bash
# HBM streaming-read ceiling probe (scripts/kbench/kbench_stream.hip)
set -e
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -o /tmp/kbench_stream scripts/kbench/kbench_stream.hip
timeout -k 10 240 /tmp/kbench_stream | tee gpurun_out/stream.log
