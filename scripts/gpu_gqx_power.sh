#!/bin/bash
set -e
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -o /tmp/kb_gqx scripts/kbench/kbench_gqx.hip 2>/dev/null
timeout -k 10 120 /tmp/kb_gqx | tee gpurun_out/gqx_power.log
