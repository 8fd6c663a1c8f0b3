#!/bin/bash
# GPU-box helper: configs[2] shape (1 066 557 x 64) -- stage times and a kernel trace of the default path.  usage: gpu_config3.sh <tag>
tag=$1
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python scripts/bench_config3.py > gpurun_out/config3_$tag.json 2> gpurun_out/config3_$tag.err || { tail -5 gpurun_out/config3_$tag.err; exit 1; }
cat gpurun_out/config3_$tag.json
export TMPDIR=/tmp CFG3_ONLY=i8/int8
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_config3_$tag -o c3 --output-format csv -- python3 scripts/bench_config3.py > gpurun_out/config3_prof_$tag.log 2>&1 || { tail -5 gpurun_out/config3_prof_$tag.log; exit 1; }
f=$(find gpurun_out/prof_config3_$tag -name '*kernel_stats.csv' | head -1)
head -40 "$f"
