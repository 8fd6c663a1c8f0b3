#!/bin/bash
# GPU-box helper (round 3): the shape sweep of DESIGN.md section 5 again with the round-3 kernels (both residencies)
tag=$1
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python scripts/shape_sweep.py int8 > gpurun_out/shape_sweep_$tag.jsonl 2> gpurun_out/shape_sweep_$tag.err || { tail -5 gpurun_out/shape_sweep_$tag.err; exit 1; }
timeout -k 10 500 python scripts/shape_sweep.py 2bit >> gpurun_out/shape_sweep_$tag.jsonl 2>> gpurun_out/shape_sweep_$tag.err || { tail -5 gpurun_out/shape_sweep_$tag.err; exit 1; }
cat gpurun_out/shape_sweep_$tag.jsonl
