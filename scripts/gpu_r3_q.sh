#!/bin/bash
# GPU-box helper (round 3): the two-process shard test without torch, then the in-process A/B of staggered short rounds in k_gq_d
tag=$1
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_stream.py tests/test_gpu_multi.py -m gpu -x -q -k "two_process or two_gpus or sharded" --durations=4 > gpurun_out/pytest_2p_$tag.log 2>&1; rc=$?
tail -8 gpurun_out/pytest_2p_$tag.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python scripts/ab_env.py GPCA_GQ_SHORT=2 GPCA_GQ_SHORT=1 10 > gpurun_out/ab_stagger_$tag.log 2>&1 || { tail -5 gpurun_out/ab_stagger_$tag.log; exit 1; }
timeout -k 10 300 python scripts/ab_env.py GPCA_GQ_SHORT=1 GPCA_GQ_SHORT=2 10 >> gpurun_out/ab_stagger_$tag.log 2>&1 || { tail -5 gpurun_out/ab_stagger_$tag.log; exit 1; }
cat gpurun_out/ab_stagger_$tag.log
