#!/bin/bash
# GPU-box helper (round 3): the concurrent pull API, then the round's measured artefacts (scripts/gpu_profile.sh) with the default bench timed
tag=$1
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_stream.py tests/test_gpu_parity.py tests/test_abi.py -m gpu -x -q -k "shared_between_threads or standardize_block or abi or c_client or no_device_memory" > gpurun_out/pytest_pull_$tag.log 2>&1; rc=$?
tail -5 gpurun_out/pytest_pull_$tag.log
[ $rc -ne 0 ] && exit $rc
SECONDS=0
bash scripts/gpu_profile.sh $tag
echo "gpu_profile.sh took $SECONDS s"
