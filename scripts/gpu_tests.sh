#!/bin/bash
# GPU-box helper: the -m gpu suite with its log kept under gpurun_out/.  usage: gpu_tests.sh <tag> [pytest args]
tag=$1; shift
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=15 "$@" > gpurun_out/pytest_$tag.log 2>&1
rc=$?
tail -40 gpurun_out/pytest_$tag.log
exit $rc
