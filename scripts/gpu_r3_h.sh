#!/bin/bash
# GPU-box helper (round 3): in-process A/B of k_gq_d's short rounds (both orders), the whole -m gpu suite without -x, smoke
tag=$1
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python scripts/ab_env.py GPCA_GQ_SHORT=1 GPCA_GQ_SHORT=0 10 > gpurun_out/ab_short_$tag.log 2>&1 || { tail -5 gpurun_out/ab_short_$tag.log; exit 1; }
timeout -k 10 300 python scripts/ab_env.py GPCA_GQ_SHORT=0 GPCA_GQ_SHORT=1 10 >> gpurun_out/ab_short_$tag.log 2>&1 || { tail -5 gpurun_out/ab_short_$tag.log; exit 1; }
cat gpurun_out/ab_short_$tag.log
python -m pytest tests -m gpu -q --durations=12 > gpurun_out/pytest_$tag.log 2>&1; rc=$?
tail -30 gpurun_out/pytest_$tag.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$tag.log 2>&1 || rc=1; tail -4 gpurun_out/smoke_$tag.log
if grep -q "Memory access fault" gpurun_out/pytest_$tag.log gpurun_out/smoke_$tag.log; then echo "GPU FAULT"; exit 1; fi
exit $rc
