#!/bin/bash
# GPU-box helper (round 3): the whole -m gpu suite + smoke, then the round's measured artefacts (scripts/gpu_profile.sh)
tag=$1
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/pytest_$tag.log 2>&1; rc=$?
tail -14 gpurun_out/pytest_$tag.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$tag.log 2>&1 || rc=1; tail -3 gpurun_out/smoke_$tag.log
if grep -q "Memory access fault" gpurun_out/pytest_$tag.log gpurun_out/smoke_$tag.log; then echo "GPU FAULT"; exit 1; fi
[ $rc -ne 0 ] && exit $rc
SECONDS=0
bash scripts/gpu_profile.sh $tag > gpurun_out/profile_$tag.out 2>&1; tail -14 gpurun_out/profile_$tag.out
echo "gpu_profile.sh took $SECONDS s"
