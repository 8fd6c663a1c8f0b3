#!/bin/bash
# GPU-box helper (round 3): in-process A/B of physically contiguous genotype storage (both orders), packed-K1 workgroup spans, whole suite
tag=$1
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python scripts/ab_env.py GPCA_CONTIG=1 GPCA_CONTIG=0 10 > gpurun_out/ab_contig_$tag.log 2>&1 || { tail -5 gpurun_out/ab_contig_$tag.log; exit 1; }
timeout -k 10 300 python scripts/ab_env.py GPCA_CONTIG=0 GPCA_CONTIG=1 10 >> gpurun_out/ab_contig_$tag.log 2>&1 || { tail -5 gpurun_out/ab_contig_$tag.log; exit 1; }
cat gpurun_out/ab_contig_$tag.log
ABL="16 18" MODE=real bash scripts/gpu_gq2_ablate.sh $tag || exit 1
python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/pytest_$tag.log 2>&1; rc=$?
tail -16 gpurun_out/pytest_$tag.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$tag.log 2>&1 || rc=1; tail -3 gpurun_out/smoke_$tag.log
if grep -q "Memory access fault" gpurun_out/pytest_$tag.log gpurun_out/smoke_$tag.log; then echo "GPU FAULT"; exit 1; fi
exit $rc
