#!/bin/bash
# GPU-box helper (round 3): what is wrong with hipDeviceMallocContiguous memory under the product (probe + diagnosis), the A/B of it per handle,
# then the whole suite on the default (plain allocations)
tag=$1
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_contig scripts/kbench/probe_contig.hip 2> gpurun_out/probe_build_$tag.err || { tail -3 gpurun_out/probe_build_$tag.err; exit 1; }
timeout -k 10 120 /tmp/probe_contig > gpurun_out/probe_contig_$tag.log 2>&1; cat gpurun_out/probe_contig_$tag.log
if grep -q "Memory access fault" gpurun_out/probe_contig_$tag.log; then echo "GPU FAULT"; exit 1; fi
timeout -k 10 200 python scripts/diag_contig.py > gpurun_out/diag_contig_$tag.log 2>&1; cat gpurun_out/diag_contig_$tag.log
if grep -q "Memory access fault" gpurun_out/diag_contig_$tag.log; then echo "GPU FAULT"; exit 1; fi
timeout -k 10 300 python scripts/ab_env.py GPCA_CONTIG=1 GPCA_CONTIG=0 8 > gpurun_out/ab_contig_$tag.log 2>&1
timeout -k 10 300 python scripts/ab_env.py GPCA_CONTIG=0 GPCA_CONTIG=1 8 >> gpurun_out/ab_contig_$tag.log 2>&1
grep -v "^Traceback\|^  File" gpurun_out/ab_contig_$tag.log | tail -8
python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/pytest_$tag.log 2>&1; rc=$?
tail -16 gpurun_out/pytest_$tag.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$tag.log 2>&1 || rc=1; tail -3 gpurun_out/smoke_$tag.log
if grep -q "Memory access fault" gpurun_out/pytest_$tag.log gpurun_out/smoke_$tag.log; then echo "GPU FAULT"; exit 1; fi
exit $rc
