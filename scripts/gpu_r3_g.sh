#!/bin/bash
# GPU-box helper (round 3): the new short K1 rounds and the folded rank agreements first, then the whole -m gpu suite + smoke, then a bench line
tag=$1
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "short_dma or alternative_kernels or allreduce_hook or partition_independent" > gpurun_out/pytest_new_$tag.log 2>&1; rc=$?
tail -5 gpurun_out/pytest_new_$tag.log
if grep -q "Memory access fault" gpurun_out/pytest_new_$tag.log; then echo "GPU FAULT"; exit 1; fi
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -m pytest tests/test_gpu_stream.py tests/test_gpu_multi.py -m gpu -x -q -k "shard or rank or process or rccl" > gpurun_out/pytest_ranks_$tag.log 2>&1; rc=$?
tail -5 gpurun_out/pytest_ranks_$tag.log
[ $rc -ne 0 ] && exit $rc
( time timeout -k 10 400 python bench.py --no-extras --no-cpu-baseline ) > gpurun_out/bench_quick_$tag.json 2> gpurun_out/bench_quick_$tag.err || { tail -5 gpurun_out/bench_quick_$tag.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/bench_quick_$tag.json').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['all_kernels_ms_per_step'])
for k in ('f32_mfma_path','packed_2bit_residency','packed_2bit_four_planes'):
    if k in d: print(k, d[k]['ms_per_step'], d[k]['roofline']['all_kernels_ms_per_step'])"
python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/pytest_$tag.log 2>&1; rc=$?
tail -25 gpurun_out/pytest_$tag.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$tag.log 2>&1 || rc=1; tail -4 gpurun_out/smoke_$tag.log
if grep -q "Memory access fault" gpurun_out/pytest_$tag.log gpurun_out/smoke_$tag.log; then echo "GPU FAULT"; exit 1; fi
exit $rc
