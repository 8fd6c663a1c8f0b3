#!/bin/bash
# GPU-box helper (round 3): buffer placement with physically contiguous allocations, packed-K1 ablation on real-like and random operands,
# the concurrent pull API + packed parity after the 3-tile tail group, a quick bench line
tag=$1
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/kbench_place scripts/kbench/kbench_place.hip 2> gpurun_out/place_build_$tag.err || { tail -3 gpurun_out/place_build_$tag.err; exit 1; }
timeout -k 10 240 /tmp/kbench_place contig > gpurun_out/place_contig_$tag.log 2>&1; tail -24 gpurun_out/place_contig_$tag.log
if grep -q "Memory access fault" gpurun_out/place_contig_$tag.log; then echo "GPU FAULT"; exit 1; fi
ABL="16 17 18 20 24" MODE=real bash scripts/gpu_gq2_ablate.sh $tag || exit 1
ABL="16 19" MODE=random bash scripts/gpu_gq2_ablate.sh $tag || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_parity.py tests/test_abi.py -m gpu -x -q -k "shared_between_threads or standardize_block or 2bit or c4_per_gpu or full_size or random_shapes" > gpurun_out/pytest_pull_$tag.log 2>&1; rc=$?
tail -5 gpurun_out/pytest_pull_$tag.log
if grep -q "Memory access fault" gpurun_out/pytest_pull_$tag.log; then echo "GPU FAULT"; exit 1; fi
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py --no-extras --no-cpu-baseline > gpurun_out/bench_quick_$tag.json 2> gpurun_out/bench_quick_$tag.err || { tail -5 gpurun_out/bench_quick_$tag.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/bench_quick_$tag.json').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline'].get('frac_of_achievable'), d['roofline']['all_kernels_ms_per_step'])
for k in ('f32_mfma_path','packed_2bit_residency','packed_2bit_four_planes'):
    if k in d: print(k, d[k]['ms_per_step'], d[k]['roofline']['all_kernels_ms_per_step'])"
