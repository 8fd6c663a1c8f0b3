#!/bin/bash
# GPU-box helper (round 3): narrow-kernel parity + config-3 timing, 3-plane accuracy table, packed-K1 ablation on realistic operands
tag=$1
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py -m gpu -x -q --deselect tests/test_gpu_parity.py::test_c4_per_gpu_shard_i8_and_2bit --deselect tests/test_gpu_stream.py::test_config5_per_gpu_shard_streamed > gpurun_out/pytest_$tag.log 2>&1
tail -5 gpurun_out/pytest_$tag.log
timeout -k 10 300 python scripts/bench_config3.py > gpurun_out/config3_$tag.json 2> gpurun_out/config3_$tag.err; tail -3 gpurun_out/config3_$tag.err
python -c "
import json; d=json.load(open('gpurun_out/config3_$tag.json'))
for k,v in d.items():
    if isinstance(v,dict): print(k, 'rsvd_ms', round(v['rsvd_ms'],3), v['gemm_launch_us'], v['stages_us_per_call'])"
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || tail -5 gpurun_out/bench_$tag.err
python -c "
import json; d=json.load(open('gpurun_out/bench_$tag.json')); print('ms_per_step', d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'], [d[k]['ms_per_step'] for k in ('f32_mfma_path','packed_2bit_residency','packed_2bit_four_planes')])"
timeout -k 10 300 python scripts/planes3_parity.py > gpurun_out/planes3_$tag.json 2> gpurun_out/planes3_$tag.err; tail -3 gpurun_out/planes3_$tag.err
python -c "
import json; d=json.load(open('gpurun_out/planes3_$tag.json'))
for k,v in d.items(): print(k, {kk:{a:float('%.2e'%b) for a,b in vv.items()} for kk,vv in v.items()})"
ABL="16 17 18 20 22 23 24" MODE=real bash scripts/gpu_gq2_ablate.sh > /dev/null 2>&1; cp gpurun_out/gq2_ablate.log gpurun_out/gq2_ablate_real_$tag.log; cat gpurun_out/gq2_ablate_real_$tag.log
ABL="16 19" MODE=random bash scripts/gpu_gq2_ablate.sh > /dev/null 2>&1; cp gpurun_out/gq2_ablate.log gpurun_out/gq2_ablate_random_$tag.log; cat gpurun_out/gq2_ablate_random_$tag.log
