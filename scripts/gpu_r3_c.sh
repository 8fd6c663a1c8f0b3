#!/bin/bash
# GPU-box helper (round 3): narrow-kernel parity + config-3 timing, 3-plane accuracy table, packed-K1 ablation on realistic operands
tag=$1
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "narrow or degenerate or rsvd_parity or wide_sketch or bitwise_repeatable" > gpurun_out/pytest_$tag.log 2>&1
tail -5 gpurun_out/pytest_$tag.log
timeout -k 10 300 python scripts/bench_config3.py > gpurun_out/config3_$tag.json 2> gpurun_out/config3_$tag.err; tail -3 gpurun_out/config3_$tag.err
python -c "
import json; d=json.load(open('gpurun_out/config3_$tag.json'))
for k,v in d.items():
    if isinstance(v,dict): print(k, 'rsvd_ms', round(v['rsvd_ms'],3), v['gemm_launch_us'], v['stages_us_per_call'])"
timeout -k 10 300 python scripts/planes3_parity.py > gpurun_out/planes3_$tag.json 2> gpurun_out/planes3_$tag.err; tail -3 gpurun_out/planes3_$tag.err
python -c "
import json; d=json.load(open('gpurun_out/planes3_$tag.json'))
for k,v in d.items(): print(k, {kk:{a:float('%.2e'%b) for a,b in vv.items()} for kk,vv in v.items()})"
ABL="16 17 18 20 22 23 24" MODE=real bash scripts/gpu_gq2_ablate.sh > /dev/null 2>&1; cp gpurun_out/gq2_ablate.log gpurun_out/gq2_ablate_real_$tag.log; cat gpurun_out/gq2_ablate_real_$tag.log
ABL="16 19" MODE=random bash scripts/gpu_gq2_ablate.sh > /dev/null 2>&1; cp gpurun_out/gq2_ablate.log gpurun_out/gq2_ablate_random_$tag.log; cat gpurun_out/gq2_ablate_random_$tag.log
