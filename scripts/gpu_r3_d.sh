#!/bin/bash
# GPU-box helper (round 3): the whole -m gpu suite with durations, config-3 timing, packed-K1 ablation on realistic operands
tag=$1
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/pytest_$tag.log 2>&1
tail -25 gpurun_out/pytest_$tag.log
timeout -k 10 300 python scripts/bench_config3.py > gpurun_out/config3_$tag.json 2> gpurun_out/config3_$tag.err; tail -3 gpurun_out/config3_$tag.err
python -c "
import json; d=json.load(open('gpurun_out/config3_$tag.json'))
for k,v in d.items():
    if isinstance(v,dict): print(k, 'rsvd_ms', round(v['rsvd_ms'],3), v['gemm_launch_us'], v['stages_us_per_call'])"
for mode in real random; do
  : > gpurun_out/gq2_ablate_${mode}_$tag.log
  for a in 16 17 18 20 22 23 24; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DGPCA_ABLATE=$a -o /tmp/kb_$a scripts/kbench/kbench_gq2.hip 2>> gpurun_out/gq2_ablate_build_$tag.err || continue
    timeout -k 10 60 /tmp/kb_$a 1024 $mode >> gpurun_out/gq2_ablate_${mode}_$tag.log 2>&1
  done
  cat gpurun_out/gq2_ablate_${mode}_$tag.log
done
tail -5 gpurun_out/gq2_ablate_build_$tag.err
