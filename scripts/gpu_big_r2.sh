#!/bin/bash
# GPU-box helper (round 2): BASELINE.json configs[3]'s per-GPU shard (1.25M x 100k int8, 125 GB) and the whole 10M x 100k matrix on ONE
# GPU as 2-bit codes, exact path.  usage: gpu_big_r2.sh <tag>
tag=$1
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout -k 10 500 python bench.py --snps 1250000 --samples 100000 --steps 3 --warmup 1 --no-cpu-baseline --no-second-path > gpurun_out/c4shard_int8_$tag.json 2> gpurun_out/c4shard_int8_$tag.err || tail -5 gpurun_out/c4shard_int8_$tag.err
python - <<PY
import json
d=json.loads(open('gpurun_out/c4shard_int8_$tag.json').read().strip().splitlines()[-1])
print('c4 shard int8', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['all_kernels_ms_per_step'])
PY
timeout -k 10 600 python bench.py --snps 10000000 --samples 100000 --storage 2bit --steps 2 --warmup 1 --no-cpu-baseline --no-second-path > gpurun_out/one_gpu_10Mx100k_2bit_$tag.json 2> gpurun_out/one_gpu_10Mx100k_2bit_$tag.err || tail -5 gpurun_out/one_gpu_10Mx100k_2bit_$tag.err
python - <<PY
import json
d=json.loads(open('gpurun_out/one_gpu_10Mx100k_2bit_$tag.json').read().strip().splitlines()[-1])
print('10Mx100k 2bit', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['all_kernels_ms_per_step'])
PY
