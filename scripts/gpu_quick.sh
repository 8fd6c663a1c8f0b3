#!/bin/bash
# GPU-box helper: streaming/generator tests, default bench line, config-5 streamed line.  usage: gpu_quick.sh <tag>
tag=$1
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_stream.py -m gpu -x -q > gpurun_out/pytest_stream_$tag.log 2>&1 || { tail -30 gpurun_out/pytest_stream_$tag.log; exit 1; }
tail -3 gpurun_out/pytest_stream_$tag.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || tail -5 gpurun_out/bench_$tag.err
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_$tag.json').read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['all_kernels_ms_per_step'])
for k in ('f32_mfma_path','packed_2bit_residency','packed_2bit_four_planes'):
    print(k, d[k]['ms_per_step'])
PY
timeout -k 10 400 python bench.py --streamed --snps 6250000 --samples 500000 -k 40 --storage 2bit --steps 1 --warmup 0 > gpurun_out/stream_c5_2bit_$tag.json 2> gpurun_out/stream_c5_2bit_$tag.err || tail -5 gpurun_out/stream_c5_2bit_$tag.err
python - <<PY
import json
d=json.loads(open('gpurun_out/stream_c5_2bit_$tag.json').read().strip().splitlines()[-1])
print('c5 2bit', d['value'], d['ms_per_step'], d['streaming'])
PY
