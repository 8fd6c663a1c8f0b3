#!/bin/bash
# GPU-box helper: smoke(), bench under torch.distributed.run (one rank), and the resident rate at config 5's shape class.
tag=$1
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$tag.log 2>&1 || { tail -20 gpurun_out/smoke_$tag.log; exit 1; }
tail -8 gpurun_out/smoke_$tag.log
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-second-path > gpurun_out/bench_torchrun1_$tag.json 2> gpurun_out/bench_torchrun1_$tag.err || { tail -20 gpurun_out/bench_torchrun1_$tag.err; exit 1; }
tail -c 600 gpurun_out/bench_torchrun1_$tag.json; echo
timeout -k 10 400 python bench.py --precision i8 --storage 2bit --snps 1250000 --samples 500000 -k 40 --steps 1 --warmup 1 --no-second-path --no-cpu-baseline > gpurun_out/resident_1.25Mx500k_k40_2bit_$tag.json 2> gpurun_out/resident_1.25Mx500k_k40_2bit_$tag.err || tail -5 gpurun_out/resident_1.25Mx500k_k40_2bit_$tag.err
python - <<PY
import json
d=json.loads(open('gpurun_out/resident_1.25Mx500k_k40_2bit_$tag.json').read().strip().splitlines()[-1])
print('resident 1.25Mx500k k40 2bit', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'])
PY
