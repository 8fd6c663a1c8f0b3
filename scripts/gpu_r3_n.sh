#!/bin/bash
# GPU-box helper (round 3): bench.py the way the driver launches ranks (torch.distributed.run, here with one rank: same code path as N > 1)
tag=$1
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 > gpurun_out/bench_torchrun_$tag.json 2> gpurun_out/bench_torchrun_$tag.err; rc=$?
tail -3 gpurun_out/bench_torchrun_$tag.err
python -c "
import json; d=json.loads([l for l in open('gpurun_out/bench_torchrun_$tag.json') if l.startswith('{')][-1])
print('ms_per_step', d['ms_per_step'], 'n_gpus', d['n_gpus'], 'multi_gpu', d.get('multi_gpu'), 'keys', [k for k in d if isinstance(d[k], dict)])"
exit $rc
