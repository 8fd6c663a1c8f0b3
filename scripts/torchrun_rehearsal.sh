#!/bin/bash
# bench.py --gpus 2 started the way the driver starts it (python -m torch.distributed.run), both ranks on device 0 over the host-staged
# exchange: the rendezvous under real torchrun variables (rank 0 hosts the hub on a private filesystem socket).  One JSON line on success.
# usage: scripts/gpu.sh <tag> sh=scripts/torchrun_rehearsal.sh
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --one-device --exchange host --steps 2 --warmup 1 --snps 200000 --samples 20000 > gpurun_out/torchrun_rehearsal.json 2> gpurun_out/torchrun_rehearsal.err
rc=$?
tail -3 gpurun_out/torchrun_rehearsal.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/torchrun_rehearsal.json").read().strip().splitlines()[-1])
m = d.get("multi_gpu", {})
print("n_gpus", d["n_gpus"], "ms_per_step", round(d["ms_per_step"], 2), "value %.3e" % d["value"], "weak_scaling_efficiency", m.get("weak_scaling_efficiency"), "vs_slowest", m.get("weak_scaling_efficiency_vs_slowest_rank"))
PY
exit $rc
