#!/bin/bash
# A/B on one box: k_gq_d with chained DMA rounds (default) vs a cold prologue per round (GPCA_GQ_CHAIN=0)
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for c in 1 0; do
  GPCA_GQ_CHAIN=$c timeout -k 10 200 python bench.py --no-cpu-baseline --no-second-path --steps 10 --warmup 2 > gpurun_out/ab_chain${c}_$rep.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('gpurun_out/ab_chain${c}_$rep.json').read().strip().splitlines()[-1])
t=d['roofline']['all_kernels_ms_per_step']
print('chain=$c rep=$rep ms/step %.3f  K1 %.4f ms/launch  K2 %.4f ms/launch' % (d['ms_per_step'], t['gemm_GQ']/3, t['gemm_GtT']/3))
PY
done; done
