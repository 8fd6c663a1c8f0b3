#!/bin/bash
# A/B of GEMM grid targets for both paths: gpu_ab.sh "<gq> <gtt> <precision>" ...
mkdir -p gpurun_out
for cfg in "$@"; do
  set -- $cfg
  GPCA_GQ_WAVES=$1 GPCA_GTT_WAVES=$2 timeout -k 10 200 python bench.py --precision $3 $4 $5 $6 $7 --steps ${STEPS:-5} --warmup 1 --no-cpu-baseline --no-second-path > gpurun_out/ab.json 2> gpurun_out/ab.err || tail -3 gpurun_out/ab.err
  python - "$cfg" <<'PY'
import json, sys
d = json.load(open('gpurun_out/ab.json'))
print(sys.argv[1], round(d['value'] / 1e9, 1), 'G/s', round(d['ms_per_step'], 2), 'ms', {k: round(v, 2) for k, v in d['roofline']['all_kernels_ms_per_step'].items()})
PY
done
