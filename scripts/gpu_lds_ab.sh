#!/bin/bash
# parity tests of the exact paths, then A/B of GPCA_LDS_PLANES for both residencies
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "i8 or 2bit or wide" 2>&1 | tail -3
for cfg in "1 int8" "0 int8" "1 2bit" "0 2bit"; do
  set -- $cfg
  GPCA_LDS_PLANES=$1 timeout -k 10 200 python bench.py --precision i8 --storage $2 --steps 5 --warmup 1 --no-cpu-baseline --no-second-path > gpurun_out/ab.json 2> gpurun_out/ab.err || tail -3 gpurun_out/ab.err
  python - "lds=$1 $2" <<'PY'
import json, sys
d = json.load(open('gpurun_out/ab.json'))
print(sys.argv[1], round(d['value'] / 1e9, 1), 'G/s', round(d['ms_per_step'], 2), 'ms', {k: round(v, 2) for k, v in d['roofline']['all_kernels_ms_per_step'].items()})
PY
done
