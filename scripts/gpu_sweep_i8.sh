#!/bin/bash
# GPU-box helper: parity tests, then bench the exact-integer path for several resident-wave targets.
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
tail -3 gpurun_out/pytest_gpu.log
for cfg in "$@"; do
  set -- $cfg
  GPCA_GQ_WAVES=$1 GPCA_GTT_WAVES=$2 timeout -k 10 200 python bench.py --precision i8 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/b8.json 2> gpurun_out/b8.err || tail -3 gpurun_out/b8.err
  python - "$cfg" <<'PY'
import json, sys
d = json.load(open('gpurun_out/b8.json'))
print(sys.argv[1], round(d['value'] / 1e9, 1), 'G/s', round(d['ms_per_step'], 2), 'ms',
      {k: round(v, 2) for k, v in d['roofline']['all_kernels_ms_per_step'].items()})
PY
done
