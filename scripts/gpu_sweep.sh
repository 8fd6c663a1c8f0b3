#!/bin/bash
# GPU-box helper: parity tests, then a sweep of the resident-wave targets of the two GEMM grids.
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
tail -3 gpurun_out/pytest_gpu.log
for cfg in "$@"; do
  set -- $cfg
  GPCA_GQ_WAVES=$1 GPCA_GTT_WAVES=$2 timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/b.err || tail -3 gpurun_out/b.err
  python - "$cfg" <<'PY'
import json, sys
d = json.load(open('gpurun_out/b.json'))
print(sys.argv[1], round(d['value'] / 1e9, 1), 'G/s', round(d['ms_per_step'], 2), 'ms',
      {k: round(v, 2) for k, v in d['roofline']['all_kernels_ms_per_step'].items()})
PY
done
