#!/bin/bash
# configs[4]'s per-GPU shard, streamed: is the line bound by the panel generator or by the GEMMs?  The same call under rocprofv3
# --kernel-trace --stats with the generator running and with GPCA_SOURCE_BENCH_HOLD (no panel regenerated): per-kernel totals of both.
# usage: scripts/gpu.sh <tag> sh=scripts/config5_who_bounds.sh     -> gpurun_out/sh_<tag>.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--streamed --snps 6250000 --samples 500000 -k 40 --storage 2bit --steps 1 --warmup 1"
for mode in real hold; do
  extra=""; [ $mode = hold ] && extra="--bench-hold"
  rm -rf gpurun_out/c5_$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c5_$mode -- python bench.py $ARGS $extra > gpurun_out/c5_$mode.json 2> gpurun_out/c5_$mode.err || { tail -5 gpurun_out/c5_$mode.err; exit 1; }
  echo "== $mode: $(python -c "import json,sys; d=json.loads(open('gpurun_out/c5_$mode.json').read().strip().splitlines()[-1]); print('ms_per_step', round(d['ms_per_step'],1), 'value', '%.3e' % d['value'], 'gemm_sweeps_ms', round(d['streaming']['gemm_sweeps_ms_per_step'],1))")"
  f=$(find gpurun_out/c5_$mode -name '*kernel_stats.csv' | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:8]:
    print("   %-60s calls %6s total %9.1f ms avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
  rm -rf gpurun_out/c5_$mode
done
