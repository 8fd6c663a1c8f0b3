#!/bin/bash
# usage: gpu_stats.sh <tag> [bench args...]  -> per-kernel table of one profiled bench run
tag=$1; shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/st_$tag -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > gpurun_out/st_$tag.log 2>&1
python - gpurun_out/st_$tag <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print(f"{r['Name'].replace('void ','')[:60]:60s} {r['Calls']:>4s} tot {float(r['TotalDurationNs'])/1e6:8.3f} ms  avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
