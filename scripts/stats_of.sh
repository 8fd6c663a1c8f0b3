#!/bin/bash
# per-kernel totals (rocprofv3 --kernel-trace --stats) of one bench.py command line.  usage: scripts/gpu.sh <tag> "sh=scripts/stats_of.sh,<bench.py arguments>"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/stats_of
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_of -- python bench.py "$@" > gpurun_out/stats_of.json 2> gpurun_out/stats_of.err || { tail -5 gpurun_out/stats_of.err; exit 1; }
f=$(find gpurun_out/stats_of -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:28]:
    print("   %-64s calls %6s total %9.2f ms avg %9.1f us" % (r["Name"][:64], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
rm -rf gpurun_out/stats_of
