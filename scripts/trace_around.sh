#!/bin/bash
# kernel trace of one bench.py command line; prints the launches around every kernel whose name contains <pattern> (start, duration, gap).
# usage: scripts/gpu.sh <tag> "sh=scripts/trace_around.sh,<pattern> <bench.py arguments>"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
pat=$1; shift
rm -rf gpurun_out/trace_around
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_around -- python bench.py "$@" > gpurun_out/trace_around.json 2> gpurun_out/trace_around.err || { tail -5 gpurun_out/trace_around.err; exit 1; }
f=$(find gpurun_out/trace_around -name '*kernel_trace.csv' | head -1)
python - "$f" "$pat" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
hits = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
for i in hits[-3:]:
    print("--")
    for j in range(max(0, i - 4), min(len(rows), i + 4)):
        r = rows[j]; s = int(r["Start_Timestamp"]); e = int(r["End_Timestamp"])
        print("  %s start %12.1f us dur %10.1f us queue %s  %s" % ("*" if j == i else " ", (s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
PY
rm -rf gpurun_out/trace_around
