#!/bin/bash
# GPU-box helper (round 3): whole suite + smoke, then a quick bench line and the Omega kernel's time from a kernel trace
tag=$1
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x --durations=5 > gpurun_out/pytest_$tag.log 2>&1; rc=$?
tail -10 gpurun_out/pytest_$tag.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$tag.log 2>&1 || rc=1; tail -2 gpurun_out/smoke_$tag.log
if grep -q "Memory access fault" gpurun_out/pytest_$tag.log gpurun_out/smoke_$tag.log; then echo "GPU FAULT"; exit 1; fi
[ $rc -ne 0 ] && exit $rc
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py --no-cpu-baseline --no-extras > gpurun_out/prof_$tag.log 2>&1
f=$(ls gpurun_out/prof_$tag/*/*kernel_stats.csv | head -1)
python - "$f" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("k_omega", "k_gq_d", "k_gtt_d", "k_quantize", "k_rightmul", "k_chol")):
        print(n.split("(")[0][:60], r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1))
P
grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_$tag.log | head -4
