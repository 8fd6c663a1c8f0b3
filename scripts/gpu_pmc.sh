#!/bin/bash
# GPU-box helper: one PMC pass over a short bench run.  usage: gpu_pmc.sh <outdir> <counters...>
out=$1; shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/$out -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/$out.log 2>&1
grep -h '"metric"' gpurun_out/$out.log | head -1 | cut -c1-200
python - gpurun_out/$out <<'PY'
import csv, collections, glob, sys
f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].replace('void ', '')[:28]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
for k, v in agg.items():
    if any(x in k for x in ('gq_f32', 'gtt_f32', 'snp_stats', 'gq_d', 'gtt_d', 'gq_2bit', 'gtt_p')):
        print(k, 'avg_ms=%.3f' % (sum(dur[k]) / len(dur[k])), {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
