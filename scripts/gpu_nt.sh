#!/bin/bash
mkdir -p gpurun_out
for nt in 0 1; do
  GPCA_STREAM_NT=$nt GPCA_GTT_WAVES=2048 timeout -k 10 200 python bench.py --precision i8 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/b8.json 2> gpurun_out/b8.err || tail -3 gpurun_out/b8.err
  python - "nt=$nt" <<'PY'
import json, sys
d = json.load(open('gpurun_out/b8.json'))
print(sys.argv[1], round(d['value'] / 1e9, 1), 'G/s', round(d['ms_per_step'], 2), 'ms', {k: round(v, 2) for k, v in d['roofline']['all_kernels_ms_per_step'].items()})
PY
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
GPCA_GTT_WAVES=2048 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f8 -- python bench.py --precision i8 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_f8.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_f8/*/*counter_collection.csv')[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'].replace('void ','')[:30]].append(float(r['Counter_Value']))
for k, v in agg.items():
    if 'i8' in k or 'snp_stats' in k: print(k, 'FETCH_SIZE avg MB', sum(v)/len(v)*1024/1e6, 'x2', 2*sum(v)/len(v)*1024/1e6)
PY
