#!/bin/bash
# GPU-box helper: config-5 shard out of core, with and without the HBM panel cache.  usage: gpu_stream_cache.sh <tag>
tag=$1
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
for c in -1 0; do
  timeout -k 10 420 python bench.py --streamed --snps 6250000 --samples 500000 -k 40 --storage 2bit --steps 1 --warmup 0 --cache-gb $c > gpurun_out/stream_c5_2bit_cache${c}_$tag.json 2> gpurun_out/stream_c5_2bit_cache${c}_$tag.err || { tail -5 gpurun_out/stream_c5_2bit_cache${c}_$tag.err; exit 1; }
  echo "--- c5 2bit cache $c"; python - <<PY
import json
d=json.loads(open('gpurun_out/stream_c5_2bit_cache${c}_$tag.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['panels_cached_in_hbm'], d['streaming'], d['snp_stats_s'])
PY
done
