#!/bin/bash
# GPU-box helper: the round's measured artefacts.  usage: gpu_profile.sh <tag>
#   1. bench.py (default config, with cpu_baseline)            -> gpurun_out/bench_<tag>.json
#   2. rocprofv3 --kernel-trace --stats of the same command    -> gpurun_out/prof_<tag>/
#   3. PMC passes (separate runs, counters only): HBM bytes, MFMA busy
# (the default bench.py run holds only full-matrix launches of every GEMM kernel, so per-kernel averages are per-launch figures)
tag=$1
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || tail -5 gpurun_out/bench_$tag.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py --no-cpu-baseline --no-extras > gpurun_out/prof_$tag.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_$tag -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > gpurun_out/pmc_fetch_$tag.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_$tag -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > gpurun_out/pmc_write_$tag.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq_$tag -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > gpurun_out/pmc_sq_$tag.log 2>&1
python scripts/summarize_profile.py $tag
